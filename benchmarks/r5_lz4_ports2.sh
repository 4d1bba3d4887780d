#!/bin/bash
# Round 5, second injection experiment: what a VECTOR instruction costs the LZ4 decoder by kind (no register source, one, two, three) and
# by where it is issued (in front of emit5 at the executor's plain priority: RCX_DUMMY_*; behind the switch to the copy rounds' priority:
# RCX_DUMMY3_*).  100 instructions a batch each.
FLAGSETS="-DRCX_NONE=1|-DRCX_DUMMY3_MOV=100|-DRCX_DUMMY3_ADD1=100|-DRCX_DUMMY3_ADD1I=100|-DRCX_DUMMY3_ADD2=100|-DRCX_DUMMY3_ALIGN=100|-DRCX_DUMMY3_SALU=100|-DRCX_DUMMY_VALU=100|-DRCX_DUMMY_SALU=100|-DRCX_NONE=2" bash benchmarks/r5_lz4_flagcount.sh
