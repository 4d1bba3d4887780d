// tu_serial.hip -- MTF, DC, RLE and the range coders (one stream per wave or lane) + their launch code (one translation unit).
#include "rcx_tu.h"
#include "k_serial.hip"

void rcx_tu_serial(hipStream_t s, int codec, rcx_kargs& k, int variant, uint32_t param) { launch_serial(s, codec, k, variant, param); }
uint64_t rcx_tu_dc_encode_scratch(uint32_t nblocks, uint64_t max_block) { return max_block < DCX_MIN ? 0 : dc_encode_scratch_bytes(nblocks); }
