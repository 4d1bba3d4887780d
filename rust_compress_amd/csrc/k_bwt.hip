#include "rcx_dev.h"
#include <string>
static uint64_t bwt_forward_scratch_bytes(uint32_t nblocks, uint64_t max_block) { return 0; }
static uint64_t bwt_inverse_scratch_bytes(uint32_t nblocks, uint64_t max_block) { return 0; }
static int launch_bwt_forward(hipStream_t s, rcx_kargs& k, int v, std::string& err) { err = "bwt forward not built yet"; return RCX_RC_BAD_ARG; }
static int launch_bwt_inverse(hipStream_t s, rcx_kargs& k, int v, std::string& err) { err = "bwt inverse not built yet"; return RCX_RC_BAD_ARG; }
