// tu_bwt.hip -- BWT forward (suffix sorting) and inverse kernels + their launch code (one translation unit).
#include "rcx_tu.h"
#include "k_bwt.hip"
#include "k_bwt_inverse.hip"

int rcx_tu_bwt_forward(hipStream_t s, rcx_kargs& k, int variant, std::string& err, bool sa_words) { return launch_bwt_forward(s, k, variant, err, sa_words); }
int rcx_tu_bwt_inversion_table(hipStream_t s, rcx_kargs& k) { return launch_bwt_inversion_table(s, k); }
int rcx_tu_bwt_inverse(hipStream_t s, rcx_kargs& k, int variant, std::string& err, bool minimal) { return launch_bwt_inverse(s, k, variant, err, minimal); }
uint64_t rcx_tu_bwt_forward_scratch(uint32_t nblocks, uint64_t max_block) { return bwt_forward_scratch_bytes(nblocks, max_block); }
uint64_t rcx_tu_bwt_inverse_scratch(uint32_t nblocks, uint64_t max_block) { return bwt_inverse_scratch_bytes(nblocks, max_block); }
