#include "rcx_dev.h"
static void launch_serial(hipStream_t s, int codec, rcx_kargs& k, int v) { hipLaunchKernelGGL(k_not_built, dim3((k.nblocks + 63) / 64), dim3(64), 0, s, k); }
