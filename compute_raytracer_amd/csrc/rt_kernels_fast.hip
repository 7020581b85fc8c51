// Fast arithmetic build of the ray-trace kernels: compiled with -ffp-contract=fast (v_fma_f32
// wherever a multiply feeds an add).  Parity tolerance: tests/test_parity_gpu.py.
#define RT_SUFFIX fast
#include "rt_kernels.inc"
