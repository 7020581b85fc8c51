// Strict arithmetic build of the ray-trace kernels: compiled with -ffp-contract=off so that
// no multiply-add is fused; division and square root are the correctly rounded ones hipcc
// emits by default.  Expected to reproduce oracle/rt_oracle.c bit for bit.
// Also owns the per-frame scene preparation kernel, which both modes share (its hoisted
// values must be the oracle's `origin - center` and `c` exactly).
#define RT_SUFFIX strict
#define RT_DEFINE_PREP 1
#include "rt_kernels.inc"
