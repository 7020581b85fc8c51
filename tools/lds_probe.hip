// Development probe: how many 512-thread workgroups with S bytes of dynamic LDS does a CU of this GPU hold?
//   hipcc -O2 --offload-arch=gfx950 tools/lds_probe.hip -o tools/bin/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 6) void k(float* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = lds[(threadIdx.x * 7) & 511];
}
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int prev = -1;
    for (int s = 30 * 1024; s <= 160 * 1024; s += 256) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 512, (size_t)s) != hipSuccess) { printf("query failed at %d\n", s); break; }
        if (n != prev) { printf("%7d bytes: %d workgroups per CU\n", s, n); prev = n; }
    }
    return 0;
}
