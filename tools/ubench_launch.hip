// Host-side cost of getting ~2.4 KB of per-frame data in front of a kernel: a kernel whose kernarg block carries it,
// versus small host-to-device copies.  hipcc --offload-arch=gfx950 -O2 tools/ubench_launch.hip -o tools/bin/ubench_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Big { float* dst; unsigned n; float data[584]; };
struct Small { float* dst; const float* src; unsigned n; };
__global__ void k_big(const Big a) { for (unsigned i = threadIdx.x; i < a.n; i += 256) a.dst[i] = a.data[i]; }
__global__ void k_small(const Small a) { for (unsigned i = threadIdx.x; i < a.n; i += 256) a.dst[i] = a.src[i]; }
__global__ void k_work(float* p, int iters) { float x = p[threadIdx.x]; for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f; p[threadIdx.x] = x; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float *d, *dsrc, *hp, *work; hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipMalloc(&d, 4096); hipMalloc(&dsrc, 4096); hipMalloc(&work, 4096); hipHostMalloc(&hp, 4096);
    static float pageable[1024];
    Big b; b.dst = d; b.n = 584; memset(b.data, 0, sizeof b.data);
    Small sm{d, dsrc, 584};
    const int K = 200;
    auto run = [&](const char* name, auto&& body, bool serial) {
        for (int w = 0; w < 20; ++w) { body(); hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, work, 20000); if (serial) hipStreamSynchronize(s); }
        hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < K; ++i) { body(); hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, work, 20000); if (serial) hipStreamSynchronize(s); }
        hipStreamSynchronize(s);
        printf("%-52s %s: %.1f us per frame\n", name, serial ? "one at a time" : "back to back ", (now() - t0) / K);
    };
    for (int serial = 1; serial >= 0; --serial) {
        run("work kernel alone", [&] {}, serial);
        run("+ kernel with a 2.4 KB kernarg block", [&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(256), 0, s, b); }, serial);
        run("+ kernel with a 24 B kernarg block", [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, s, sm); }, serial);
        run("+ hipMemcpyAsync 2.4 KB from pinned memory", [&] { hipMemcpyAsync(d, hp, 2336, hipMemcpyHostToDevice, s); }, serial);
        run("+ hipMemcpyAsync 2.4 KB from pageable memory", [&] { hipMemcpyAsync(d, pageable, 2336, hipMemcpyHostToDevice, s); }, serial);
        run("+ 3 x hipMemcpyAsync (1 KB, 320 B, 16 B) pinned", [&] { hipMemcpyAsync(d, hp, 992, hipMemcpyHostToDevice, s); hipMemcpyAsync(d + 256, hp + 256, 320, hipMemcpyHostToDevice, s); hipMemcpyAsync(d + 512, hp + 512, 16, hipMemcpyHostToDevice, s); }, serial);
    }
    return 0;
}
