// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/icache_probe tools/probes/icache_probe.hip
// Developer probe (round 4): what does straight-line code that a wave executes ONCE cost when the launch
// starts with a cold instruction cache?  The frame kernels are 20-27 KB of mostly straight-line code that a
// workgroup walks once per strip row / tile; a mean strip row takes 14 us for a few thousand instructions.
// Every wave runs a body of N independent-ish VALU instructions (8 accumulators, no memory operations)
// twice: pass 1 right after launch (cold), pass 2 over the same code (warm).  Reports per-wave ticks
// (100 MHz wall clock) for both passes and for launches back to back (is the cache kept across launches?).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int N>
__device__ __forceinline__ void Body(float (&a)[8]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // one v_fma per line, constants differ per step so that nothing is rolled up or shared
        a[i & 7] = __builtin_fmaf(a[i & 7], 1.0f + static_cast<float>(i) * 1e-6f, a[(i + 3) & 7]);
    }
}

template <int N>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, float *sink) {
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = static_cast<float>(threadIdx.x + k);
    const unsigned long long t0 = wall_clock64();
    unsigned long long t1 = 0, t2 = 0;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        Body<N>(a);
        asm volatile("" ::: "memory");
        if (pass == 0) t1 = wall_clock64();
        else t2 = wall_clock64();
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63u) == 0) {
        const size_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
        out[3 * w] = t0;
        out[3 * w + 1] = t1 - t0;
        out[3 * w + 2] = t2 - t1;
    }
}

template <int N>
void Run(int grid, const char *name) {
    unsigned long long *d = nullptr;
    float *sink = nullptr;
    hipMalloc(&d, sizeof(unsigned long long) * 3 * grid * 4);
    hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(3 * grid * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<N>, dim3(grid), dim3(256), 0, 0, d, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> p1, p2;
        unsigned long long tmin = ~0ull, tend = 0;
        for (int w = 0; w < grid * 4; ++w) {
            p1.push_back(h[3 * w + 1] * 0.01);
            p2.push_back(h[3 * w + 2] * 0.01);
            tmin = std::min(tmin, h[3 * w]);
            tend = std::max(tend, h[3 * w] + h[3 * w + 1] + h[3 * w + 2]);
        }
        std::sort(p1.begin(), p1.end());
        std::sort(p2.begin(), p2.end());
        auto q = [](const std::vector<double> &v, double f) { return v[static_cast<size_t>(f * (v.size() - 1))]; };
        printf("%s grid %d rep %d: kernel %.1f us (in-kernel span %.1f); pass1 us min %.2f p50 %.2f p90 %.2f max %.2f | pass2 min %.2f p50 %.2f p90 %.2f max %.2f\n",
               name, grid, rep, ms * 1e3, (tend - tmin) * 0.01, q(p1, 0), q(p1, .5), q(p1, .9), q(p1, 1), q(p2, 0), q(p2, .5), q(p2, .9), q(p2, 1));
    }
    hipFree(d);
    hipFree(sink);
}

int main() {
    // instructions per pass: 8 bytes each (v_fma_f32 with a literal is 12)
    Run<256>(1280, "N=256  (~3 KB)");
    Run<1024>(1280, "N=1024 (~12 KB)");
    Run<2048>(1280, "N=2048 (~24 KB)");
    Run<2048>(256, "N=2048 (~24 KB)");
    Run<4096>(1280, "N=4096 (~48 KB)");
    return 0;
}
