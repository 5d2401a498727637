// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/probes/bin/issue_probe tools/probes/issue_probe.hip
// Developer probe (round 2): what a wave can issue on gfx950.  One workgroup of 64*W*4 threads on
// one CU = W waves per SIMD; loops of (a) dependent f32 adds, (b) 4 independent chains,
// (c) IEEE divisions, (d) sqrtf, (e) dependent LDS reads (uniform address, like the command fetch),
// (f) LDS read + readfirstlane + scalar branch (the interpreter's dispatch).  Reports shader
// cycles (s_memtime) and ns (s_memrealtime, 100 MHz) per operation per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 2048;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float *out, unsigned long long *clk, float seed, const uint32_t *tab) {
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x & 1023] = (threadIdx.x * 7u + 1u) & 1023u;
    __syncthreads();
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f, c = seed * 0.25f, d = seed * 0.125f;
    uint32_t p = threadIdx.x >> 6;  // wave-uniform
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = clock64();
    if (MODE == 0) {
#pragma unroll 16
        for (int i = 0; i < kIters; ++i) a = a + 1.0001f;
    } else if (MODE == 1) {
#pragma unroll 4
        for (int i = 0; i < kIters / 4; ++i) { a = a + 1.0001f; b = b + 1.0002f; c = c + 1.0003f; d = d + 1.0004f; }
    } else if (MODE == 2) {
#pragma unroll 4
        for (int i = 0; i < kIters; ++i) a = 1.0001f / (a + 0.5f);
    } else if (MODE == 3) {
#pragma unroll 4
        for (int i = 0; i < kIters; ++i) a = sqrtf(a + 1.5f);
    } else if (MODE == 4) {
#pragma unroll 4
        for (int i = 0; i < kIters; ++i) p = lds[p];
    } else if (MODE == 5) {
        for (int i = 0; i < kIters; ++i) {
            p = lds[p];
            const uint32_t t = __builtin_amdgcn_readfirstlane(p) & 3u;
            if (t == 0) a += 1.0f; else if (t == 1) a *= 1.0001f; else if (t == 2) a -= 0.5f; else a = a * 0.5f + 1.0f;
        }
    } else if (MODE == 6) {  // 4 independent divisions per step
#pragma unroll 2
        for (int i = 0; i < kIters / 4; ++i) { a = 1.0001f / (a + 0.5f); b = 1.0002f / (b + 0.5f); c = 1.0003f / (c + 0.5f); d = 1.0004f / (d + 0.5f); }
    } else if (MODE == 7) {  // packed half blend x + (y - x) * a, dependent
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 x; x.x = (_Float16)a; x.y = (_Float16)b; h2 y; y.x = (_Float16)0.3f; y.y = (_Float16)0.6f; h2 al; al.x = (_Float16)0.01f; al.y = (_Float16)0.02f;
#pragma unroll 8
        for (int i = 0; i < kIters / 3; ++i) x = x + (y - x) * al;
        a = (float)x.x + (float)x.y;
    }
    const unsigned long long c1 = clock64();
    const unsigned long long r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + p;
    if ((threadIdx.x & 63) == 0) {
        clk[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = c1 - c0;
        clk[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char *name, float *d_out, unsigned long long *d_clk) {
    for (int threads : {64, 256, 512, 1024}) {
        for (int grid : {1, 1024}) {
            if (grid > 1 && threads != 256 && threads != 1024) continue;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), 0, 0, d_out, d_clk, 1.0f, nullptr);
                hipDeviceSynchronize();
            }
            const int waves = grid * threads / 64;
            std::vector<unsigned long long> h(waves * 2);
            hipMemcpy(h.data(), d_clk, waves * 16, hipMemcpyDeviceToHost);
            double cyc = 0, ns = 0;
            for (int w = 0; w < waves; ++w) { cyc += h[2 * w]; ns += h[2 * w + 1] * 10.0; }
            cyc /= waves; ns /= waves;
            printf("%-28s grid %4d x %4d thr (%d waves/SIMD%s): %.1f clk64-ticks/op  %.2f ns/op  (tick = %.3f ns)\n", name, grid, threads,
                   threads <= 256 ? 1 : threads / 256, grid > 1 ? ", chip full" : "", cyc / kIters, ns / kIters, ns / cyc);
        }
    }
}

int main() {
    float *d_out; unsigned long long *d_clk;
    hipMalloc(&d_out, 1024 * 1024 * 4); hipMalloc(&d_clk, 1024 * 16 * 16);
    run<0>("dependent v_add_f32", d_out, d_clk);
    run<1>("4 independent v_add_f32", d_out, d_clk);
    run<2>("dependent IEEE divide", d_out, d_clk);
    run<6>("4 independent IEEE divides", d_out, d_clk);
    run<3>("dependent sqrtf", d_out, d_clk);
    run<4>("dependent LDS read (uniform)", d_out, d_clk);
    run<5>("LDS read+readfirstlane+branch", d_out, d_clk);
    run<7>("dependent pk half blend (3 op)", d_out, d_clk);
    return 0;
}
