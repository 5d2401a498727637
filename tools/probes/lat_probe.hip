// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/lat_probe tools/probes/lat_probe.hip
// Developer probe: dependent-load round-trip time on MI355X, one wave alone vs a full grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void chase(const uint32_t *next, uint32_t steps, uint32_t stride_elems, unsigned long long *out, uint32_t *sink) {
    uint32_t p = (blockIdx.x * 256u + threadIdx.x) * stride_elems % (1u << 16);
    const unsigned long long t0 = wall_clock64();
    for (uint32_t i = 0; i < steps; ++i) p = next[p];
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (p == 0xffffffffu) *sink = p;
}

__global__ void barrier_probe(uint32_t steps, unsigned long long *out) {
    __shared__ uint32_t s[4];
    const unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
    for (uint32_t i = 0; i < steps; ++i) {
        if ((threadIdx.x & 63u) == 0) s[threadIdx.x >> 6] = i + acc;
        __syncthreads();
        acc += s[0] + s[1] + s[2] + s[3];
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) + (acc & 0u);
}

int main() {
    const uint32_t n = 1u << 16;  // 256 KB table: L2 resident
    std::vector<uint32_t> h(n);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(1);
    // one big cycle (Sattolo)
    for (uint32_t i = n - 1; i > 0; --i) { std::uniform_int_distribution<uint32_t> d(0, i - 1); std::swap(h[i], h[d(rng)]); }
    uint32_t *d_next, *d_sink; unsigned long long *d_out;
    hipMalloc(&d_next, n * 4); hipMalloc(&d_sink, 4); hipMalloc(&d_out, 4096 * 8);
    hipMemcpy(d_next, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned long long> o(4096);
    const uint32_t steps = 64;
    for (uint32_t grid : {1u, 256u, 1024u, 2048u}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, 0, d_next, steps, 97u, d_out, d_sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(o.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
        std::sort(o.begin(), o.begin() + grid);
        printf("chase grid %4u x256 thr: per dependent load p50 %.0f ns  max %.0f ns\n", grid, o[grid / 2] * 10.0 / steps, o[grid - 1] * 10.0 / steps);
    }
    for (uint32_t grid : {1u, 1024u}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(barrier_probe, dim3(grid), dim3(256), 0, 0, 256u, d_out);
            hipDeviceSynchronize();
        }
        hipMemcpy(o.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
        std::sort(o.begin(), o.begin() + grid);
        printf("barrier pair + LDS, grid %4u: p50 %.0f ns per iteration\n", grid, o[grid / 2] * 10.0 / 256);
    }
    // back-to-back launches of a long-running chase to see if clocks ramp: time 200 launches
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(chase, dim3(1024), dim3(256), 0, 0, d_next, steps, 97u, d_out, d_sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(o.data(), d_out, 1024 * 8, hipMemcpyDeviceToHost);
        std::sort(o.begin(), o.begin() + 1024);
        printf("round %d: 200 launches %.3f ms; last launch per-load p50 %.0f ns\n", round, ms, o[512] * 10.0 / steps);
    }
    return 0;
}
