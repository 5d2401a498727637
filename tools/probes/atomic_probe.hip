// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe tools/probes/atomic_probe.hip
// Developer probe (round 2): what does a dynamic work hand-out cost on gfx950?  1024 workgroups x 4
// waves (the frame kernels' persistent grid); every wave takes K tickets from one of NP counters
// (each in its own 128-byte line), with some dependent ALU work between tickets.  Reports the
// latency of one ticket (100 MHz wall clock) and the kernel's span.  Scopes: agent (device-wide
// atomics) and, with the counter picked by the wave's XCC_ID, workgroup scope (executes in the
// XCD's own L2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kTickets = 3;

__device__ __forceinline__ uint32_t XccId() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

template <int SCOPE>  // 0 agent, 1 workgroup scope + XCC-local counter
__global__ __launch_bounds__(256) void probe(uint32_t *ctr, uint32_t np, unsigned long long *lat, uint32_t *tickets, uint32_t *xcc_out,
                                             int work) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t part = wave % np;
    if (SCOPE == 1) part = XccId() * (np / 8u) + (wave / 8u) % (np / 8u);
    uint32_t *c = ctr + 32u * part;
    float acc = static_cast<float>(lane);
    unsigned long long worst = 0, sum = 0;
    for (int k = 0; k < kTickets; ++k) {
        const unsigned long long t0 = wall_clock64();
        uint32_t t = 0;
        if (lane == 0) {
            if (SCOPE == 0)
                t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        const unsigned long long t1 = wall_clock64();
        worst = max(worst, t1 - t0);
        sum += t1 - t0;
        if (lane == 0) tickets[(static_cast<size_t>(wave) * kTickets + k)] = (part << 20) | t;
        for (int i = 0; i < work; ++i) acc = 1.0001f / (acc + 0.5f);  // ~30 ns each
    }
    if (lane == 0) {
        lat[2 * wave] = sum;
        lat[2 * wave + 1] = worst;
        xcc_out[wave] = XccId() | (blockIdx.x << 8);
    }
    if (acc == 123.456f) tickets[0] = 1;
}

int main() {
    const int grid = 1024, waves = grid * 4;
    uint32_t *ctr, *tickets, *xcc;
    unsigned long long *lat;
    hipMalloc(&ctr, 4096 * 128);
    hipMalloc(&tickets, waves * kTickets * 4);
    hipMalloc(&xcc, waves * 4);
    hipMalloc(&lat, waves * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<uint32_t> hx(waves);
    for (int scope = 0; scope < 2; ++scope)
        for (int work : {0, 200})
            for (uint32_t np : {1u, 8u, 32u, 128u, 512u}) {
                if (scope == 1 && np < 8) continue;
                float best = 1e9f;
                std::vector<unsigned long long> hl(waves * 2);
                std::vector<uint32_t> ht(waves * kTickets);
                for (int rep = 0; rep < 5; ++rep) {
                    hipMemset(ctr, 0, 4096 * 128);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    if (scope == 0)
                        probe<0><<<grid, 256>>>(ctr, np, lat, tickets, xcc, work);
                    else
                        probe<1><<<grid, 256>>>(ctr, np, lat, tickets, xcc, work);
                    hipEventRecord(e1);
                    hipDeviceSynchronize();
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = std::min(best, ms);
                }
                hipMemcpy(hl.data(), lat, waves * 16, hipMemcpyDeviceToHost);
                hipMemcpy(ht.data(), tickets, waves * kTickets * 4, hipMemcpyDeviceToHost);
                hipMemcpy(hx.data(), xcc, waves * 4, hipMemcpyDeviceToHost);
                // every ticket of every partition exactly once?
                std::sort(ht.begin(), ht.end());
                size_t dup = 0;
                for (size_t i = 1; i < ht.size(); ++i) dup += ht[i] == ht[i - 1];
                double mean = 0;
                unsigned long long worst = 0;
                for (int w = 0; w < waves; ++w) {
                    mean += hl[2 * w];
                    worst = std::max(worst, hl[2 * w + 1]);
                }
                mean = mean / (waves * kTickets) * 10.0;
                printf("scope %s work %3d np %3u: span %.1f us, ticket latency mean %.0f ns worst %.0f ns, duplicate tickets %zu\n",
                       scope ? "wg+xcc" : "agent ", work, np, best * 1e3, mean, worst * 10.0, dup);
            }
    // XCC of workgroup b
    int hist[16][16] = {};
    for (int w = 0; w < waves; w += 4) hist[(hx[w] >> 8) % 8][hx[w] & 15]++;
    printf("workgroups by (blockIdx %% 8) x XCC_ID:\n");
    for (int a = 0; a < 8; ++a) {
        for (int b = 0; b < 8; ++b) printf("%5d", hist[a][b]);
        printf("\n");
    }
    return 0;
}
