// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/probes/bin/mix_probe tools/probes/mix_probe.hip
// Developer probe: half(a * b) with f32 a, b -- the compiler turns it into v_fma_mixlo_f16(a, b, 0).
// Does that round twice (f32, then f16) like the source says, or once?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __host__ inline uint32_t Hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void probe(const uint32_t *u, const uint32_t *v, uint32_t n, uint16_t *out, uint16_t *out2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = __uint_as_float(u[i]);
    const float b = __uint_as_float(v[i]);
    const _Float16 r = static_cast<_Float16>(a * b);
    out[i] = __builtin_bit_cast(uint16_t, r);
    float p = a * b;
    asm volatile("" : "+v"(p));  // the product as an f32 value in a register: two separate roundings
    out2[i] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(p));
}
int main() {
    const uint32_t n = 1u << 24;
    std::vector<uint32_t> h(n), v(n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t u = Hash(2 * i + 1), w = Hash(2 * i + 2);
        h[i] = (u & 0x807fffffu) | ((118u + (u >> 8) % 12u) << 23);  // |a| in [2^-9, 8)
        v[i] = (w & 0x807fffffu) | ((118u + (w >> 8) % 12u) << 23);
    }
    uint32_t *dh, *dv; uint16_t *dout, *dout2;
    (void)hipMalloc(&dh, n * 4); (void)hipMalloc(&dv, n * 4); (void)hipMalloc(&dout, n * 2); (void)hipMalloc(&dout2, n * 2);
    (void)hipMemcpy(dh, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dh, dv, n, dout, dout2);
    std::vector<uint16_t> o(n), o2(n);
    (void)hipMemcpy(o.data(), dout, n * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(o2.data(), dout2, n * 2, hipMemcpyDeviceToHost);
    FILE *f = fopen("gpurun_out/mix_probe.bin", "wb");
    fwrite(&n, 4, 1, f); fwrite(h.data(), 4, n, f); fwrite(v.data(), 4, n, f); fwrite(o.data(), 2, n, f); fwrite(o2.data(), 2, n, f); fclose(f);
    printf("wrote %u cases\n", n);
    return 0;
}
