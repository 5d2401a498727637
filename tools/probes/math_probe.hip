// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/probes/bin/math_probe tools/probes/math_probe.hip
// Developer probe: are sqrtf, f32 division and f32->f16 conversion as compiled for the kernels
// correctly rounded on gfx950?  Writes inputs and results to math_probe.bin for numpy to check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __host__ inline uint32_t Hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void probe(const uint32_t *a, const uint32_t *b, uint32_t n, uint32_t *out_sqrt, uint32_t *out_div, uint16_t *out_h) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(a[i]), y = __uint_as_float(b[i]);
    out_sqrt[i] = __float_as_uint(sqrtf(fabsf(x)));
    out_div[i] = __float_as_uint(x / y);
    const _Float16 h = static_cast<_Float16>(x);
    out_h[i] = __builtin_bit_cast(uint16_t, h);
}

int main() {
    const uint32_t n = 1u << 24;
    std::vector<uint32_t> a(n), b(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t u = Hash(i * 2u + 1u), v = Hash(i * 2u + 2u);
        // mostly "geometry sized" values, some anywhere
        if (i & 3u) { u = (u & 0x807fffffu) | ((96u + (Hash(u) % 48u)) << 23); v = (v & 0x807fffffu) | ((96u + (Hash(v) % 48u)) << 23); }
        a[i] = u; b[i] = v;
    }
    uint32_t *da, *db, *ds, *dd; uint16_t *dh;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dh, n * 2);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, da, db, n, ds, dd, dh);
    std::vector<uint32_t> s(n), d(n); std::vector<uint16_t> h(n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost);
    FILE *f = fopen("gpurun_out/math_probe.bin", "wb");
    fwrite(&n, 4, 1, f); fwrite(a.data(), 4, n, f); fwrite(b.data(), 4, n, f); fwrite(s.data(), 4, n, f); fwrite(d.data(), 4, n, f); fwrite(h.data(), 2, n, f);
    fclose(f);
    printf("wrote %u cases\n", n);
    return 0;
}
