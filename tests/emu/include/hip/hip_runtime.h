// TEST INFRASTRUCTURE -- never shipped, never measured, never loaded by the product.
//
// A stand-in for <hip/hip_runtime.h> that lets the kernels of piet_metal_amd/csrc/*.hip be
// compiled as plain C++ and run LANE BY LANE ON THE CPU, wave64 semantics included (ballots,
// readlane, DPP, bpermute, workgroup barriers), so that kernel logic can be checked against the
// oracle on a box without a GPU (tests/test_emu_cpu.py).  Every lane of a workgroup is a fiber;
// a cross-lane operation parks the lane until all live lanes of its wave have arrived, then the
// scheduler computes every lane's result (tests/emu/emu_runtime.cpp).  Workgroups run one after
// the other.  Nothing here is fast and nothing here is a fallback: libpiet_metal_amd.so never sees
// this file, and the emulated library is only ever loaded by tests (tests/conftest.py, PM_TEST_EMU=1).
//
// What it cannot show: data races (lanes run one at a time between synchronisation points),
// performance, register pressure, instruction selection (v_fma_mix folding and friends).
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>

#define PM_EMU 1

// ---- qualifiers ------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

// ---- vector types ------------------------------------------------------------------------------
struct alignas(8) uint2 { uint32_t x, y; };
struct uint3 { uint32_t x, y, z; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 {
    uint32_t x, y, z;
    dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- per-lane identity (set by the scheduler whenever a lane is resumed) -----------------------
namespace pm_emu {
extern uint3 g_thread_idx, g_block_idx, g_block_dim, g_grid_dim;
}
#define threadIdx (pm_emu::g_thread_idx)
#define blockIdx (pm_emu::g_block_idx)
#define blockDim (pm_emu::g_block_dim)
#define gridDim (pm_emu::g_grid_dim)

// ---- cross-lane operations ---------------------------------------------------------------------
namespace pm_emu {
enum Op : int { kBallot = 1, kReadFirst, kReadLane, kDpp, kShfl, kShflUp, kWaveBarrier, kBlockBarrier };
// Parks the calling lane until every live lane of its wave has arrived at the same operation
// (same kind, same call site), then returns this lane's result.
uint64_t Collective(Op op, uint64_t a, uint64_t b, uint64_t c, void *site);
void BlockBarrier(void *site);
void Sleep();
uint32_t LaneId();
}  // namespace pm_emu

#define PM_EMU_SITE() __builtin_return_address(0)

__attribute__((noinline)) static uint64_t __ballot(int pred) { return pm_emu::Collective(pm_emu::kBallot, pred != 0, 0, 0, PM_EMU_SITE()); }
__attribute__((noinline)) static int __builtin_amdgcn_readfirstlane(int v) {
    return static_cast<int>(pm_emu::Collective(pm_emu::kReadFirst, static_cast<uint32_t>(v), 0, 0, PM_EMU_SITE()));
}
__attribute__((noinline)) static int __builtin_amdgcn_readlane(int v, int lane) {
    return static_cast<int>(pm_emu::Collective(pm_emu::kReadLane, static_cast<uint32_t>(v), static_cast<uint32_t>(lane), 0, PM_EMU_SITE()));
}
// old = what the lane keeps when its source is invalid and bound_ctrl is off (or its row/bank is masked)
__attribute__((noinline)) static int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const uint64_t packed = static_cast<uint64_t>(static_cast<uint32_t>(ctrl)) | (static_cast<uint64_t>(row_mask & 15) << 32) |
                            (static_cast<uint64_t>(bank_mask & 15) << 36) | (static_cast<uint64_t>(bound_ctrl ? 1 : 0) << 40);
    return static_cast<int>(pm_emu::Collective(pm_emu::kDpp, static_cast<uint32_t>(src), static_cast<uint32_t>(old), packed, PM_EMU_SITE()));
}
__attribute__((noinline)) static int __builtin_amdgcn_mov_dpp(int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const uint64_t packed = static_cast<uint64_t>(static_cast<uint32_t>(ctrl)) | (static_cast<uint64_t>(row_mask & 15) << 32) |
                            (static_cast<uint64_t>(bank_mask & 15) << 36) | (static_cast<uint64_t>(bound_ctrl ? 1 : 0) << 40);
    return static_cast<int>(pm_emu::Collective(pm_emu::kDpp, static_cast<uint32_t>(src), 0u, packed, PM_EMU_SITE()));
}
__attribute__((noinline)) static int __shfl(int v, int src_lane, int width = 64) {
    (void)width;
    return static_cast<int>(pm_emu::Collective(pm_emu::kShfl, static_cast<uint32_t>(v), static_cast<uint32_t>(src_lane), 0, PM_EMU_SITE()));
}
__attribute__((noinline)) static uint32_t __shfl_up(uint32_t v, unsigned delta, int width = 64) {
    (void)width;
    return static_cast<uint32_t>(pm_emu::Collective(pm_emu::kShflUp, v, delta, 0, PM_EMU_SITE()));
}
__attribute__((noinline)) static double __shfl_xor(double v, int lane_mask, int width = 64) {
    (void)width;
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    const uint64_t r = pm_emu::Collective(pm_emu::kShfl, bits, (pm_emu::LaneId() ^ static_cast<uint32_t>(lane_mask)) & 63u, 0, PM_EMU_SITE());
    double out;
    std::memcpy(&out, &r, 8);
    return out;
}
static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
__attribute__((noinline)) static void __builtin_amdgcn_wave_barrier() { (void)pm_emu::Collective(pm_emu::kWaveBarrier, 0, 0, 0, PM_EMU_SITE()); }
__attribute__((noinline)) static void __syncthreads() { pm_emu::BlockBarrier(PM_EMU_SITE()); }
__attribute__((noinline)) static void __builtin_amdgcn_s_barrier() { pm_emu::BlockBarrier(PM_EMU_SITE()); }

static inline uint32_t __lane_id() { return pm_emu::LaneId(); }
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t add) {
    const uint32_t l = pm_emu::LaneId();
    return add + static_cast<uint32_t>(__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u))));
}
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t add) {
    const uint32_t l = pm_emu::LaneId();
    return add + (l > 32 ? static_cast<uint32_t>(__builtin_popcount(mask & ((1u << (l - 32)) - 1u))) : 0u);
}
#define __builtin_amdgcn_fence(...) ((void)0)
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __threadfence_block() {}
static inline void __threadfence() {}

// ---- scalar helpers ------------------------------------------------------------------------------
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
using std::max;
using std::min;
static inline uint32_t min(uint32_t a, int b) { return std::min<uint32_t>(a, static_cast<uint32_t>(b)); }
static inline unsigned long long wall_clock64() { return 0ull; }

// (lanes run one at a time: plain read-modify-write)
template <typename T> static inline T atomicAdd(T *p, T v) { const T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { const T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicExch(T *p, T v) { const T o = *p; *p = v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { const T o = *p; *p = std::max(o, v); return o; }
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
// (__hip_atomic_* are clang builtins on every target)

// ---- runtime API (host memory stands in for HBM; everything is synchronous) ----------------------
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorUnknown = 999, hipErrorNotReady = 600, hipErrorOutOfMemory = 2 };
typedef struct pm_emu_stream *hipStream_t;
typedef struct pm_emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum : unsigned { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct hipDeviceProp_t {
    char gcnArchName[256];
    int multiProcessorCount;
};

hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int dev);
hipError_t hipSetDevice(int dev);
const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t pm_emu_malloc(void **p, size_t n);
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return pm_emu_malloc(reinterpret_cast<void **>(p), n); }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { return pm_emu_malloc(reinterpret_cast<void **>(p), n); }
static inline hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipFree(void *p);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind k);
hipError_t hipMemset(void *p, int v, size_t n);
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipDeviceSynchronize();

// ---- launches: every workgroup of the grid, one after the other, its lanes as fibers ---------------
namespace pm_emu {
void Launch(dim3 grid, dim3 block, const std::function<void()> &lane_body);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    pm_emu::Launch((grid), (block), [&]() { (kernel)(__VA_ARGS__); })
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ev0, ev1, flags, ...) \
    pm_emu::Launch((grid), (block), [&]() { (kernel)(__VA_ARGS__); })
