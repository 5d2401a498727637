// Register pins: the few places where the kernels tell the gfx950 register allocator / instruction
// selector what it must not do.  Kept apart from the kernels because they are written in GCN
// assembler constraints ("v" = a VGPR); tests/emu/ supplies its own pm_pin.h with the same
// functions in plain C++ when it runs the kernels lane by lane on the CPU (test infrastructure).
#pragma once

#include <cstdint>

namespace pm {
namespace {

// Values the compiler must not hoist out of the tile loops (it keeps hoisted copies live across a
// whole tile and then spills them): a zero, and a lane index it cannot see through.
__device__ __forceinline__ uint32_t OpaqueZero() {
    uint32_t v;
    asm volatile("v_mov_b32 %0, 0" : "=v"(v));
    return v;
}
// A returning atomic add issued by ONE lane whose result is looked at LATER.  With a uniform address the compiler's
// atomic optimizer rewrites the operation for a whole wave (one lane adds the sum, v_readfirstlane hands the result
// round) -- and the readfirstlane, with the wait for the atomic's round trip in front of it, sits right behind the
// atomic whatever the source does in between.  An address it cannot prove uniform is left alone.
__device__ __forceinline__ uint32_t AtomicAddOneLane(uint32_t *p, uint32_t v) {
    typedef __attribute__((address_space(1))) uint32_t *GlobalU32;
    unsigned long long a = reinterpret_cast<unsigned long long>(p);
    asm volatile("" : "+v"(a));
    return __hip_atomic_fetch_add((uint32_t *)(GlobalU32)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t Opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}

// A value that is the result of f32 arithmetic, pinned in a register before it is converted or
// compared: instruction selection otherwise folds the producing operation into the consumer
// (v_fma_mixlo_f16 rounds ONCE to binary16 where the source rounds to f32 first).
__device__ __forceinline__ void PinF32(float &x) { asm volatile("" : "+v"(x)); }

// +infinity materialized where it is used (as a literal the compiler hoists copies of it out of
// the tile loop and spills them).
__device__ __forceinline__ float OpaqueInfinity() {
    float v;
    asm volatile("v_mov_b32 %0, 0x7f800000" : "=v"(v));
    return v;
}

// Eight loaded values that must all have arrived before any of them is looked at: keeps the
// compiler from sinking the loads into the branches that use them (one round trip, not three).
__device__ __forceinline__ void PinLoaded8(uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3, uint32_t &a4, uint32_t &a5,
                                           uint32_t &a6, uint32_t &a7) {
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
}

// A tile's piece as the list building first sees it -- header, first segments, first candidates: all arrived
// before anything behind this line is issued (and none of them looked at before all are requested).
__device__ __forceinline__ void PinPiece(uint32_t &h0, uint32_t &h1, float &s0, float &s1, float &s2, float &s3, uint32_t &a0, uint32_t &a1, uint32_t &a2,
                                         uint32_t &a3, uint32_t &b0, uint32_t &b1, uint32_t &b2, uint32_t &b3) {
    // (scalars, not members of the vector structs: an asm operand that is a struct member puts the struct on the stack)
    asm volatile("" : "+v"(h0), "+v"(h1), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2),
                 "+v"(b3));
}

// A loaded slot of the binning record {meta word, segment}: arrived before anything behind this line is issued.
__device__ __forceinline__ void PinSlot(uint32_t &m, float4 &s) { asm volatile("" : "+v"(m), "+v"(s.x), "+v"(s.y), "+v"(s.z), "+v"(s.w)); }

// v_writelane_b32: a wave-uniform value into lane kLane of a VGPR (one instruction; the compiler has
// no builtin for it and would otherwise build `lane == kLane ? s : v` from a compare and a select).
template <uint32_t kLane>
__device__ __forceinline__ void WriteLane(uint32_t &v, uint32_t uniform_value) {
    // (readfirstlane: a value the compiler cannot PROVE uniform would be handed over in a VGPR; it folds
    //  away for values already in SGPRs)
    const uint32_t sv = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(uniform_value)));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(sv), "n"(kLane));
}

// ---- hand-overs inside one launch (pm_frame_kernel) --------------------------------------------------------------------
// The eight XCDs' L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores: what one
// workgroup writes for another is stored WRITE-THROUGH (sc1: the bytes leave the XCD's L2 for memory, the line is dropped)
// and read with agent-scope (sc1) loads, which are served from memory's side of the fabric, never from a stale line.
typedef uint32_t pm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void StoreWT16(uint4 *p, uint4 v) {
    const pm_u32x4 d = {v.x, v.y, v.z, v.w};
    // (an asm store is absent from the compiler's count of outstanding memory operations: its waits for its own loads
    //  only get longer; the data registers must not be rewritten for one more state)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ void StoreWT8(uint2 *p, uint2 v) {
    typedef __attribute__((address_space(1))) unsigned long long *GlobalU64;
    __hip_atomic_store((unsigned long long *)(GlobalU64)p, static_cast<unsigned long long>(v.x) | (static_cast<unsigned long long>(v.y) << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void StoreWT4(uint32_t *p, uint32_t v) {
    typedef __attribute__((address_space(1))) uint32_t *GlobalU32;
    __hip_atomic_store((uint32_t *)(GlobalU32)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 LoadCoherent8(const void *p) {
    typedef __attribute__((address_space(1))) unsigned long long *GlobalU64;
    const unsigned long long x = __hip_atomic_load((unsigned long long *)(GlobalU64)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2(static_cast<uint32_t>(x), static_cast<uint32_t>(x >> 32));
}
__device__ __forceinline__ uint32_t LoadCoherent4(const uint32_t *p) {
    typedef __attribute__((address_space(1))) uint32_t *GlobalU32;
    return __hip_atomic_load((uint32_t *)(GlobalU32)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a 16-byte record as two halves (the data was complete before anybody was told where it is)
__device__ __forceinline__ uint4 LoadCoherent16(const void *p) {
    const uint2 a = LoadCoherent8(p), b = LoadCoherent8(static_cast<const uint8_t *>(p) + 8);
    return make_uint4(a.x, a.y, b.x, b.y);
}
// every store this wave has issued has been acknowledged (also the write-through ones the compiler does not count)
__device__ __forceinline__ void DrainStores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// between two looks at a word somebody else will write
__device__ __forceinline__ void SleepPoll() { __builtin_amdgcn_s_sleep(24); }
// ... between two looks at an LDS word another wave of the workgroup will write (64 cycles)
__device__ __forceinline__ void SpinPause() { __builtin_amdgcn_s_sleep(1); }
__device__ __forceinline__ unsigned long long PollClock() { return wall_clock64(); }

}  // namespace
}  // namespace pm
