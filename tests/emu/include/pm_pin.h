// TEST INFRASTRUCTURE: plain-C++ stand-in for piet_metal_amd/csrc/gfx950/pm_pin.h (register pins
// written as GCN asm constraints) when the kernels run lane by lane on the CPU (tests/emu/).
#pragma once

#include <cstdint>
#include <cstring>

namespace pm {
namespace {

inline uint32_t OpaqueZero() { return 0u; }
inline uint32_t Opaque(uint32_t v) { return v; }
inline uint32_t AtomicAddOneLane(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
inline void PinF32(float &x) {  // (a compiler barrier is all a CPU build needs: no v_fma_mix here)
    asm volatile("" : "+x"(x));
}
inline float OpaqueInfinity() {
    const uint32_t b = 0x7f800000u;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
inline void PinSlot(uint32_t &, float4 &) {}
inline void PinPiece(uint32_t &, uint32_t &, float &, float &, float &, float &, uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &,
                     uint32_t &) {}
inline void PinLoaded8(uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t &) {}

template <uint32_t kLane>
inline void WriteLane(uint32_t &v, uint32_t uniform_value) {
    if (__lane_id() == kLane) v = uniform_value;
}

// hand-overs inside one launch: plain accesses (workgroups run one after the other here)
inline void StoreWT16(uint4 *p, uint4 v) { *p = v; }
inline void StoreWT8(uint2 *p, uint2 v) { *p = v; }
inline void StoreWT4(uint32_t *p, uint32_t v) { *p = v; }
inline uint2 LoadCoherent8(const void *p) { return *static_cast<const uint2 *>(p); }
inline uint32_t LoadCoherent4(const uint32_t *p) { return *p; }
inline uint4 LoadCoherent16(const void *p) { return *static_cast<const uint4 *>(p); }
inline void DrainStores() {}
// (a lane that waits for another wave of its workgroup lets the scheduler run that wave: tests/emu/emu_runtime.cpp)
inline void SleepPoll() { pm_emu::Sleep(); }
inline void SpinPause() { pm_emu::Sleep(); }
// (no clock: a wait counts its own polls -- see PollClock's callers)
inline unsigned long long PollClock() {
    static unsigned long long t = 0;
    return ++t;
}

}  // namespace
}  // namespace pm
