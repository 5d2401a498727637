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

}  // namespace
}  // namespace pm
