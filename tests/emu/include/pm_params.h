// TEST INFRASTRUCTURE: stand-in for piet_metal_amd/csrc/gfx950/pm_params.h.  On the GPU the kernel
// arguments sit in a VGPR and are read with v_readlane, also from divergent code; a lane-by-lane CPU
// run has no register file another lane could read, so here every lane keeps the struct itself.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace pm {
namespace {

struct ParamRegs {
    FrameParams p;
};

inline ParamRegs LoadParams(const FrameParams &P) {
    ParamRegs r;
    r.p = P;
    return r;
}

template <size_t kOff>
inline uint32_t ParamU32(const ParamRegs &r) {
    static_assert(kOff % 4 == 0 && kOff < sizeof(FrameParams), "field offset");
    uint32_t v;
    std::memcpy(&v, reinterpret_cast<const char *>(&r.p) + kOff, 4);
    return v;
}

template <typename T, size_t kOff>
inline T ParamPtr(const ParamRegs &r) {
    const uint64_t lo = ParamU32<kOff>(r), hi = ParamU32<kOff + 4>(r);
    return reinterpret_cast<T>(lo | (hi << 32));
}

}  // namespace
}  // namespace pm
