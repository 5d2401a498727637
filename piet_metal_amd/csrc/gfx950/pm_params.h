// The kernel arguments of pm_bin_kernel, one dword per lane.  The kernel is short of SGPRs: left to
// the compiler, every late use of a FrameParams field becomes its own s_load + s_waitcnt (each a
// 0.2 us scalar round trip, a dozen of them before the first useful load).  Instead the whole
// struct is fetched with ONE vector load at entry and fields are picked out with v_readlane -- no
// memory traffic, no waits.  (tests/emu/ supplies a plain-C++ pm_params.h: reading a lane's
// register from divergent code is not something a lane-by-lane CPU run can reproduce.)
#pragma once

#include <cstddef>
#include <cstdint>
#include <type_traits>

namespace pm {
namespace {

struct ParamRegs {
    uint32_t w[(sizeof(FrameParams) / 4 + 63) / 64];
};

__device__ __forceinline__ ParamRegs LoadParams(const FrameParams &P) {
    static_assert(sizeof(FrameParams) % 4 == 0, "FrameParams is read dword-wise");
    ParamRegs r;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&P);
    const uint32_t lane = __lane_id();
#pragma unroll
    for (uint32_t k = 0; k < sizeof(r.w) / 4; ++k) {
        const uint32_t ix = k * 64u + lane;
        r.w[k] = ix < sizeof(FrameParams) / 4 ? src[ix] : 0u;
    }
    return r;
}

template <size_t kOff>
__device__ __forceinline__ uint32_t ParamU32(const ParamRegs &r) {
    static_assert(kOff % 4 == 0 && kOff < sizeof(FrameParams), "field offset");
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(r.w[kOff / 256]), static_cast<int>((kOff / 4) & 63)));
}

// A pointer rebuilt from two dwords has no address space the compiler could know: every access
// through it would be a FLAT operation, which counts against the LDS counter as well (it might be an
// LDS address) and drags a vmcnt(0) into every LDS-only barrier.  All pointers in FrameParams point to
// device memory: the value is made as a GLOBAL (address space 1) pointer first, and address-space
// inference turns the accesses behind the cast back to generic into global_load / global_store.
template <typename T, size_t kOff>
__device__ __forceinline__ T ParamPtr(const ParamRegs &r) {
    const uint64_t lo = ParamU32<kOff>(r), hi = ParamU32<kOff + 4>(r);
    typedef __attribute__((address_space(1))) std::remove_pointer_t<T> *GlobalPtr;
    return (T)(GlobalPtr)(lo | (hi << 32));
}

}  // namespace
}  // namespace pm
