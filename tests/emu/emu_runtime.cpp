// TEST INFRASTRUCTURE -- see include/hip/hip_runtime.h in this directory.
//
// The lane scheduler and the stand-in runtime API of the CPU emulation build.  A workgroup's lanes
// are fibers (own stacks, hand-written context switch); a lane runs until it reaches a cross-lane
// operation, a workgroup barrier or its end.  When every live lane of a wave is parked at the same
// operation the scheduler computes all results at once (active mask = the lanes that arrived) and
// makes them runnable again.  Lanes parked at DIFFERENT operations are a divergent collective: the
// group at the lowest call site goes first (code is laid out in program order, a branch body comes
// before its join point); PM_EMU_STRICT=1 (meant for the -O0 build, where one source-level operation
// is one call site) aborts instead.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <cstdio>
#include <random>
#include <vector>

namespace pm_emu {

uint3 g_thread_idx, g_block_idx, g_block_dim, g_grid_dim;

namespace {

extern "C" void pm_emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl pm_emu_switch
.type pm_emu_switch,@function
pm_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size pm_emu_switch,.-pm_emu_switch
)");

enum State : int { kRunnable, kWaitWave, kWaitBlock, kDone, kSleeping };

struct Lane {
    void *sp = nullptr;
    char *stack = nullptr;
    State state = kDone;
    Op op = kBallot;
    uint64_t a = 0, b = 0, c = 0, result = 0;
    void *site = nullptr;
    uint32_t tid = 0;
};

constexpr size_t kStackBytes = 256 * 1024;
constexpr uint32_t kMaxLanes = 1024;

Lane g_lanes[kMaxLanes];
uint32_t g_n_lanes = 0;
Lane *g_cur = nullptr;
void *g_sched_sp = nullptr;
const std::function<void()> *g_body = nullptr;
char *g_stacks = nullptr;
long g_divergent = 0, g_site_mismatch = 0;
bool g_strict = false, g_init = false;
std::mt19937 g_rng(12345);
bool g_shuffle = false;

void Yield() { pm_emu_switch(&g_cur->sp, g_sched_sp); }

void LaneMain() {
    (*g_body)();
    g_cur->state = kDone;
    Yield();
    std::abort();  // a finished lane is never resumed
}

void InitOnce() {
    if (g_init) return;
    g_init = true;
    g_stacks = static_cast<char *>(mmap(nullptr, kStackBytes * kMaxLanes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (g_stacks == MAP_FAILED) {
        std::fprintf(stderr, "pm_emu: cannot map lane stacks\n");
        std::abort();
    }
    if (const char *v = std::getenv("PM_EMU_STRICT")) g_strict = std::atoi(v) != 0;
    if (const char *v = std::getenv("PM_EMU_SHUFFLE")) {
        g_shuffle = true;
        g_rng.seed(static_cast<unsigned>(std::atoi(v)));
    }
}

void StartLane(Lane *l, uint32_t tid) {
    l->stack = g_stacks + kStackBytes * tid;
    l->tid = tid;
    l->state = kRunnable;
    // initial frame for pm_emu_switch: six callee-saved registers, then LaneMain as the return
    // address; at LaneMain's entry rsp must be 8 mod 16 (as after a call)
    uintptr_t top = reinterpret_cast<uintptr_t>(l->stack + kStackBytes);
    top &= ~static_cast<uintptr_t>(15);
    uint64_t *sp = reinterpret_cast<uint64_t *>(top);
    *--sp = 0;  // keeps the alignment rule
    *--sp = reinterpret_cast<uint64_t>(&LaneMain);
    for (int k = 0; k < 6; ++k) *--sp = 0;
    l->sp = sp;
}

void Resume(Lane *l) {
    g_cur = l;
    g_thread_idx = uint3{l->tid, 0, 0};
    pm_emu_switch(&g_sched_sp, l->sp);
    g_cur = nullptr;
}

// source lane of a DPP control for lane i, or -1
int DppSource(uint32_t ctrl, int i) {
    const int row = i & ~15, r = i & 15;
    if (ctrl <= 0xff) return (i & ~3) + static_cast<int>((ctrl >> (2 * (i & 3))) & 3u);
    if (ctrl >= 0x101 && ctrl <= 0x10f) {  // row_shl
        const int s = r + static_cast<int>(ctrl & 15u);
        return s < 16 ? row + s : -1;
    }
    if (ctrl >= 0x111 && ctrl <= 0x11f) {  // row_shr
        const int s = r - static_cast<int>(ctrl & 15u);
        return s >= 0 ? row + s : -1;
    }
    if (ctrl >= 0x121 && ctrl <= 0x12f) return row + ((r - static_cast<int>(ctrl & 15u)) & 15);  // row_ror
    if (ctrl == 0x130) return i + 1 < 64 ? i + 1 : -1;  // wave_shl:1
    if (ctrl == 0x138) return i >= 1 ? i - 1 : -1;      // wave_shr:1
    if (ctrl == 0x134) return (i + 1) & 63;             // wave_rol:1
    if (ctrl == 0x13c) return (i - 1) & 63;             // wave_ror:1
    if (ctrl == 0x140) return row + 15 - r;             // row_mirror
    if (ctrl == 0x141) return (i & ~7) + 7 - (i & 7);   // row_half_mirror
    if (ctrl == 0x142) return row >= 16 ? row - 1 : -1; // row_bcast15
    if (ctrl == 0x143) return i >= 32 ? 31 : -1;        // row_bcast31
    std::fprintf(stderr, "pm_emu: DPP control 0x%x not modelled\n", ctrl);
    std::abort();
}

// All lanes in `who` (bit per lane of the wave starting at base) are parked at the same operation.
void Complete(uint32_t base, uint64_t who) {
    Lane *w = g_lanes + base;
    const int first = __builtin_ctzll(who);
    const Op op = w[first].op;
    uint64_t ballot = 0;
    if (op == kBallot)
        for (int i = 0; i < 64; ++i)
            if (((who >> i) & 1u) && w[i].a) ballot |= 1ull << i;
    for (int i = 0; i < 64; ++i) {
        if (!((who >> i) & 1u)) continue;
        Lane &l = w[i];
        switch (op) {
            case kBallot: l.result = ballot; break;
            case kReadFirst: l.result = w[first].a; break;
            case kReadLane: {
                const uint32_t s = static_cast<uint32_t>(l.b) & 63u;
                // (the hardware reads the register of an inactive lane too; here only parked lanes have one)
                l.result = ((who >> s) & 1u) ? w[s].a : 0u;
                break;
            }
            case kDpp: {
                const uint32_t ctrl = static_cast<uint32_t>(l.c);
                const uint32_t row_mask = static_cast<uint32_t>(l.c >> 32) & 15u, bank_mask = static_cast<uint32_t>(l.c >> 36) & 15u;
                const bool bound = ((l.c >> 40) & 1u) != 0;
                uint64_t r = l.b;  // old
                if (((row_mask >> (i >> 4)) & 1u) && ((bank_mask >> ((i >> 2) & 3)) & 1u)) {
                    const int s = DppSource(ctrl, i);
                    if (s >= 0 && ((who >> s) & 1u)) r = w[s].a;
                    else if (bound) r = 0;
                }
                l.result = r;
                break;
            }
            case kShfl: {
                const uint32_t s = static_cast<uint32_t>(l.b) & 63u;
                l.result = ((who >> s) & 1u) ? w[s].a : 0u;
                break;
            }
            case kShflUp: {
                const int s = i - static_cast<int>(l.b);
                l.result = (s >= 0 && ((who >> s) & 1u)) ? w[s].a : l.a;
                break;
            }
            case kWaveBarrier: l.result = 0; break;
            default: std::abort();
        }
    }
    for (int i = 0; i < 64; ++i)
        if ((who >> i) & 1u) w[i].state = kRunnable;
}

void RunBlock() {
    const uint32_t n = g_n_lanes, n_waves = (n + 63) / 64;
    std::vector<uint32_t> order(n_waves);
    for (uint32_t k = 0; k < n_waves; ++k) order[k] = k;
    unsigned long idle_rounds = 0;  // rounds in which nothing happened but sleepers waking up
    for (;;) {
        bool progress = false;
        uint32_t live = 0, at_barrier = 0;
        // lanes that slept (a wave polling for another wave of the workgroup) get their next look now that every
        // other wave has had a turn
        bool woke = false;
        for (uint32_t i = 0; i < n; ++i)
            if (g_lanes[i].state == kSleeping) {
                g_lanes[i].state = kRunnable;
                woke = true;
            }
        if (g_shuffle) std::shuffle(order.begin(), order.end(), g_rng);
        for (uint32_t wk = 0; wk < n_waves; ++wk) {
            const uint32_t base = order[wk] * 64u, cnt = std::min(64u, n - base);
            for (;;) {
                bool ran = false;
                for (uint32_t i = 0; i < cnt; ++i)
                    if (g_lanes[base + i].state == kRunnable) {
                        Resume(&g_lanes[base + i]);
                        ran = true;
                        if (g_lanes[base + i].state != kSleeping) progress = true;
                    }
                // every lane of the wave is parked or done
                uint64_t waiting = 0, blocked = 0;
                for (uint32_t i = 0; i < cnt; ++i) {
                    if (g_lanes[base + i].state == kWaitWave) waiting |= 1ull << i;
                    if (g_lanes[base + i].state == kWaitBlock) blocked |= 1ull << i;
                }
                if (!waiting) break;
                // one operation for all of them?
                const int f = __builtin_ctzll(waiting);
                void *lo_site = g_lanes[base + f].site;
                bool same_kind = true, same_site = true;
                for (uint32_t i = 0; i < cnt; ++i)
                    if ((waiting >> i) & 1u) {
                        if (g_lanes[base + i].op != g_lanes[base + f].op) same_kind = false;
                        if (g_lanes[base + i].site != g_lanes[base + f].site) same_site = false;
                        if (g_lanes[base + i].site < lo_site) lo_site = g_lanes[base + i].site;
                    }
                uint64_t who = waiting;
                // (an optimizing build duplicates call sites -- tail duplication puts one source-level
                //  barrier into both arms of a branch -- so only the strict mode of the -O0 build
                //  insists on equal sites; otherwise equal kinds are taken as one operation)
                if ((g_strict ? !same_site : !same_kind) || blocked) {
                    // divergent collective: the lanes at the lowest call site go first
                    g_divergent += 1;
                    if (g_strict) {
                        std::fprintf(stderr, "pm_emu: divergent cross-lane operation in block %u wave %u (sites differ, kinds %s)\n", g_block_idx.x,
                                     base / 64u, same_kind ? "equal" : "differ");
                        for (uint32_t i = 0; i < cnt; ++i)
                            std::fprintf(stderr, "  lane %u state %d op %d site %p\n", i, g_lanes[base + i].state, g_lanes[base + i].op, g_lanes[base + i].site);
                        std::abort();
                    }
                    who = 0;
                    for (uint32_t i = 0; i < cnt; ++i)
                        if (((waiting >> i) & 1u) && g_lanes[base + i].site == lo_site) who |= 1ull << i;
                }
                Complete(base, who);
                progress = true;
                (void)ran;
            }
            for (uint32_t i = 0; i < cnt; ++i) {
                if (g_lanes[base + i].state != kDone) live += 1;
                if (g_lanes[base + i].state == kWaitBlock) at_barrier += 1;
            }
        }
        if (live == 0) return;
        if (at_barrier == live) {
            for (uint32_t i = 0; i < n; ++i)
                if (g_lanes[i].state == kWaitBlock) g_lanes[i].state = kRunnable;
            continue;
        }
        bool sleepers = woke;
        for (uint32_t i = 0; i < n; ++i)
            if (g_lanes[i].state == kSleeping) sleepers = true;
        if (!progress && sleepers && ++idle_rounds < (1ul << 22)) continue;  // (only sleepers: they look again)
        if (progress) idle_rounds = 0;
        if (!progress) {
            std::fprintf(stderr, "pm_emu: deadlock in block %u: %u live lanes, %u at the workgroup barrier\n", g_block_idx.x, live, at_barrier);
            std::abort();
        }
    }
}

}  // namespace

uint32_t LaneId() { return g_cur->tid & 63u; }

uint64_t Collective(Op op, uint64_t a, uint64_t b, uint64_t c, void *site) {
    Lane *l = g_cur;
    l->op = op;
    l->a = a;
    l->b = b;
    l->c = c;
    l->site = site;
    l->state = kWaitWave;
    Yield();
    return l->result;
}

// A lane that polls for something another wave of its workgroup will do: parked until every other wave has had a turn.
void Sleep() {
    Lane *l = g_cur;
    l->state = kSleeping;
    Yield();
}

void BlockBarrier(void *site) {
    Lane *l = g_cur;
    l->op = kBlockBarrier;
    l->site = site;
    l->state = kWaitBlock;
    Yield();
}

void Launch(dim3 grid, dim3 block, const std::function<void()> &lane_body) {
    InitOnce();
    if (g_cur != nullptr) {
        std::fprintf(stderr, "pm_emu: nested launch\n");
        std::abort();
    }
    const uint32_t n = block.x * block.y * block.z;
    if (n == 0 || n > kMaxLanes || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
        std::fprintf(stderr, "pm_emu: launch shape not modelled\n");
        std::abort();
    }
    g_body = &lane_body;
    g_block_dim = uint3{block.x, 1, 1};
    g_grid_dim = uint3{grid.x, 1, 1};
    for (uint32_t b = 0; b < grid.x; ++b) {
        g_block_idx = uint3{b, 0, 0};
        g_n_lanes = n;
        for (uint32_t t = 0; t < n; ++t) StartLane(&g_lanes[t], t);
        RunBlock();
    }
    g_body = nullptr;
}

}  // namespace pm_emu

// ---- runtime API ---------------------------------------------------------------------------------

struct pm_emu_stream { int id; };
struct pm_emu_event { int id; };

extern "C" long pm_emu_divergent_collectives() { return pm_emu::g_divergent; }

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->gcnArchName, "gfx950:emulated-on-cpu");
    const char *v = std::getenv("PM_EMU_CUS");
    p->multiProcessorCount = v ? std::max(1, std::atoi(v)) : 256;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t pm_emu_malloc(void **p, size_t n) {
    // Arenas are sized for the worst case (hundreds of MB, mostly untouched): map them lazily.
    const size_t bytes = (std::max<size_t>(n, 1) + 4095 + 64) & ~static_cast<size_t>(4095);
    char *m = static_cast<char *>(mmap(nullptr, bytes + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (m == MAP_FAILED) return hipErrorOutOfMemory;
    *reinterpret_cast<size_t *>(m) = bytes + 4096;
    *p = m + 4096;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    char *m = static_cast<char *>(p) - 4096;
    munmap(m, *reinterpret_cast<size_t *>(m));
    return hipSuccess;
}
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t y = 0; y < height; ++y) std::memmove(static_cast<char *>(dst) + y * dpitch, static_cast<const char *>(src) + y * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = new pm_emu_stream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new pm_emu_stream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new pm_emu_event{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new pm_emu_event{0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }  // (no clock: a token duration)
hipError_t hipDeviceSynchronize() { return hipSuccess; }
