// CoarseTile: the tile-level half of tileKernel + TileEncoder for ONE queued tile, run by one wave.
// Shared by pm_coarse_kernel (stand-alone pass, capture variant) and pm_tile_kernel (coarse + fine
// fused per tile, pm_fine.hip).  See pm_kernels_common.h for the decomposition.
#pragma once
#include "pm_kernels_common.h"

namespace pm {
namespace {

constexpr uint32_t kWaveCands = 64;  // candidates handled per pass

// (3 076 bytes: it shares its bytes with the renderer's fragment tables, not with the staged commands -- the first
//  chunk of a tile's list is built INTO the LDS the renderer reads it from, pm_fine.hip, WaveLds.  What only the
//  closing command of an item needs -- colour words, bbox word, backdrop -- stays in the registers of the lane that
//  loaded the candidate and travels by ds_bpermute.)
struct CoarseLds {
    float4 segw[64];          // the first 64 segments of a piece, fetched together with its candidates
    uint32_t htag[kWaveCands];
    uint32_t haux0[kWaveCands];
    uint32_t hrel[kWaveCands];   // relevant segments of the candidate in this tile
    uint32_t hwoff[kWaveCands];  // index of its first relevant segment in the piece
    uint32_t hcnt[kWaveCands];   // stream elements (relevant segments, or 1 pseudo element)
    uint32_t hoff[kWaveCands + 1];
    uint32_t own[64];            // stream position of a round -> candidate that starts there
    uint32_t any[kWaveCands];
};


// What the four waves of a workgroup exchange when they build ONE tile's list together (CoarseTile<..., kPar>):
// a candidate pass whose stream is 65 .. 256 elements long is cut into its rounds of 64, a wave each.
struct CoarseShared {
    uint32_t any_mask[2];    // per candidate of the pass: one of its elements emitted a command
    uint32_t state[4];       // wave 0's {n_pending, list_len, solid_color, rel_done} when the pass begins
    uint4 rec[kWaves];       // per wave: {commands of its round, position of its last opaque Solid + 1, of its last drawing command + 1, that Solid's colour}
};

// Returns the number of commands left in the tile's list for the fine stage (0: nothing to
// interpret -- empty, Bail tile already painted here, or arena overflow).
// Developer timeline (kProf instantiations only): 10 ns ticks per stage of the list building
struct CoarseTicks {
    unsigned long long hdr = 0;    // (unused since the tile's piece arrives in one batch of loads)
    unsigned long long cand = 0;   // piece header + candidates + first segments: loads and scans
    unsigned long long owner = 0;  // round: owners of the stream elements
    unsigned long long scan = 0;   // (unused: binning leaves the tile's segments in paint order)
    unsigned long long seg = 0;    // round: segment fetch + phase-2 tests
    unsigned long long emit = 0;   // round: closing commands, slots, stores
    unsigned long long rounds = 0, records = 0;
};
#define PM_CT_TICK(v) \
    if (kProf) v = wall_clock64()

// lds_chunks != nullptr (the fused tile kernel): the first lds_n * 64 commands of the list are left in LDS ONLY,
// 64 per chunk, chunk k at lds_chunks + k * lds_stride bytes -- the wave's own staged-command area (one chunk), or,
// for a tile the whole workgroup will render, the areas of the workgroup's other waves, idle while this wave
// builds the list (kLdsChunks of them).  The renderer starts from LDS; only what lies beyond goes to the tile's
// list in HBM and is read back chunk by chunk (round 3 wrote every command there and read the single-wave
// tiles' lists straight back: 113 MB each way at config 4).
constexpr uint32_t kLdsChunks = 3;

// kPar (a tile the whole workgroup will render; ALL FOUR waves call, with the same arguments but their own L): the
// longest lists -- the 4K Tiger's longest, 166 commands from 150 stream elements, took one wave 7.5 us to build and
// ended the launch -- are built by the four waves together: every wave loads the pass's candidates, wave w runs the
// phase-2 tests of elements [64 w, 64 w + 64), the waves exchange what they emit (CoarseShared) and each writes its
// own commands at their place in the list.  Passes of at most 64 (or more than 256) elements are wave 0's alone, the
// others walk the pieces without touching them; only wave 0's return value counts.
// kCoh (one launch per frame, pm_frame_kernel): the pieces were written during THIS launch, possibly by a workgroup on another
// XCD -- they are read with agent-scope loads (no L1 line, no line of this XCD's L2 can be older than the piece it holds).
template <bool kCapture, bool kProf = false, bool kPar = false, bool kCoh = false>
__device__ __forceinline__ uint32_t CoarseTile(const FrameParams &P, CoarseLds &L, const uint4 qe,
                                               const uint32_t lane, const uint64_t lanes_below, CoarseTicks *ticks = nullptr,
                                               uint8_t *const lds_chunks = nullptr, const uint32_t lds_stride = 0, const uint32_t lds_n = 0,
                                               CoarseShared *const sh = nullptr) {
    // (kPar instantiations serve single-wave tiles too: sh == nullptr, one copy of the code in the kernel)
    const bool wg = kPar && sh != nullptr;
    const uint32_t pw = wg ? WaveId() : 0u;  // this wave's place among the workgroup's
    const uint32_t lds_cmds = lds_chunks != nullptr ? lds_n * 64u : 0u;  // list positions below this live in LDS
    Cmd *out_cmds = nullptr;  // (set below)
    auto put = [&](uint32_t q, const Cmd &c) {
        if (q < lds_cmds)
            *reinterpret_cast<Cmd *>(lds_chunks + (q >> 6) * lds_stride + (q & 63u) * static_cast<uint32_t>(sizeof(Cmd))) = c;
        else
            out_cmds[q] = c;
    };
    auto quad = [](const uint4 *q) -> uint4 {
        if constexpr (kCoh) return LoadCoherent16(q);
        else return *q;
    };
    auto quad_f = [&](const float4 *q) -> float4 {
        if constexpr (kCoh) {
            const uint4 v = LoadCoherent16(q);
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else {
            return *q;
        }
    };
    unsigned long long tk0 = 0, tk1 = 0;
    // (the entry names the tile by column | row << 16: no division by the width of the grid on the way to its pixels)
    const uint32_t tx = qe.x & 0xffffu, ty_rel = qe.x >> 16;
    const uint32_t tile = ty_rel * P.tiles_x + tx;
    if (qe.y == 0xffffffffu) {  // the command-list arena overflowed (pm_sync re-renders the frame)
        if (lane == 0) P.tile_ncmd[tile] = 0;
        return 0;
    }
    out_cmds = reinterpret_cast<Cmd *>(P.tarena + qe.y);  // this tile's private command slots
    const uint32_t ty = P.row0 + ty_rel;
    const int x0 = static_cast<int>(tx * kTileW);
    const int y0 = static_cast<int>(ty * kTileH);
    const float fx0 = static_cast<float>(x0), fy0 = static_cast<float>(y0);
    const float fx1 = static_cast<float>(x0 + static_cast<int>(kTileW));
    const float fy1 = static_cast<float>(y0 + static_cast<int>(kTileH));

    uint32_t solid_color = 0xffffffffu;  // TileEncoder::solidColor (:74)
    uint32_t n_pending = 0;              // commands written to the tile's list so far
    uint32_t list_len = 0;               // logical list length since tileBegin (capture)

    // The tile's pieces, one per binning record of its strip row (almost always one): the candidates
    // that can emit here and their relevant segments, contiguous and in paint order.
    uint32_t piece = qe.z;
    uint32_t piece_n = qe.w;
    while (piece != 0) {
        PM_CT_TICK(tk0);
        const uint32_t nhit = piece_n & kPieceHitMask, nrel = piece_n >> kPieceHitBits;
        const uint4 *const pc = P.tarena + piece;
        const float4 *const segs = reinterpret_cast<const float4 *>(pc + 1u);  // {header, segments, candidates}
        const uint4 *const cands = pc + 1u + nrel;
        // Header (the next piece), the first 64 candidates and the first 64 segments: all REQUESTED before any of them is
        // looked at, one round trip.  (Written as it reads -- the header through Scalar4, the candidates inside the
        // pass loop -- the compiler waited for the header before it asked for the segments, and for those before the
        // candidates: three round trips in front of every tile's list, 1.5-2 us of its 8-11.)
        uint2 hdr_l;  // (every lane the same address)
        if constexpr (kCoh) hdr_l = LoadCoherent8(pc);
        else hdr_l = *reinterpret_cast<const uint2 *>(pc);
        // (unconditional, from clamped indices -- a piece has at least one candidate, and what a lane beyond the piece's
        //  segments or candidates reads is never looked at: a load under `if (lane < n)` merges with a zero behind it, and
        //  the copy that merge needs waits for the load on the spot)
        const uint32_t li = Opaque(lane);
        const float4 seg_l = quad_f(segs + min(li, nrel));  // (index nrel: the first candidate's quad -- inside the piece)
        const uint4 *const cr0 = cands + 2u * min(li, nhit - 1u);
        const uint4 ca_l = quad(cr0), cb_l = quad(cr0 + 1);
        uint32_t h0 = hdr_l.x, h1 = hdr_l.y, a0 = ca_l.x, a1 = ca_l.y, a2 = ca_l.z, a3 = ca_l.w, b0 = cb_l.x, b1 = cb_l.y, b2 = cb_l.z, b3 = cb_l.w;
        float s0 = seg_l.x, s1 = seg_l.y, s2 = seg_l.z, s3 = seg_l.w;
        PinPiece(h0, h1, s0, s1, s2, s3, a0, a1, a2, a3, b0, b1, b2, b3);
        const float4 seg_first = make_float4(s0, s1, s2, s3);
        const uint4 cand_a = make_uint4(a0, a1, a2, a3), cand_b = make_uint4(b0, b1, b2, b3);
        const uint2 hdr = make_uint2(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(h0))),
                                     static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(h1))));
        // (kPar: a piece whose passes cannot be longer than a round is wave 0's alone)
        const bool piece_shared = wg && nrel + nhit > 64u;
        if (pw != 0u && !piece_shared) {
            piece = hdr.x;
            piece_n = hdr.y;
            continue;
        }
        uint32_t rel_done = 0;  // relevant segments owned by earlier candidate passes
        if (kProf) ticks->records += 1;

        for (uint32_t cb = 0; cb < nhit; cb += kWaveCands) {
            if (cb != 0) PM_CT_TICK(tk0);
            const uint32_t nh = min(kWaveCands, nhit - cb);
            // this lane's candidate: what only an item's closing command needs stays here (fetched by the owner's
            // last element with ds_bpermute, below)
            uint32_t my_rgba = 0, my_aux1 = 0, my_rg = 0, my_ba = 0;
            int my_backdrop = 0;
            if (lane < nh) {
                uint4 a = cand_a, b = cand_b;  // (the first pass's: requested with the piece's header)
                if (cb != 0) {
                    const uint4 *cr = cands + 2u * (cb + lane);
                    a = quad(cr);
                    b = quad(cr + 1);
                }
                L.htag[lane] = a.x & 0xffffu;
                my_rgba = a.y;
                L.haux0[lane] = a.z;
                my_aux1 = a.w;
                const uint32_t rel = b.x & kCtCountMask;
                L.hrel[lane] = rel;
                L.hcnt[lane] = rel ? rel : 1u;  // circle / backdrop-only fill: one pseudo element
                my_rg = b.z;
                my_ba = b.w;
                my_backdrop = static_cast<int>(b.x) >> kCtShift;
                L.any[lane] = 0;
            }
            if (cb == 0) L.segw[lane] = seg_first;
            WaveSync();
            uint32_t stream_len, pass_rel;
            {
                const uint32_t v = (lane < nh) ? L.hcnt[lane] : 0u;
                const uint32_t r = (lane < nh) ? L.hrel[lane] : 0u;
                const uint32_t incl = WaveInclusiveScan(v);
                const uint32_t rincl = WaveInclusiveScan(r);
                if (lane < nh) {
                    L.hoff[lane] = incl - v;
                    L.hwoff[lane] = rel_done + rincl - r;  // first relevant segment of the candidate
                }
                stream_len = WaveLast(incl);
                pass_rel = WaveLast(rincl);
                if (lane == 0) L.hoff[nh] = stream_len;
            }
            WaveSync();

            if (kProf) {
                tk1 = wall_clock64();
                ticks->cand += tk1 - tk0;
            }
            // (uniform over the workgroup: every wave computed the same stream)
            const bool par = piece_shared && stream_len > 64u && stream_len <= 64u * static_cast<uint32_t>(kWaves);
            if (pw != 0u && !par) {  // wave 0's pass
                rel_done += pass_rel;
                continue;
            }
            if (par) {
                if (pw == 0u) {
                    if (lane < 2u) sh->any_mask[lane] = 0u;
                    if (lane == 0u) {
                        sh->state[0] = n_pending;
                        sh->state[1] = list_len;
                        sh->state[2] = solid_color;
                        sh->state[3] = rel_done;
                    }
                }
                __syncthreads();  // (a full barrier: commands of earlier passes that went to HBM are behind every wave too -- a restarted list reuses their slots)
                n_pending = sh->state[0];
                list_len = sh->state[1];
                solid_color = sh->state[2];
                if (pw != 0u) {  // (the segment offsets were computed from this wave's own idea of rel_done)
                    const uint32_t d = sh->state[3] - rel_done;
                    if (lane < nh) L.hwoff[lane] += d;
                    rel_done = sh->state[3];
                    WaveSync();
                }
            }
            auto mark_any = [&](uint32_t cc) {
                if (par) atomicOr(&sh->any_mask[cc >> 5], 1u << (cc & 31u));
                else atomicOr(&L.any[cc], 1u);
            };
            auto get_any = [&](uint32_t cc) -> bool { return par ? ((sh->any_mask[cc >> 5] >> (cc & 31u)) & 1u) != 0u : L.any[cc] != 0u; };
            // ---- stream rounds: phase-2 tests -> ordered commands --------------------------
            // (par: this wave's one round; the owner of the element before it is the last candidate that starts earlier)
            uint32_t own_carry = 0;  // owner of the last element of the previous round
            if (par && pw != 0u) own_carry = static_cast<uint32_t>(__popcll(__ballot(lane < nh && L.hoff[lane < nh ? lane : 0u] < 64u * pw))) - 1u;
            for (uint32_t e0 = par ? 64u * pw : 0u; par ? e0 == 64u * pw : e0 < stream_len; e0 += par ? 0x40000000u : 64u) {
                PM_CT_TICK(tk0);
                const uint32_t e = e0 + lane;
                // Owner of every element without a search: each candidate marks the stream
                // position where it starts, a prefix maximum spreads the marks (owners only
                // grow along the stream).
                L.own[lane] = 0;
                WaveSync();
                if (lane < nh) {
                    const uint32_t p = L.hoff[lane] - e0;
                    if (p < 64u) L.own[p] = lane;
                }
                WaveSync();
                const uint32_t owner = max(WaveInclusiveMax(L.own[lane]), own_carry);
                own_carry = WaveLast(owner);
                uint32_t n_em = 0;   // commands of this stream element (0..2)
                Cmd c0, c1;
                c0.tag = 0; c1.tag = 0;
                bool is_last = false;
                bool draws = false;  // any of this lane's commands clears solidColor
                uint32_t c = 0, ctag = 0;
                if (e < stream_len) c = owner;
                // the owner's colour / bbox words and backdrop, from the lane that loaded the candidate (every lane
                // takes part: a lane that is switched off hands nothing over)
                const uint32_t frgba = static_cast<uint32_t>(__shfl(static_cast<int>(my_rgba), static_cast<int>(c)));
                const uint32_t faux1 = static_cast<uint32_t>(__shfl(static_cast<int>(my_aux1), static_cast<int>(c)));
                const uint32_t frg = static_cast<uint32_t>(__shfl(static_cast<int>(my_rg), static_cast<int>(c)));
                const uint32_t fba = static_cast<uint32_t>(__shfl(static_cast<int>(my_ba), static_cast<int>(c)));
                const int fbackdrop = __shfl(my_backdrop, static_cast<int>(c));
                if (e < stream_len) {
                    const uint32_t k = e - L.hoff[c];
                    is_last = (k + 1 == L.hcnt[c]);
                    ctag = L.htag[c];
                    if (ctag == kItemCircle) {  // :218-222
                        n_em = 1;
                        c0.tag = kCmdCircle;
                        c0.body[0] = frgba;  // CmdCircle.flags (extension: bit 0 = ellipse)
                        c0.body[1] = L.haux0[c];
                        c0.body[2] = faux1;
                        c0.body[3] = 0;
                        c0.body[4] = 0;
                        draws = true;
                    }
                }
                if (kProf) {
                    tk1 = wall_clock64();
                    ticks->owner += tk1 - tk0;
                    ticks->rounds += 1;
                }
                if (kProf) tk0 = tk1;
                if (e < stream_len) {
                    if (ctag == kItemFill && L.hrel[c] == 0) {
                        // backdrop-only fill: nothing to test, the closing command decides
                    } else if (ctag != kItemCircle) {
                        // the element's segment: position hwoff + k of the piece (the first 64 are in LDS already)
                        const uint32_t si = L.hwoff[c] + (e - L.hoff[c]);
                        float4 s;
                        if (si < 64u) s = L.segw[si];
                        else s = quad_f(segs + si);
                        const float a = s.w - s.y;
                        const float b = s.x - s.z;
                        const float cc = -(a * s.x + b * s.y);
                        if (ctag == kItemFill) {  // :302-357
                            const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
                            const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
                            const float left = a * fx0;
                            const float right = a * fx1;
                            const float ytop = fmaxf(fy0, ymin);
                            const float ybot = fminf(fy1, ymax);
                            const float top = b * ytop;
                            const float bot = b * ybot;
                            const float s00 = Sgn(top + left + cc);
                            const float s01 = Sgn(top + right + cc);
                            const float s10 = Sgn(bot + left + cc);
                            const float s11 = Sgn(bot + right + cc);
                            // (the backdrop term of :326-333 was summed by the binning kernel)
                            const bool straddle = Straddles(s00, s01, s10, s11);
                            if (xmin < fx0 && xmax > fx0) {
                                const float tt = (s.x - fx0) / b;
                                const float y_edge = s.y + (s.w - s.y) * tt;  // mix(start.y, end.y, tt)
                                if (y_edge >= fy0 && y_edge < fy1) {
                                    n_em = 2;
                                    c0.tag = kCmdFillEdge;
                                    c0.body[0] = static_cast<uint32_t>(static_cast<int>(s00));
                                    c0.body[1] = __float_as_uint(y_edge);
                                    c0.body[2] = 0; c0.body[3] = 0; c0.body[4] = 0;
                                    c1.tag = kCmdFill;
                                    c1.body[0] = 0;
                                    if (b > 0.0f) {
                                        c1.body[1] = __float_as_uint(s.x); c1.body[2] = __float_as_uint(s.y);
                                        c1.body[3] = __float_as_uint(fx0); c1.body[4] = __float_as_uint(y_edge);
                                    } else {
                                        c1.body[1] = __float_as_uint(fx0); c1.body[2] = __float_as_uint(y_edge);
                                        c1.body[3] = __float_as_uint(s.z); c1.body[4] = __float_as_uint(s.w);
                                    }
                                } else if (straddle) {
                                    n_em = 1;
                                }
                            } else if (straddle && xmin < fx1 && xmax > fx0) {
                                n_em = 1;
                            }
                            if (n_em == 1) {
                                c0.tag = kCmdFill;
                                c0.body[0] = 0;
                                c0.body[1] = __float_as_uint(s.x); c0.body[2] = __float_as_uint(s.y);
                                c0.body[3] = __float_as_uint(s.z); c0.body[4] = __float_as_uint(s.w);
                            }
                            if (n_em) mark_any(c);
                        } else {
                            // Line (:223-247) and Poly phase 2 (:406-440) share the inflated-box test
                            const float width = __uint_as_float(L.haux0[c]);
                            const float hw = 0.5f * width + 0.5f;
                            bool pass = true;
                            if (ctag == kItemPoly) {
                                const float xmin = fminf(s.x, s.z), ymin = fminf(s.y, s.w);
                                const float xmax = fmaxf(s.x, s.z), ymax = fmaxf(s.y, s.w);
                                pass = ymax > fy0 - hw && ymin < fy1 + hw && xmax > fx0 - hw && xmin < fx1 + hw;
                            }
                            if (pass) {
                                const float left = a * (fx0 - hw);
                                const float right = a * (fx1 + hw);
                                const float top = b * (fy0 - hw);
                                const float bot = b * (fy1 + hw);
                                const float s00 = Sgn(top + left + cc);
                                const float s01 = Sgn(top + right + cc);
                                const float s10 = Sgn(bot + left + cc);
                                const float s11 = Sgn(bot + right + cc);
                                pass = Straddles(s00, s01, s10, s11);
                            }
                            if (pass) {
                                n_em = 1;
                                c0.tag = kCmdLine;
                                c0.body[0] = 0;
                                c0.body[1] = __float_as_uint(s.x); c0.body[2] = __float_as_uint(s.y);
                                c0.body[3] = __float_as_uint(s.z); c0.body[4] = __float_as_uint(s.w);
                                draws = true;
                                mark_any(c);
                            }
                        }
                    }
                }
                if (par) LdsBarrier();  // per-candidate accumulators complete (every wave's round)
                else WaveSync();        // ... for elements <= this round
                if (kProf) {
                    tk1 = wall_clock64();
                    ticks->seg += tk1 - tk0;
                }

                // ---- per-item closing command (DrawFill / Solid / Stroke) --------------------
                bool has_fin = false;
                bool opaque_solid = false;
                Cmd fin;
                fin.tag = 0;
                fin.body[0] = fin.body[1] = fin.body[2] = fin.body[3] = fin.body[4] = 0;
                if (is_last) {
                    const uint32_t rg = frg, ba = fba;
                    if (ctag == kItemFill) {  // :359-363
                        const int backdrop = fbackdrop;
                        const uint32_t even_odd = L.haux0[c] & kFillEvenOdd;  // PietFill.flags (extension)
                        // (closing commands are built by the writers pm_layoutgen emits from the layout description)
                        if (get_any(c)) {
                            has_fin = true;
                            fin = gen::ptcl::Cmd_DrawFill_pack(backdrop, frgba, rg, ba, even_odd);
                            draws = true;
                        } else if (even_odd ? (backdrop & 1) != 0 : backdrop != 0) {  // wholly inside: non-zero / odd winding
                            has_fin = true;
                            fin = gen::ptcl::Cmd_Solid_pack(frgba, rg, ba);
                            opaque_solid = (frgba & 0xff000000u) == 0xff000000u;  // :132
                        }
                    } else if (ctag == kItemPoly || ctag == kItemLine) {  // :441-443, :243
                        if (get_any(c)) {
                            has_fin = true;
                            fin = gen::ptcl::Cmd_Stroke_pack(0.5f * __uint_as_float(L.haux0[c]), frgba, rg, ba);
                            draws = true;
                        }
                    }
                }
                const uint32_t lane_total = n_em + (has_fin ? 1u : 0u);  // 0..3

                // ---- wave-wide slots (ballots + popcounts, no scan network) ---------------------
                const uint64_t m0 = __ballot((lane_total & 1u) != 0);
                const uint64_t m1 = __ballot((lane_total & 2u) != 0);
                uint32_t pos = RankBelow(m0) + 2u * RankBelow(m1);  // (mbcnt: no 64-bit mask of the lanes below kept in registers across the tile loop)
                uint32_t round_total = static_cast<uint32_t>(__popcll(m0)) + 2u * static_cast<uint32_t>(__popcll(m1));
                if (!par && round_total == 0) continue;  // uniform
                const uint64_t ms = __ballot(opaque_solid);
                const uint64_t md = __ballot(draws);
                int last_solid = -1, last_draw = -1;
                uint32_t solid_rgba = 0;  // colour of the round's last opaque Solid
                if (ms) {
                    last_solid = static_cast<int>(WaveAtHighest(pos + n_em, ms));
                    solid_rgba = WaveAtHighest(fin.body[0], ms);
                }
                if (md) last_draw = static_cast<int>(WaveAtHighest(pos + lane_total, md)) - 1;
                if (par) {
                    // the four rounds as ONE: positions count from the first wave's first command, the last opaque Solid
                    // and the last drawing command are the last ones of any wave
                    if (lane == 0) sh->rec[pw] = make_uint4(round_total, static_cast<uint32_t>(last_solid + 1), static_cast<uint32_t>(last_draw + 1), solid_rgba);
                    LdsBarrier();
                    uint32_t before = 0, all = 0;
                    last_solid = last_draw = -1;
#pragma unroll
                    for (uint32_t v = 0; v < static_cast<uint32_t>(kWaves); ++v) {
                        const uint4 r = sh->rec[v];
                        if (v < pw) before += r.x;
                        if (r.y) {
                            last_solid = static_cast<int>(all + r.y) - 1;
                            solid_rgba = r.w;
                        }
                        if (r.z) last_draw = static_cast<int>(all + r.z) - 1;
                        all += r.x;
                    }
                    pos += before;
                    round_total = all;
                    if (round_total == 0) continue;  // uniform over the workgroup
                }

                uint32_t base;        // list slot of round position 0 (may be "negative")
                uint32_t first_kept;  // round positions below this are dropped
                if (last_solid >= 0) {
                    // TileEncoder::encodeSolid with an opaque colour (:132-135): dst = tileBegin
                    first_kept = static_cast<uint32_t>(last_solid);
                    n_pending = 0;
                    list_len = 0;
                    base = 0u - first_kept;
                } else {
                    first_kept = 0;
                    base = n_pending;
                }
                {
                    uint32_t p = pos;
                    if (n_em >= 1) {
                        if (p >= first_kept) {
                            put(base + p, c0);
                            if (kCapture) {
                                const uint32_t li = list_len + p - first_kept;
                                if (li < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = c0;
                            }
                        }
                        ++p;
                    }
                    if (n_em == 2) {
                        if (p >= first_kept) {
                            put(base + p, c1);
                            if (kCapture) {
                                const uint32_t li = list_len + p - first_kept;
                                if (li < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = c1;
                            }
                        }
                        ++p;
                    }
                    if (has_fin && p >= first_kept) {
                        put(base + p, fin);
                        if (kCapture) {
                            const uint32_t li = list_len + p - first_kept;
                            if (li < P.dbg_max) {
                                // capture in the reference's layout: no pre-converted colour words
                                Cmd ref = fin;
                                if (fin.tag == kCmdSolid) { ref.body[1] = 0; ref.body[2] = 0; }
                                else { ref.body[2] = 0; ref.body[3] = 0; }
                                P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + li] = ref;
                            }
                        }
                    }
                }
                n_pending = base + round_total;
                list_len += round_total - first_kept;
                if (last_solid >= 0) solid_color = solid_rgba;
                if (last_draw > last_solid) solid_color = 0;  // encodeCircle/Line/Stroke/DrawFill (:81,:90,:99,:124)
                WaveSync();
                // A pass the four waves built together ends with every wave's commands stored: wave 0 may run the next pass (of this
                // piece or the next) alone, and an opaque Solid there restarts the list over the very slots waves 1-3 have just
                // written -- a store of theirs still in flight would land on top of the newer command.  (Commands beyond the LDS
                // chunks went to HBM: the full barrier also waits for those stores.)
                if (par) {
                    if (n_pending > lds_cmds) __syncthreads();
                    else LdsBarrier();
                }
                if (kProf) ticks->emit += wall_clock64() - tk1;
            }
            rel_done += pass_rel;
        }
        piece = hdr.x;
        piece_n = hdr.y;
    }

    // (capture: wave 0's closing record of the list may land on a slot another wave captured a dropped command in)
    if (kCapture && wg) __syncthreads();
    if (pw != 0u) return 0;  // (wave 0 finishes the tile)
    // ---- TileEncoder::end() (:144-151): Bail tiles are finished here (composite :34-44) ----
    if (lane == 0 && lds_chunks == nullptr) {
        P.tile_ncmd[tile] = solid_color ? 0u : n_pending;  // what a stand-alone pm_fine_kernel reads (the fused kernel has the return value)
    }
    if (solid_color != 0) {
        // the tile is one opaque colour, bytes as stored: 64 lanes x 16 B = the whole tile
        const uint32_t ol = Opaque(lane);  // (made here: hoisted out of the tile loop these were spilled)
        const uint32_t pxi = static_cast<uint32_t>(x0) + (ol & 3u) * 4u;
        const uint32_t prow = ol >> 2;
        const uint32_t pyi = static_cast<uint32_t>(y0) + prow;
        if (pyi < P.height && pxi < P.width) {
            uint8_t *dst = P.fb + static_cast<size_t>(ty_rel * kTileH + prow) * P.fb_stride + static_cast<size_t>(pxi) * 4;
            const uint32_t px = StoreOrder(solid_color, P.fb_bgra);
            if (pxi + 4 <= P.width && P.fb_vec16) {
                StorePixels4(dst, make_uint4(px, px, px, px));
            } else {
                for (uint32_t k = 0; k < 4 && pxi + k < P.width; ++k) reinterpret_cast<uint32_t *>(dst)[k] = px;
            }
        }
    }
    if (kCapture && lane == 0) {
        // list as the reference leaves it: {Bail} or cmds + End
        P.dbg_solid[tile] = solid_color;
        Cmd tail;
        tail.body[0] = tail.body[1] = tail.body[2] = tail.body[3] = tail.body[4] = 0;
        if (solid_color != 0) {
            P.dbg_counts[tile] = 1;
            tail.tag = kCmdBail;
            if (P.dbg_max > 0) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max] = tail;
        } else {
            P.dbg_counts[tile] = list_len + 1;
            tail.tag = kCmdEnd;
            if (list_len < P.dbg_max) P.dbg_cmds[static_cast<size_t>(tile) * P.dbg_max + list_len] = tail;
        }
    }
    return solid_color ? 0u : n_pending;
}

#undef PM_CT_TICK

}  // namespace
}  // namespace pm
