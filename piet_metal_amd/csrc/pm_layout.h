// Scene-buffer and command layouts shared by the host encoder and the HIP kernels.
//
// The byte layout is the contract kept from piet-metal so that an encoded scene
// is interchangeable with the reference's:
//   SimpleGroup / ShortBbox / PietItem variants   src/lib.rs:15-77
//   generated readers                             TestApp/GenTypes.h:21-328
//   per-tile command records                      TestApp/GenTypes.h:330-495
//   tile size                                     TestApp/PietShaderTypes.h:17-18
#pragma once

#include <cstddef>
#include <cstdint>

#include "pm_layout_gen.h"  // generated from piet_metal_amd/layout/piet_layout.pgpu by pm_layoutgen

namespace pm {

constexpr uint32_t kTileW = 16;
constexpr uint32_t kTileH = 16;
// Binning granularity: one "strip row" = 16 tiles wide, 1 tile tall.  The
// reference bins with 16x2-tile threadgroups (PietShaderTypes.h:21-22); its
// segment pre-cull depends on that geometry, so the 256x32 px group rectangle
// still appears in the predicates (kGroupW/kGroupH).
constexpr uint32_t kStripTiles = 16;
constexpr uint32_t kGroupW = 256;  // tilerGroupWidth * tileWidth
constexpr uint32_t kGroupH = 32;   // tilerGroupHeight * tileHeight

enum ItemTag : uint32_t {  // src/lib.rs:70-77
    kItemCircle = 1,
    kItemLine = 2,
    kItemFill = 3,
    kItemPoly = 4,
    // Extension (not in the reference, which stops at "This will get more interesting when we
    // have nested groups", src/lib.rs:148): an item that stands for a nested SimpleGroup.
    kItemGroup = 5,
};

// PietFill.flags (src/lib.rs:54, "will be used for winding number rule", TestApp/SceneEncoder.h:44)
constexpr uint32_t kFillEvenOdd = 1u;  // even-odd instead of non-zero (PietRender.metal:539-540)
// Extension (decision D11, "need to deal with subpaths", src/lib.rs:194): PietFill.flags bit 1 = the
// point array holds several closed sub-paths that share one winding sum and one DrawFill.  Every
// sub-path is followed by a separator entry {x = NaN, y = bits of the index of its first point};
// n_points counts points and separators (FillSegmentEnd in pm_kernels_common.h).
constexpr uint32_t kFillCompound = 2u;
constexpr uint32_t kSubpathSeparatorBits = 0x7fc00000u;
// Extension (decision D10): bit 16 of a Circle's item_type word (the reference reads the tag as a
// ushort, PietRender.metal:216) asks for the ellipse inscribed in the item's bbox -- the shading
// PietRender.metal:488-489 leaves as a TODO.  Travels to the tile's list in CmdCircle.flags.
constexpr uint32_t kCircleEllipse = 1u << 16;
constexpr uint32_t kCmdCircleEllipse = 1u;

struct SimpleGroup {  // src/lib.rs:15-20
    uint32_t n_items;
    uint32_t items_ix;
};
static_assert(sizeof(SimpleGroup) == 8, "SimpleGroup is 8 bytes");

struct ShortBbox {  // src/lib.rs:22-24
    uint16_t x0, y0, x1, y1;
};
static_assert(sizeof(ShortBbox) == 8, "ShortBbox is 8 bytes");

struct PietCircle {  // src/lib.rs:33-37
    uint32_t item_type;
};

struct PietStrokeLine {  // src/lib.rs:39-48
    uint32_t item_type;
    uint32_t flags;
    uint32_t rgba;
    float width;
    float start[2];
    float end[2];
};
static_assert(sizeof(PietStrokeLine) == 32, "PietStrokeLine is 32 bytes");
static_assert(offsetof(PietStrokeLine, rgba) == 8 && offsetof(PietStrokeLine, width) == 12 &&
                  offsetof(PietStrokeLine, start) == 16 && offsetof(PietStrokeLine, end) == 24,
              "PietStrokeLine offsets (GenTypes.h:119-138)");

struct PietFill {  // src/lib.rs:50-58
    uint32_t item_type;
    uint32_t flags;
    uint32_t rgba;
    uint32_t n_points;
    uint32_t points_ix;
};
static_assert(sizeof(PietFill) == 20 && offsetof(PietFill, n_points) == 12 &&
                  offsetof(PietFill, points_ix) == 16,
              "PietFill offsets (GenTypes.h:193-209)");

struct PietStrokePolyLine {  // src/lib.rs:60-68
    uint32_t item_type;
    uint32_t rgba;
    float width;
    uint32_t n_points;
    uint32_t points_ix;
};
static_assert(sizeof(PietStrokePolyLine) == 20 && offsetof(PietStrokePolyLine, rgba) == 4 &&
                  offsetof(PietStrokePolyLine, width) == 8 &&
                  offsetof(PietStrokePolyLine, n_points) == 12 &&
                  offsetof(PietStrokePolyLine, points_ix) == 16,
              "PietStrokePolyLine offsets (GenTypes.h:257-273)");

struct PietGroup {  // extension: a nested group in its parent's item list
    uint32_t item_type;  // kItemGroup
    uint32_t flags;      // reserved, 0
    uint32_t group_ix;   // byte offset of the nested SimpleGroup (+ its ShortBbox array, like the root's)
};
static_assert(sizeof(PietGroup) == 12 && offsetof(PietGroup, group_ix) == 8, "PietGroup offsets");

constexpr size_t kItemSize = 32;  // sizeof(union PietItem), src/lib.rs:27-31

enum CmdTag : uint32_t {  // TestApp/GenTypes.h:440-495
    kCmdEnd = 1,
    kCmdCircle = 2,
    kCmdLine = 3,
    kCmdFill = 4,
    kCmdStroke = 5,
    kCmdFillEdge = 6,
    kCmdDrawFill = 7,
    kCmdSolid = 8,
    kCmdBail = 9,
};

using Cmd = gen::ptcl::Cmd;  // {tag, body[5]}, TestApp/GenTypes.h:430-433 -- the generated record
static_assert(sizeof(Cmd) == 24, "Cmd is 24 bytes");

// The hand-written structs above are what the host encoder writes; the generated header is what
// the layout description says.  They must agree, field by field.
namespace layout_check {
using namespace gen::scene;
using namespace gen::ptcl;
static_assert(PIET_ITEM_SIZE == kItemSize && CMD_SIZE == sizeof(Cmd), "record sizes");
static_assert(PietItem_Circle == kItemCircle && PietItem_Line == kItemLine && PietItem_Fill == kItemFill &&
                  PietItem_Poly == kItemPoly && PietItem_Group == kItemGroup, "item tags");
static_assert(Cmd_End == kCmdEnd && Cmd_Circle == kCmdCircle && Cmd_Line == kCmdLine && Cmd_Fill == kCmdFill &&
                  Cmd_Stroke == kCmdStroke && Cmd_FillEdge == kCmdFillEdge && Cmd_DrawFill == kCmdDrawFill &&
                  Cmd_Solid == kCmdSolid && Cmd_Bail == kCmdBail, "command tags");
static_assert(SimpleGroup_items_ix_OFFSET == offsetof(SimpleGroup, items_ix) && SimpleGroup_bbox_OFFSET == sizeof(SimpleGroup), "SimpleGroup");
static_assert(PietStrokeLine_rgba_color_OFFSET == offsetof(PietStrokeLine, rgba) && PietStrokeLine_width_OFFSET == offsetof(PietStrokeLine, width) &&
                  PietStrokeLine_start_OFFSET == offsetof(PietStrokeLine, start) && PietStrokeLine_end_OFFSET == offsetof(PietStrokeLine, end) &&
                  PIET_STROKE_LINE_SIZE == sizeof(PietStrokeLine), "PietStrokeLine");
static_assert(PietFill_flags_OFFSET == offsetof(PietFill, flags) && PietFill_rgba_color_OFFSET == offsetof(PietFill, rgba) &&
                  PietFill_n_points_OFFSET == offsetof(PietFill, n_points) && PietFill_points_ix_OFFSET == offsetof(PietFill, points_ix) &&
                  PIET_FILL_SIZE == sizeof(PietFill), "PietFill");
static_assert(PietStrokePolyLine_rgba_color_OFFSET == offsetof(PietStrokePolyLine, rgba) && PietStrokePolyLine_width_OFFSET == offsetof(PietStrokePolyLine, width) &&
                  PietStrokePolyLine_n_points_OFFSET == offsetof(PietStrokePolyLine, n_points) &&
                  PietStrokePolyLine_points_ix_OFFSET == offsetof(PietStrokePolyLine, points_ix) && PIET_STROKE_POLY_LINE_SIZE == sizeof(PietStrokePolyLine),
              "PietStrokePolyLine");
static_assert(PietGroup_flags_OFFSET == offsetof(PietGroup, flags) && PietGroup_group_ix_OFFSET == offsetof(PietGroup, group_ix) &&
                  PIET_GROUP_SIZE == sizeof(PietGroup), "PietGroup");
// command words as the kernels index them (body[k] sits at byte 4 + 4k)
static_assert(CmdCircle_bbox_OFFSET == 8 && CmdLine_start_OFFSET == 8 && CmdLine_end_OFFSET == 16 && CmdFill_start_OFFSET == 8 &&
                  CmdFill_end_OFFSET == 16, "geometry words: body[1..4]");
static_assert(CmdStroke_halfWidth_OFFSET == 4 && CmdStroke_rgba_color_OFFSET == 8 && CmdStroke_rg_OFFSET == 12 && CmdStroke_ba_OFFSET == 16, "CmdStroke");
static_assert(CmdFillEdge_sign_OFFSET == 4 && CmdFillEdge_y_OFFSET == 8, "CmdFillEdge");
static_assert(CmdDrawFill_backdrop_OFFSET == 4 && CmdDrawFill_rgba_color_OFFSET == 8 && CmdDrawFill_rg_OFFSET == 12 && CmdDrawFill_ba_OFFSET == 16 &&
                  CmdDrawFill_flags_OFFSET == 20, "CmdDrawFill");
static_assert(CmdSolid_rgba_color_OFFSET == 4 && CmdSolid_rg_OFFSET == 8 && CmdSolid_ba_OFFSET == 12, "CmdSolid");
}  // namespace layout_check

}  // namespace pm
