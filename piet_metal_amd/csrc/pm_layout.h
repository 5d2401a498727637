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

struct Cmd {  // TestApp/GenTypes.h:430-433
    uint32_t tag;
    uint32_t body[5];
};
static_assert(sizeof(Cmd) == 24, "Cmd is 24 bytes");

}  // namespace pm
