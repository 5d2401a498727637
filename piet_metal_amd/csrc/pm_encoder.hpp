// pm::Encoder -- the piet-metal `Encoder` (src/lib.rs:79-254) as a C++ class.
#pragma once

#include <cstddef>
#include <cstdint>

// Beyond the reference: BeginGroup inside an open group starts a NESTED group (it takes one item
// slot of its parent; EndGroup closes it and writes its bounding box), and Fill takes the
// winding-rule flag the reference reserves.

#include "../../include/piet_metal_amd.h"
#include "pm_layout.h"

namespace pm {

class Encoder {
public:
    enum Status { kOk = 0, kMisuse = 1, kCapacity = 2 };

    Encoder(uint8_t *buf, size_t cap);  // Encoder::new, src/lib.rs:104

    size_t Alloc(size_t size);                       // :114
    void BeginGroup(size_t n_items);                 // :132
    void EndGroup();                                 // :146
    void Circle(double cx, double cy, double r);     // :167
    void Ellipse(double cx, double cy, double rx, double ry);  // extension: Circle item + kCircleEllipse
    void StrokeLine(double x0, double y0, double x1, double y1, float width, uint32_t rgba);  // :177
    void Fill(const double *pts_xy, size_t n, uint32_t rgba, uint32_t flags = 0);             // :195 (+ PietFill.flags)
    // extension: several closed sub-paths under one winding sum (PietFill.flags bit 1, pm_layout.h)
    void FillCompound(const double *pts_xy, const uint32_t *sub_counts, size_t n_sub, uint32_t rgba, uint32_t flags = 0);
    void Polyline(const double *pts_xy, size_t n, uint32_t rgba, float width);                // :209
    // :224 -- returns points_ix, bbox_out = {x0,y0,x1,y1}
    size_t EncodePoints(const double *pts_xy, size_t n, double bbox_out[4]);

    // :122 write_struct -- `len` raw bytes at offset `at` of the buffer (bounds-checked: the Rust slice
    // index panics, here the status turns kCapacity and nothing is written)
    void WriteStruct(size_t at, const void *src, size_t len) { Put(at, src, len); }

    size_t bytes_used() const { return free_space_; }
    size_t bytes_free() const { return free_space_ < cap_ ? cap_ - free_space_ : 0; }
    bool ok() const { return status_ == kOk; }
    int c_status() const {
        return status_ == kOk ? PM_OK : (status_ == kCapacity ? PM_ERR_CAPACITY : PM_ERR_INVALID);
    }

private:
    template <typename Item>
    void AddItem(const Item &item, const ShortBbox &bbox);  // :151
    void Put(size_t at, const void *src, size_t len);       // write_struct, :122

    uint8_t *buf_;
    size_t cap_;
    size_t free_space_ = 0;
    size_t group_count_ = 0;
    size_t group_ix_ = 0;
    size_t group_start_ = 0;
    bool group_open_ = false;
    struct Frame {
        size_t count, ix, start;
    };
    static constexpr int kMaxDepth = 32;
    Frame stack_[kMaxDepth];  // enclosing groups of the one being filled
    int depth_ = 0;
    Status status_ = kOk;
};

int64_t SceneCardioid(uint8_t *buf, size_t cap);
int64_t ScenePathTest(uint8_t *buf, size_t cap);

}  // namespace pm
