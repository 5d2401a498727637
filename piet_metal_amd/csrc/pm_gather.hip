// pm_comm_* / pm_gather: presentation of a frame rendered as tile-row bands on several GPUs
// (SURVEY.md 8b row 2, 8e).  The reference is single-device (one MTLDevice, TestApp/
// PietRenderer.m:27); this is the one place the multi-GPU design exchanges data: every rank's
// band of RGBA8 rows goes to the root over RCCL (xGMI inside a node), straight into its place in
// the final image -- one grouped ncclSend/ncclRecv, i.e. 7 concurrent point-to-point transfers
// into the root's 7 links at 8 GPUs, no ring, no staging copy.
//
// RCCL is bound at run time (dlopen): a host process that already carries an RCCL (PyTorch
// bundles its own librccl.so) keeps exactly one copy, and single-GPU users need none.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/piet_metal_amd.h"
#include "pm_layout.h"

namespace pm {
void SetLastError(const std::string &s);  // pm_context.hip
int ContextDevice(const pm_ctx *c);
int ContextViewport(const pm_ctx *c, uint32_t *width, uint32_t *height, uint32_t *row0, uint32_t *row1);
int ContextLastFrame(pm_ctx *c, const void **fb, size_t *stride, hipStream_t gather_stream);
hipStream_t ContextStream(pm_ctx *c);
}  // namespace pm

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *LoadRccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    // ONE RCCL per process: a host that already carries one (PyTorch maps its own librccl.so, and its process
    // group runs on it) must get THAT copy -- a second runtime next to it would bring its own proxy threads,
    // IPC handles and topology state.  So: the file the process has mapped already (by its path in
    // /proc/self/maps), then the usual names WITHOUT loading (RTLD_NOLOAD answers only for resident objects),
    // and only then a fresh load.  PM_RCCL_LIB overrides all of it.
    const char *forced = std::getenv("PM_RCCL_LIB");
    if (forced && *forced) r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!r.handle && !(forced && *forced)) {
        if (FILE *maps = std::fopen("/proc/self/maps", "r")) {
            char line[1024];
            while (!r.handle && std::fgets(line, sizeof(line), maps)) {
                char *path = std::strchr(line, '/');
                if (!path || !std::strstr(path, "librccl")) continue;
                path[std::strcspn(path, "\n")] = 0;
                r.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            }
            std::fclose(maps);
        }
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) {
        pm::SetLastError("RCCL not found (dlopen librccl.so.1; set PM_RCCL_LIB)");
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(r.handle, name);
        if (!p) ok = false;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        pm::SetLastError("RCCL library lacks a required symbol");
        dlclose(r.handle);
        r.handle = nullptr;
        return nullptr;
    }
    return &r;
}

int RcclFail(const Rccl *r, ncclResult_t e, const char *what) {
    pm::SetLastError(std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
    return PM_ERR_HIP;
}

}  // namespace

struct pm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

static_assert(sizeof(ncclUniqueId) == PM_COMM_ID_BYTES, "pm_comm id = ncclUniqueId");

extern "C" {

int pm_comm_unique_id(uint8_t id[PM_COMM_ID_BYTES]) {
    if (!id) return PM_ERR_INVALID;
    Rccl *r = LoadRccl();
    if (!r) return PM_ERR_NO_DEVICE;
    ncclUniqueId u;
    const ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) return RcclFail(r, e, "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof(u));
    return PM_OK;
}

pm_comm *pm_comm_create(pm_ctx *c, const uint8_t id[PM_COMM_ID_BYTES], int rank, int world, int *err) {
    int dummy;
    if (!err) err = &dummy;
    if (!c || !id || world < 1 || rank < 0 || rank >= world) {
        *err = PM_ERR_INVALID;
        return nullptr;
    }
    Rccl *r = LoadRccl();
    if (!r) {
        *err = PM_ERR_NO_DEVICE;
        return nullptr;
    }
    pm_comm *m = new (std::nothrow) pm_comm();
    if (!m) {
        *err = PM_ERR_CAPACITY;
        return nullptr;
    }
    m->rank = rank;
    m->world = world;
    m->device = pm::ContextDevice(c);
    if (hipSetDevice(m->device) != hipSuccess) {
        *err = PM_ERR_HIP;
        delete m;
        return nullptr;
    }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    const ncclResult_t e = r->CommInitRank(&m->comm, world, u, rank);  // collective: every rank calls it
    if (e != ncclSuccess) {
        *err = RcclFail(r, e, "ncclCommInitRank");
        delete m;
        return nullptr;
    }
    *err = PM_OK;
    return m;
}

void pm_comm_destroy(pm_comm *m) {
    if (!m) return;
    Rccl *r = LoadRccl();
    if (r && m->comm) {
        (void)hipSetDevice(m->device);
        (void)r->CommDestroy(m->comm);
    }
    delete m;
}

int pm_comm_info(pm_comm *m, char *lib_path, size_t lib_path_cap, int *ranks) {
    Rccl *r = LoadRccl();
    if (!r) return PM_ERR_NO_DEVICE;
    if (lib_path && lib_path_cap) {
        // the shared object ncclCommInitRank was bound from: what a bench line or a bug report should name
        Dl_info info;
        const char *name = "";
        if (dladdr(reinterpret_cast<void *>(r->CommInitRank), &info) && info.dli_fname) name = info.dli_fname;
        std::snprintf(lib_path, lib_path_cap, "%s", name);
    }
    if (ranks) {
        *ranks = 0;
        if (m && m->comm) {
            const ncclResult_t e = r->CommCount(m->comm, ranks);
            if (e != ncclSuccess) return RcclFail(r, e, "ncclCommCount");
        }
    }
    return PM_OK;
}

int pm_gather(pm_ctx *c, pm_comm *m, const void *src_band, size_t src_stride, const uint32_t *band_tile_rows, int root, void *dst_image,
              size_t dst_stride, void *hip_stream) {
    if (!c || !m || !band_tile_rows || root < 0 || root >= m->world) return PM_ERR_INVALID;
    Rccl *r = LoadRccl();
    if (!r) return PM_ERR_NO_DEVICE;
    uint32_t width = 0, height = 0, row0 = 0, row1 = 0;
    if (pm::ContextViewport(c, &width, &height, &row0, &row1) != PM_OK) return PM_ERR_INVALID;
    const size_t tight = static_cast<size_t>(width) * 4;
    // (a rank whose own entry is empty -- a sub-band of a band with fewer tile rows than chunks -- sends nothing: any context of the device does)
    if (band_tile_rows[2 * m->rank] < band_tile_rows[2 * m->rank + 1] && (band_tile_rows[2 * m->rank] != row0 || band_tile_rows[2 * m->rank + 1] != row1)) {
        pm::SetLastError("pm_gather: band_tile_rows does not name this context's band");
        return PM_ERR_INVALID;
    }
    if (m->rank == root && (!dst_image || dst_stride != tight)) {
        pm::SetLastError("pm_gather: the root needs a tightly packed width*4 x height destination");
        return PM_ERR_INVALID;
    }
    if (hipSetDevice(m->device) != hipSuccess) return PM_ERR_HIP;
    hipStream_t q = hip_stream ? static_cast<hipStream_t>(hip_stream) : pm::ContextStream(c);
    if (!src_band) {  // this context's last frame; the gather is ordered behind it on q
        const int rs = pm::ContextLastFrame(c, &src_band, &src_stride, q);
        if (rs != PM_OK) return rs;
    }
    if (src_stride != tight) {
        pm::SetLastError("pm_gather: bands must be tightly packed (stride == width * 4)");
        return PM_ERR_INVALID;
    }
    auto band_bytes = [&](int k, size_t *offset) -> size_t {
        const uint64_t y0 = static_cast<uint64_t>(band_tile_rows[2 * k]) * pm::kTileH;
        const uint64_t y1 = std::min<uint64_t>(static_cast<uint64_t>(band_tile_rows[2 * k + 1]) * pm::kTileH, height);
        *offset = static_cast<size_t>(y0 * tight);
        return y1 > y0 ? static_cast<size_t>((y1 - y0) * tight) : 0;
    };
    // one group: the root's receives all progress concurrently, one xGMI link per peer.  The root's OWN
    // band never goes through RCCL: a frame rendered straight into its rows of dst_image (pm_render_to)
    // is already where it belongs, any other source is one device copy on the same stream.
    // (A root-side failure found from here on is reported AFTER the group: the peers have posted their sends
    //  by now, and a root that returned early would leave them waiting on their streams for receives that
    //  never come -- round-3 advisor finding.  The argument checks above are the same on every rank or
    //  concern dst_image itself: after a PM_ERR_INVALID from them the communicator has to be torn down.)
    int late = PM_OK;
    size_t off = 0;
    const size_t mine = band_bytes(m->rank, &off);
    if (m->rank == root && mine) {
        uint8_t *place = static_cast<uint8_t *>(dst_image) + off;
        if (place != static_cast<const uint8_t *>(src_band)) {
            const uint8_t *sb = static_cast<const uint8_t *>(src_band);
            if (sb < place + mine && place < sb + mine) {
                pm::SetLastError("pm_gather: the root's band overlaps its rows of dst_image without being them");
                late = PM_ERR_INVALID;
            } else if (hipMemcpyAsync(place, src_band, mine, hipMemcpyDeviceToDevice, q) != hipSuccess) {
                pm::SetLastError("pm_gather: copying the root's own band failed");
                late = PM_ERR_HIP;
            }
        }
    }
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return RcclFail(r, e, "ncclGroupStart");
    if (m->rank != root && mine && e == ncclSuccess) e = r->Send(src_band, mine, ncclUint8, root, m->comm, q);
    if (m->rank == root) {
        for (int k = 0; k < m->world && e == ncclSuccess; ++k) {
            if (k == root) continue;
            const size_t n = band_bytes(k, &off);
            if (n) e = r->Recv(static_cast<uint8_t *>(dst_image) + off, n, ncclUint8, k, m->comm, q);
        }
    }
    const ncclResult_t eg = r->GroupEnd();
    if (e != ncclSuccess) return RcclFail(r, e, "ncclSend/ncclRecv");
    if (eg != ncclSuccess) return RcclFail(r, eg, "ncclGroupEnd");
    return late;
}

}  // extern "C"
