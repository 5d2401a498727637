/* TEST INFRASTRUCTURE: a stand-in for librccl.so that pm_gather.hip can bind through PM_RCCL_LIB on a
 * box without GPUs (tests/test_dist_cpu.py, together with the CPU emulation of tests/emu, where "device"
 * pointers are host pointers).  The eight entry points pm_gather.hip looks up, with RCCL's signatures:
 * point-to-point messages travel as files in the directory PM_MOCK_RCCL_DIR names (one per (source,
 * destination, sequence number)); receives posted inside a group complete at ncclGroupEnd, sends at once
 * -- enough to exercise the argument, offset and grouping logic of pm_comm_* / pm_gather across real
 * processes.  Nothing here is RCCL, nothing here is measured. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef int ncclResult_t; /* ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 */
typedef int ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct mock_comm {
    int rank, world;
    char dir[96];
    unsigned send_seq[64], recv_seq[64];
};
typedef struct mock_comm *ncclComm_t;
typedef void *hipStream_t;

struct pending { void *buf; size_t bytes; int peer; ncclComm_t comm; };
static struct pending g_pending[256];
static int g_npending = 0, g_group = 0;

static size_t type_size(ncclDataType_t t) { return (t == 0 || t == 1) ? 1 : (t == 2 || t == 3 || t == 7) ? 4 : 8; }

const char *ncclGetErrorString(ncclResult_t e) { return e == 0 ? "no error" : "mock rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    const char *dir = getenv("PM_MOCK_RCCL_DIR");
    if (!dir || strlen(dir) >= 96) return 2;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "%s", dir);
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return 4;
    /* fault injection (tests/test_dist_cpu.py): PM_MOCK_RCCL_FAIL_INIT = "all", or one rank -- the id has travelled, the library is
     * loaded, and the communicator cannot be made (ncclSystemError): what a job must survive with ONE answer on every rank */
    {
        const char *f = getenv("PM_MOCK_RCCL_FAIL_INIT");
        if (f && *f && (strcmp(f, "all") == 0 || atoi(f) == rank)) return 2;
    }
    struct mock_comm *c = (struct mock_comm *)calloc(1, sizeof(*c));
    if (!c) return 2;
    c->rank = rank;
    c->world = world;
    id.internal[95] = 0;
    snprintf(c->dir, sizeof(c->dir), "%s", id.internal);
    *comm = c;
    return 0;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return 4;
    *count = ((const struct mock_comm *)comm)->world;
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    free(comm);
    return 0;
}

static ncclResult_t do_recv(const struct pending *p) {
    char path[192];
    snprintf(path, sizeof(path), "%s/msg_%d_%d_%u.bin", p->comm->dir, p->peer, p->comm->rank, p->comm->recv_seq[p->peer]);
    for (int tries = 0; tries < 30000; ++tries) { /* 30 s */
        FILE *f = fopen(path, "rb");
        if (f) {
            const size_t n = fread(p->buf, 1, p->bytes, f);
            fseek(f, 0, SEEK_END);
            const long total = ftell(f);
            fclose(f);
            unlink(path);
            p->comm->recv_seq[p->peer] += 1;
            return (n == p->bytes && (size_t)total == p->bytes) ? 0 : 4; /* a size mismatch is an argument error */
        }
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, NULL);
    }
    return 2;
}

ncclResult_t ncclGroupStart(void) {
    g_group += 1;
    return 0;
}

ncclResult_t ncclGroupEnd(void) {
    if (g_group <= 0) return 5;
    g_group -= 1;
    if (g_group) return 0;
    ncclResult_t r = 0;
    for (int i = 0; i < g_npending; ++i) {
        const ncclResult_t e = do_recv(&g_pending[i]);
        if (e && !r) r = e;
    }
    g_npending = 0;
    return r;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    (void)s;
    if (!comm || peer < 0 || peer >= comm->world) return 4;
    char tmp[200], path[192];
    snprintf(path, sizeof(path), "%s/msg_%d_%d_%u.bin", comm->dir, comm->rank, peer, comm->send_seq[peer]);
    snprintf(tmp, sizeof(tmp), "%s.tmp", path);
    FILE *f = fopen(tmp, "wb");
    if (!f) return 2;
    const size_t bytes = count * type_size(t);
    const size_t n = fwrite(buf, 1, bytes, f);
    fclose(f);
    if (n != bytes || rename(tmp, path) != 0) return 2;
    comm->send_seq[peer] += 1;
    return 0;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    (void)s;
    if (!comm || peer < 0 || peer >= comm->world) return 4;
    struct pending p = {buf, count * type_size(t), peer, comm};
    if (g_group > 0) {
        if (g_npending >= 256) return 3;
        g_pending[g_npending++] = p;
        return 0;
    }
    return do_recv(&p);
}
