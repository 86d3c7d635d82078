// Internal: what the translation units of libnellie_hip.so (nellie_hip.hip, nellie_label.hip, nellie_markers.hip) share on the host
// side -- the communicator dispatch, the entry-point macros and a few launch helpers.  gfx950 only.
#pragma once
#include <stdarg.h>
#include <stdlib.h>
#include <type_traits>
#include <thread>
#include <mutex>
#include <vector>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "nl_common.h"

#define NL_MASK_SLOTS 2      // cumulative h_mask bit planes (ping-pong between consecutive scales)
#define NL_VERSION "nellie_amd-hip 0.1.0 (gfx950)"
#define SCAN_CHUNK 4096      // elements per workgroup of the exclusive scans (label_voxels.inc)

// What the entry points call: RCCL's names, dispatched per communicator -- a communicator created from a loopback id
// (nl_comm_loopback_id) lives in loopback.inc, every other one is RCCL's (dlopen()ed on first use).  Defined in nellie_hip.hip.
struct CommApi {
    std::atomic<int> n_real{0};
    ncclResult_t GetUniqueId(ncclUniqueId *id);
    ncclResult_t CommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank);
    ncclResult_t CommDestroy(ncclComm_t comm);
    const char *GetErrorString(ncclResult_t r);
    ncclResult_t GroupStart();
    ncclResult_t GroupEnd();
    ncclResult_t Send(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st);
    ncclResult_t Recv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st);
    ncclResult_t AllReduce(const void *src, void *dst, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st);
    ncclResult_t AllGather(const void *src, void *dst, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st);
    ncclResult_t Broadcast(const void *src, void *dst, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t st);
};
CommApi &rccl();

#include "device_math.inc"
#include "convert.inc"
#include "gauss_launch.h"

// 2-D images: 256 columns per workgroup in x, rows by a stride loop in the kernel (~1024 workgroups = four per CU: the kernels end
// in one atomic per workgroup on one word, ~10 ns each -- with 4096 workgroups they were 40 of vesselness2d_kernel's 60 us at 2048^2)
static inline dim3 grid2d_rows(i64 nx, i64 ny) {
    const i64 gx = (nx + 255) / 256;
    i64 gy = (1024 + gx - 1) / gx;
    if (gy > ny) gy = ny;
    if (gy < 1) gy = 1;
    return dim3((unsigned)gx, (unsigned)gy, 1);
}
static inline unsigned int grid1d(i64 n, int block = 256, i64 cap = 256 * 32) {
    i64 g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned int)g;
}

// Zeroing a few counter words between kernels with a one-wave kernel of our own instead of the runtime's fill path
// (~35 of these per frame).  `bytes` is a multiple of 4.  (static: every translation unit has its own copy)
static __global__ void zero_words_kernel(unsigned int *p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
static inline hipError_t zero_small(void *p, size_t bytes, hipStream_t st) {
    zero_words_kernel<<<1, 256, 0, st>>>((unsigned int *)p, (int)(bytes / 4));
    return hipGetLastError();
}

static float *gauss_cur(const nl_ctx *c) { return c->gauss_ext ? c->gauss_ext : c->f[c->i_gauss]; }
static VolGeom geom(const nl_ctx *c) {
    VolGeom v{c->nzl, c->ny, c->nx, c->gz0, c->gnz};
    // 128-element chunks read 128 + 2R elements for 128 outputs; 256 halves the excess (about 1 % of the Gaussian passes at 1024^3,
    // within run-to-run noise) but also halves the number of workgroups, so only where those are plentiful
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("NELLIE_GM_CHUNK"); forced = e ? atoi(e) : 0; }
    // (a 2-D image has one plane: 32-row chunks, or its Gaussian passes run on a handful of workgroups)
    v.chunk = forced > 0 ? forced : (c->two_d ? 32 : (c->n >= ((i64)1 << 29) ? 256 : 128));
    return v;
}

static size_t dtype_size(int dt) {
    switch (dt) {
        case NL_U8: case NL_I8: return 1;
        case NL_U16: case NL_I16: return 2;
        case NL_U32: case NL_I32: case NL_F32: return 4;
        case NL_F64: case NL_U64: case NL_I64: return 8;
    }
    return 0;
}

#define NL_ENTER(c)                                                    \
    if (!(c)) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");   \
    ++(c)->epoch;                                                      \
    NL_HIP(hipSetDevice((c)->device));

// Entry points of the copy threads of nellie_amd/streaming.py (nl_input_load_async, nl_outputs_fetch_async, nl_outputs_wait):
// they run CONCURRENTLY with the compute thread's calls on the same context, touch only the copy streams, the input slots
// and the staging buffers, and therefore leave `epoch` (the compute state's version) alone.
#define NL_ENTER_IO(c)                                                 \
    if (!(c)) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");   \
    NL_HIP(hipSetDevice((c)->device));

// nl_mask_volume_fused leaves the support of the Frangi frame (the opened mask, 1 bit/voxel) behind; nl_label_run may
// use it to skip the 98 % of the frame that is zero -- but only if nothing else ran in between.  Every entry point
// bumps `epoch`; the few that read the frame without touching it or the mask planes carry the validity forward.
#define NL_KEEP_SUPPORT(c) if ((c)->support_epoch + 1 == (c)->epoch.load()) (c)->support_epoch = (c)->epoch.load();
// the same for the label bits nl_label_run leaves in m[1] (read by nl_markers_begin): carried over entry points that write neither the
// label volume nor the bit planes
#define NL_KEEP_LABBITS(c) if ((c)->labbits_epoch + 1 == (c)->epoch.load()) (c)->labbits_epoch = (c)->epoch.load();

// Orders the main stream after whatever is still running on the side stream (the resolve kernel of the previous
// scale).  Called by every entry point that touches the vesselness volume, the mask planes or the queue.
#define NL_JOIN_SIDE(c)                                                            \
    if ((c)->side_pending) {                                                       \
        NL_HIP(hipStreamWaitEvent((c)->stream, (c)->ev_side, 0));                  \
        (c)->side_pending = 0;                                                     \
    }

#define NL_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) { c->comm_poisoned = 1; return nl_fail(err, errlen, NL_ECOMM, "%s: %s", #expr, rccl().GetErrorString(r_)); } \
    } while (0)


// ---- reductions across the ranks, on the device (nl_comm_fuse) ---------------------------------------------------------
// With a communicator in "fused" mode the sampling / statistics entry points below finish with the GLOBAL value: the RCCL
// collective sits on the context stream between the kernels, so a threshold costs one host round trip instead of one per
// pass plus one per host-level all-reduce.  Every rank must make the same calls in the same order (they do: the path is SPMD).
static inline bool fused(const nl_ctx *c) { return c->comm && c->fuse_reduce; }

// defined in nellie_hip.hip, used by the other units
int upload_convert(nl_ctx *c, const void *host, int dtype, float *dst, i64 count, char *err, size_t errlen);
int fetch_counted(nl_ctx *c, const float *stage, const unsigned int *d_n, i64 max_count, float *out, i64 cap, int64_t *n, char *err, size_t errlen);
int store_planes(nl_ctx *c, const void *dev_base, void *host, size_t elem, int64_t z0, int64_t z1, char *err, size_t errlen);
int64_t vq_alloc_entries(int64_t nzl, int64_t ny, int64_t nx);
bool gyx_tiled();
// kernels of one unit launched from another
void nl_launch_threshold_pack(unsigned int grid, hipStream_t st, const float *f, const unsigned long long *support, unsigned long long *bits,
                              int has_thr, float thr, int nx, i64 nrows, int wpr, const float *thr_dev);      // nellie_label.hip
void nl_launch_pack_labels(unsigned int grid, hipStream_t st, const int *lab, unsigned long long *bits, int nx, i64 nrows, int wpr);   // nellie_markers.hip
