// libnellie_hip.so -- hand-written HIP for gfx950 (MI355X): Nellie's Filter -> Label hot path.
// C-ABI in include/nellie_amd.h.  Compile with -ffp-contract=off: every float operation
// below is meant to round exactly where numpy/scipy round.
#include <stdarg.h>
#include <stdlib.h>
#include <type_traits>
#include <thread>
#include <vector>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "nl_common.h"

// RCCL is loaded on first use (dlopen) instead of being linked: librccl.so is ~570 MB and would be paged in by every
// single-GPU process that merely loads this library.
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    bool ok = false;
};
static RcclApi &rccl_real() {
    static RcclApi api;
    if (!api.handle) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) { api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.handle) break; }
        if (api.handle) {
#define NL_SYM(F) api.F = (decltype(api.F))dlsym(api.handle, "nccl" #F)
            NL_SYM(GetUniqueId); NL_SYM(CommInitRank); NL_SYM(CommDestroy); NL_SYM(GetErrorString); NL_SYM(GroupStart);
            NL_SYM(GroupEnd); NL_SYM(Send); NL_SYM(Recv); NL_SYM(AllReduce); NL_SYM(Broadcast); NL_SYM(AllGather);
#undef NL_SYM
            api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GetErrorString && api.GroupStart &&
                     api.GroupEnd && api.Send && api.Recv && api.AllReduce && api.Broadcast && api.AllGather;
        }
    }
    return api;
}

#include "loopback.inc"

// What the entry points call: the same names, dispatched per communicator -- a communicator created from a loopback id
// (nl_comm_loopback_id) lives in loopback.inc, every other one is RCCL's.  librccl.so is only loaded when a real id is asked
// for or used.
struct CommApi {
    std::atomic<int> n_real{0};
    static ncclResult_t missing() { return (ncclResult_t)lb::kMissing; }
    ncclResult_t GetUniqueId(ncclUniqueId *id) { return rccl_real().ok ? rccl_real().GetUniqueId(id) : missing(); }
    ncclResult_t CommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
        if (lb::is_loopback_id(id.internal)) return lb::comm_init(comm, world, id.internal, rank);
        if (!rccl_real().ok) return missing();
        const ncclResult_t r = rccl_real().CommInitRank(comm, world, id, rank);
        if (r == ncclSuccess) ++n_real;
        return r;
    }
    ncclResult_t CommDestroy(ncclComm_t comm) {
        if (lb::is_ours(comm)) return lb::comm_destroy(comm);
        if (!rccl_real().ok) return missing();
        --n_real;
        return rccl_real().CommDestroy(comm);
    }
    const char *GetErrorString(ncclResult_t r) {
        if ((int)r == lb::kMissing) return "librccl.so could not be loaded";
        if (rccl_real().handle && rccl_real().ok) return rccl_real().GetErrorString(r);
        switch (r) {
            case ncclInvalidArgument: return "invalid argument (loopback transport)";
            case ncclSystemError: return "rendezvous timed out or a peer failed (loopback transport)";
            case ncclUnhandledCudaError: return "HIP error (loopback transport)";
            default: return "error (loopback transport)";
        }
    }
    ncclResult_t GroupStart() {
        lb::group_start();
        return n_real.load() > 0 ? rccl_real().GroupStart() : ncclSuccess;
    }
    ncclResult_t GroupEnd() {
        const ncclResult_t r = lb::group_end();
        const ncclResult_t q = n_real.load() > 0 ? rccl_real().GroupEnd() : ncclSuccess;
        return r != ncclSuccess ? r : q;
    }
    ncclResult_t Send(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
        if (lb::is_ours(comm)) return lb::submit(lb::Op{0, buf, nullptr, count, dt, ncclSum, peer, (lb::Comm *)comm, st});
        return rccl_real().Send(buf, count, dt, peer, comm, st);
    }
    ncclResult_t Recv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
        if (lb::is_ours(comm)) return lb::submit(lb::Op{1, nullptr, buf, count, dt, ncclSum, peer, (lb::Comm *)comm, st});
        return rccl_real().Recv(buf, count, dt, peer, comm, st);
    }
    ncclResult_t AllReduce(const void *src, void *dst, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
        if (lb::is_ours(comm)) return lb::submit(lb::Op{2, src, dst, count, dt, op, -1, (lb::Comm *)comm, st});
        return rccl_real().AllReduce(src, dst, count, dt, op, comm, st);
    }
    ncclResult_t AllGather(const void *src, void *dst, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st) {
        if (lb::is_ours(comm)) return lb::submit(lb::Op{3, src, dst, count, dt, ncclSum, -1, (lb::Comm *)comm, st});
        return rccl_real().AllGather(src, dst, count, dt, comm, st);
    }
    ncclResult_t Broadcast(const void *src, void *dst, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t st) {
        if (lb::is_ours(comm)) return lb::submit(lb::Op{4, src, dst, count, dt, ncclSum, root, (lb::Comm *)comm, st});
        return rccl_real().Broadcast(src, dst, count, dt, root, comm, st);
    }
};
static CommApi &rccl() { static CommApi api; return api; }

#define NL_MASK_SLOTS 2      // cumulative h_mask bit planes (ping-pong between consecutive scales)
#define NL_VERSION "nellie_amd-hip 0.1.0 (gfx950)"

#include "device_math.inc"
#include "convert.inc"
#include "gauss.inc"
#include "sampling.inc"
#include "hessian.inc"
#include "hessian_pair.inc"
#include "hv_launch.h"
#include "filter2d.inc"
#include "markers.inc"
#include "label_voxels.inc"
#include "label_runs.inc"
#include "pack_out.inc"
#include "network.inc"
#include "thresholds.inc"
#include "chain.inc"
#include "percentile.inc"

// =================================================================================================
// host side: context, launch helpers, C-ABI
// =================================================================================================
// workgroups of the lattice reductions (range, histogram): every workgroup ends with atomics on the same few words, which
// retire ~10 ns apart -- 1024 workgroups spent 10-30 us on that alone (a 1e6-point gather is not longer); NELLIE_SAMPLE_GRID
// NELLIE_CHAIN_UNFUSED_SAMPLING=1: the chain's first round as two separate range + histogram sequences (A/B, tests)
static bool chain_unfused_sampling() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("NELLIE_CHAIN_UNFUSED_SAMPLING"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
static i64 sample_grid_cap() {
    static i64 v = 0;
    if (!v) { const char *e = getenv("NELLIE_SAMPLE_GRID"); v = (e && atoll(e) > 0) ? atoll(e) : 256; }
    return v;
}
// 2-D images: 256 columns per workgroup in x, rows by a stride loop in the kernel (~4096 workgroups: the statistics kernels end
// in one set of atomics per workgroup)
static inline dim3 grid2d_rows(i64 nx, i64 ny) {
    const i64 gx = (nx + 255) / 256;
    i64 gy = (4096 + gx - 1) / gx;
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
// (~35 of these per frame).  `bytes` is a multiple of 4.
__global__ void zero_words_kernel(unsigned int *p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
static inline hipError_t zero_small(void *p, size_t bytes, hipStream_t st) {
    zero_words_kernel<<<1, 256, 0, st>>>((unsigned int *)p, (int)(bytes / 4));
    return hipGetLastError();
}

// fused Y+X Gaussian: tiled / register-blocked X pass (default) or the row-at-a-time kernel (NELLIE_GYX_TILE=0)
static bool gyx_tiled() {
    static int on = -1;
    if (on < 0) { const char *e = getenv("NELLIE_GYX_TILE"); on = (e && !atoi(e)) ? 0 : 1; }
    return on != 0;
}
// grid of the queue kernels: NELLIE_RESOLVE_GRID caps it (waves then walk several regions each).  Measured at 512 x 1024 x 1024
// (ms/step of the resolve group): 1792 workgroups 2.77, 2048 2.69, 4096 2.67, 8192 2.79, 16384 (rounds 1-2) and more 3.16 --
// a wave that walks 8-16 regions amortises its start-up and evens out the regions' very different entry counts
static unsigned resolve_grid(unsigned blocks) {
    static long cap = -1;
    if (cap < 0) { const char *e = getenv("NELLIE_RESOLVE_GRID"); cap = e ? atol(e) : 8192; }
    return (cap > 0 && (unsigned)cap < blocks) ? (unsigned)cap : blocks;
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
// tile height of the Hessian kernels (experiment knob; 8 -> 512-thread workgroups, 16 -> 1024)
static int hm_ty() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("NELLIE_HM_TY"); v = (e && atoi(e) == 16) ? 16 : 8; }
    return v;
}
// Hessian walk: two voxels per thread (hessian_pair.inc), tile 2*RS rows; NELLIE_HV_RS = 8 (default) / 4, 0 = the
// one-voxel kernel of hessian.inc
static int hv_rs_env() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("NELLIE_HV_RS"); v = e ? atoi(e) : 8; if (v != 0 && v != 8 && v != 16) v = 8; }
    return v;
}
// the pair kernel addresses the planes of a Z chunk through one buffer resource (32-bit byte offsets)
static int hv_rs(const nl_ctx *c) { return ((i64)(HM_ZCHUNK + 4) * c->ny * c->nx * 4 < ((i64)1 << 32)) ? hv_rs_env() : 0; }
static HessP hessp(const nl_ctx *c);
static HessDv<1> hessdv_fast(const nl_ctx *c) { return hessdv_fast(hessp(c)); }      // (dv_* and the HessP forms: hv_launch.h)
static HessDv<0> hessdv_exact(const nl_ctx *c) { return hessdv_exact(hessp(c)); }
// the pair walk (its own translation unit, hv_launch.h): division variant as proven for this context's divisors
static int hv_fastv(const nl_ctx *c) { return c->fast_div2 ? 2 : (c->fast_div ? 1 : 0); }
// Exhaustive proof that the 3-instruction division is exact for the six divisors in use.
static int check_fast_div(nl_ctx *c, char *err, size_t errlen) {
    const float ds[6] = {c->hz, c->hy, c->hx, c->hz2, c->hy2, c->hx2};
    unsigned int *bad = (unsigned int *)c->d_small + 64;
    NL_HIP(zero_small(bad, 12 * 4, c->stream));
    for (int k = 0; k < 6; ++k) {
        divcheck_kernel<1><<<(1u << 23) / 256, 256, 0, c->stream>>>(dv_fast(ds[k]), ds[k], bad + k);
        divcheck_kernel<2><<<(1u << 23) / 256, 256, 0, c->stream>>>(dv_two(ds[k]), ds[k], bad + 6 + k);
    }
    NL_CHECK_LAUNCH();
    unsigned int *h = (unsigned int *)c->h_small + 64;
    NL_HIP(hipMemcpyAsync(h, bad, 12 * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    c->fast_div = 1; c->fast_div2 = 1;
    for (int k = 0; k < 6; ++k) {
        const bool normal = ds[k] > 1e-30f && ds[k] < 1e30f;
        if (h[k] || !normal) c->fast_div = 0;
        if (h[6 + k] || !normal) c->fast_div2 = 0;
    }
    // NELLIE_EXACT_DIV=1: the float64 path everywhere; =3: at most the three-instruction sequence (A/B of the two-instruction one)
    const char *e = getenv("NELLIE_EXACT_DIV");
    if (e && atoi(e) == 1) { c->fast_div = 0; c->fast_div2 = 0; }
    if (e && atoi(e) == 3) c->fast_div2 = 0;
    return NL_OK;
}
static HessP hessp(const nl_ctx *c) { return HessP{c->hz, c->hy, c->hx, c->hz2, c->hy2, c->hx2}; }

static size_t dtype_size(int dt) {
    switch (dt) {
        case NL_U8: case NL_I8: return 1;
        case NL_U16: case NL_I16: return 2;
        case NL_U32: case NL_I32: case NL_F32: return 4;
        case NL_F64: case NL_U64: case NL_I64: return 8;
    }
    return 0;
}

extern "C" const char *nl_version(void) { return NL_VERSION; }

extern "C" int nl_device_count(int *count, char *err, size_t errlen) {
    if (!count) return nl_fail(err, errlen, NL_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return nl_fail(err, errlen, NL_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return NL_OK;
}

extern "C" int nl_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes, char *err, size_t errlen) {
    NL_HIP(hipSetDevice(device));
    size_t f = 0, t = 0;
    NL_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return NL_OK;
}

extern "C" int nl_device_name(int device, char *name, size_t namelen, char *err, size_t errlen) {
    hipDeviceProp_t p;
    NL_HIP(hipGetDeviceProperties(&p, device));
    if (name && namelen) snprintf(name, namelen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return NL_OK;
}

// Global eigen queue: every wave of the vesselness kernel owns HM_REGION entries (32 bytes each), so one launch
// needs (padded plane) x (Z chunks x HM_ZCHUNK) entries.  A launch covers as many Z chunks as fit 2^28 entries
// (8 GiB); smaller volumes go in one launch.  NELLIE_VQ_CAP (entries) is a test knob to force several launches.
static int64_t vq_padded_plane(int64_t ny, int64_t nx) { return ((nx + HM_TX - 1) / HM_TX) * HM_TX * (((ny + 31) / 32) * 32); }
static int64_t vq_chunks(int64_t nzl, int64_t ny, int64_t nx) {
    static int64_t lim = 0;
    if (!lim) { const char *e = getenv("NELLIE_VQ_CAP"); lim = (e && atoll(e) > 0) ? atoll(e) : ((int64_t)1 << 28); }
    const int64_t per_chunk = vq_padded_plane(ny, nx) * HM_ZCHUNK;
    int64_t chunks = lim / per_chunk;
    const int64_t need = (nzl + HM_ZCHUNK - 1) / HM_ZCHUNK;
    if (chunks > need) chunks = need;
    return chunks < 1 ? 1 : chunks;
}
static int64_t vq_entries(int64_t nzl, int64_t ny, int64_t nx) { return vq_chunks(nzl, ny, nx) * vq_padded_plane(ny, nx) * HM_ZCHUNK; }
// One-pass (MODE 2) vesselness needs every region of the whole slab at once, HM_SPEC_CAP entries each; it is
// offered when that fits NELLIE_SPEC_MAX_GB (default 64) GiB.  Returns 0 when it does not.
static int64_t vq_spec_regions(int64_t nzl, int64_t ny, int64_t nx) {
    return vq_padded_plane(ny, nx) / 64 * ((nzl + HM_ZCHUNK - 1) / HM_ZCHUNK);
}
static int64_t vq_spec_entries(int64_t nzl, int64_t ny, int64_t nx) {
    static int64_t lim = -1;
    if (lim < 0) { const char *e = getenv("NELLIE_SPEC_MAX_GB"); lim = (e ? atoll(e) : 64) << 30; }
    const int64_t ent = vq_spec_regions(nzl, ny, nx) * HM_SPEC_CAP;
    return ent * 32 <= lim ? ent : 0;
}
static int64_t vq_alloc_entries(int64_t nzl, int64_t ny, int64_t nx) {
    const int64_t a = vq_entries(nzl, ny, nx), b = vq_spec_entries(nzl, ny, nx);
    return a > b ? a : b;
}
static int64_t vq_alloc_regions(int64_t nzl, int64_t ny, int64_t nx) {
    const int64_t a = vq_entries(nzl, ny, nx) / HM_REGION, b = vq_spec_entries(nzl, ny, nx) ? vq_spec_regions(nzl, ny, nx) : 0;
    return a > b ? a : b;
}

extern "C" int64_t nl_ctx_bytes(int64_t nz_local, int64_t ny, int64_t nx) {
    const int64_t n = nz_local * ny * nx;
    const int64_t qe = vq_alloc_entries(nz_local, ny, nx);
    return n * (4 * 4 + 3) + qe * 32 + vq_alloc_regions(nz_local, ny, nx) * 4 + (1 << 16) + ((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) * 4;
}

static void comm_release(nl_ctx *c, void *comm, int role);      // RCCL communicators go back to a per-process pool (see nl_comm_init)
extern "C" int nl_ctx_destroy(nl_ctx *c) {
    if (!c) return NL_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &kv : c->prof) for (auto &r : kv.second) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto &r : c->prof_pool) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (int k = 0; k < 4; ++k) if (c->f[k]) hipFree(c->f[k]);
    for (int k = 0; k < 3; ++k) if (c->m[k]) hipFree(c->m[k]);
    if (c->d_small) hipFree(c->d_small);
    if (c->d_input && !c->input_borrowed) hipFree(c->d_input);
    for (int k = 0; k < 2; ++k) { if (c->d_in_slot[k]) hipFree(c->d_in_slot[k]); if (c->ev_in[k]) hipEventDestroy(c->ev_in[k]); }
    if (c->d_stage_fr) hipFree(c->d_stage_fr);
    if (c->d_stage_lab) hipFree(c->d_stage_lab);
    if (c->ev_staged) hipEventDestroy(c->ev_staged);
    if (c->ev_fetched) hipEventDestroy(c->ev_fetched);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->ev_side) hipEventDestroy(c->ev_side);
    if (c->ev_main) hipEventDestroy(c->ev_main);
    if (c->ev_ahead) hipEventDestroy(c->ev_ahead);
    if (c->copy_in) hipStreamDestroy(c->copy_in);
    if (c->copy_out) hipStreamDestroy(c->copy_out);
    if (c->d_blk) hipFree(c->d_blk);
    for (int k = 0; k < 4; ++k) if (c->d_2d[k]) hipFree(c->d_2d[k]);
    if (c->d_fsq_cache) hipFree(c->d_fsq_cache);
    if (c->d_vq) hipFree(c->d_vq);
    if (c->mk_scratch) hipFree(c->mk_scratch);
    if (c->d_vq_count) hipFree(c->d_vq_count);
    if (c->d_rows) hipFree(c->d_rows);
    if (c->d_ag) hipFree(c->d_ag);
    if (c->d_sl) hipFree(c->d_sl);
    if (c->d_pct) hipFree(c->d_pct);
    if (c->h_pct) hipHostFree(c->h_pct);
    if (c->h_sl) hipHostFree(c->h_sl);
    if (c->d_pack) hipFree(c->d_pack);
    if (c->h_ag) hipHostFree(c->h_ag);
    if (c->gbits[0]) hipFree(c->gbits[0]);
    if (c->gbits[1]) hipFree(c->gbits[1]);
    if (c->grows) hipFree(c->grows);
    if (c->h_small) hipHostFree(c->h_small);
    if (c->h_prefix) hipHostFree(c->h_prefix);
    if (c->t0) hipEventDestroy(c->t0);
    if (c->t1) hipEventDestroy(c->t1);
    if (c->xstream) { hipStreamSynchronize(c->xstream); hipStreamDestroy(c->xstream); }
    if (c->ev_x_main) hipEventDestroy(c->ev_x_main);
    if (c->ev_x_done) hipEventDestroy(c->ev_x_done);
    if (c->d_chain) hipFree(c->d_chain);
    if (c->h_chain) hipHostFree(c->h_chain);
    if (c->ev_chain) hipEventDestroy(c->ev_chain);
    comm_release(c, c->comm2, 2);
    comm_release(c, c->comm, 1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return NL_OK;
}

extern "C" int nl_ctx_create(nl_ctx **out, int device, int64_t nzl, int64_t ny, int64_t nx,
                             int64_t gz0, int64_t gnz, int64_t own_lo, int64_t own_hi, char *err, size_t errlen) {
    if (!out) return nl_fail(err, errlen, NL_EINVAL, "out is NULL");
    *out = nullptr;
    if (nzl < 1 || ny < 1 || nx < 1) return nl_fail(err, errlen, NL_EINVAL, "empty volume (%lld,%lld,%lld)", (i64)nzl, (i64)ny, (i64)nx);
    if (gz0 < 0 || gz0 + nzl > gnz) return nl_fail(err, errlen, NL_EINVAL, "slab [%lld,%lld) outside the global volume of %lld planes", (i64)gz0, (i64)(gz0 + nzl), (i64)gnz);
    if (own_lo < 0 || own_hi > nzl || own_lo >= own_hi) return nl_fail(err, errlen, NL_EINVAL, "bad owned range [%lld,%lld)", (i64)own_lo, (i64)own_hi);
    const i64 n = (i64)nzl * ny * nx;
    if (n >= ((i64)1 << 31)) return nl_fail(err, errlen, NL_EINVAL, "local slab of %lld voxels exceeds the int32 label index range; shard over Z", n);
    if (ny > 65535 || nzl > 65535) return nl_fail(err, errlen, NL_EINVAL, "Y and local Z extents must be <= 65535");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return nl_fail(err, errlen, NL_ENODEV, "GPU backend requested but no HIP device is available (%s)", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return nl_fail(err, errlen, NL_ENODEV, "GPU backend requested but device %d does not exist (%d devices)", device, ndev);
    NL_HIP(hipSetDevice(device));
    nl_ctx *c = new nl_ctx();
    c->device = device; c->nzl = nzl; c->ny = ny; c->nx = nx; c->gz0 = gz0; c->gnz = gnz;
    c->own_lo = own_lo; c->own_hi = own_hi; c->n = n;
    int rc = NL_OK;
    auto alloc = [&](void **p, size_t bytes) -> bool {
        hipError_t ee = hipMalloc(p, bytes);
        if (ee != hipSuccess) {
            rc = nl_fail(err, errlen, ee == hipErrorOutOfMemory ? NL_ENOMEM : NL_EHIP,
                         "hipMalloc(%zu bytes): %s%s", bytes, hipGetErrorString(ee), ee == hipErrorOutOfMemory ? " [out of memory]" : "");
            return false;
        }
        return true;
    };
    bool ok = true;
    for (int k = 0; k < 4 && ok; ++k) ok = alloc((void **)&c->f[k], (size_t)n * 4);
    // m[0] doubles as the cumulative bit mask of Filter: one 64-bit word per 64 x-voxels of a row
    const size_t mask_words = (size_t)NL_MASK_SLOTS * nzl * ny * ((nx + 63) / 64);
    const size_t plane_words_bytes = (size_t)nzl * ny * ((nx + 63) / 64) * 8;     // one bit plane
    for (int k = 0; k < 3 && ok; ++k) {
        size_t bytes = (size_t)n;
        if (k == 0 && mask_words * 8 > bytes) bytes = mask_words * 8;
        if (plane_words_bytes > bytes) bytes = plane_words_bytes;
        ok = alloc((void **)&c->m[k], bytes);
    }
    if (ok) ok = alloc((void **)&c->d_rows, ((size_t)nzl * ny + 2) * 2 * 4);
    if (ok) ok = alloc(&c->d_small, 1 << 16);
    c->vq_chunks = (int)vq_chunks(nzl, ny, nx);
    {
        const size_t qe = (size_t)vq_alloc_entries(nzl, ny, nx);
        if (ok) ok = alloc((void **)&c->d_vq, qe * 32);
        if (ok) ok = alloc((void **)&c->d_vq_count, (size_t)vq_alloc_regions(nzl, ny, nx) * 4);
        c->spec_ok = vq_spec_entries(nzl, ny, nx) > 0;
    }
    c->blk_cap = (n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
    if (ok) ok = alloc(&c->d_blk, (size_t)c->blk_cap * 4);
    if (ok && hipHostMalloc(&c->h_small, 1 << 16, hipHostMallocDefault) != hipSuccess) {
        rc = nl_fail(err, errlen, NL_ENOMEM, "hipHostMalloc failed [out of memory]"); ok = false;
    }
    // side stream: the compute-bound resolve kernel runs beside the memory-bound Gaussian of the next scale; NELLIE_SIDE_PRIO
    // (-1 high, 0 normal, 1 low) chooses which of the two the dispatcher serves first
    int side_prio = -1;
    { const char *e = getenv("NELLIE_SIDE_PRIO"); if (e) side_prio = atoi(e); }
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // lo = numerically largest = lowest priority
    const int side_p = side_prio < 0 ? prio_hi : (side_prio > 0 ? prio_lo : (prio_lo + prio_hi) / 2);
    if (ok && (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, side_p) != hipSuccess ||
               hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&c->ev_ahead, hipEventDisableTiming) != hipSuccess)) {
        rc = nl_fail(err, errlen, NL_EHIP, "stream/event creation failed"); ok = false;
    }
    if (ok && (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
               hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess)) {
        rc = nl_fail(err, errlen, NL_EHIP, "stream/event creation failed"); ok = false;
    }
    if (!ok) { nl_ctx_destroy(c); return rc; }
    *out = c;
    return NL_OK;
}

extern "C" int nl_sync(nl_ctx *c, char *err, size_t errlen) {
    if (!c) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");
    NL_HIP(hipSetDevice(c->device));
    NL_HIP(hipStreamSynchronize(c->stream));
    NL_HIP(hipStreamSynchronize(c->side));
    if (c->xstream) NL_HIP(hipStreamSynchronize(c->xstream));
    return NL_OK;
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

// Orders the main stream after whatever is still running on the side stream (the resolve kernel of the previous
// scale).  Called by every entry point that touches the vesselness volume, the mask planes or the queue.
#define NL_JOIN_SIDE(c)                                                            \
    if ((c)->side_pending) {                                                       \
        NL_HIP(hipStreamWaitEvent((c)->stream, (c)->ev_side, 0));                  \
        (c)->side_pending = 0;                                                     \
    }

static int upload_convert(nl_ctx *c, const void *host, int dtype, float *dst, i64 count, char *err, size_t errlen) {
    const size_t es = dtype_size(dtype);
    if (!es) return nl_fail(err, errlen, NL_EINVAL, "unsupported dtype code %d", dtype);
    if (dtype == NL_F32) {
        NL_HIP(hipMemcpyAsync(dst, host, (size_t)count * 4, hipMemcpyHostToDevice, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
        return NL_OK;
    }
    // stage raw bytes in free float volumes (2 consecutive volumes cover 8-byte types)
    void *raw = nullptr;
    bool own = false;
    // find a free f[] buffer that is not dst's buffer
    for (int k = 0; k < 4 && !raw; ++k) {
        const bool contains = (dst >= c->f[k] && dst < c->f[k] + c->n);
        if (!contains && k != c->i_vmax && es <= 4) raw = c->f[k];
    }
    if (!raw) { NL_HIP(hipMalloc(&raw, (size_t)count * es)); own = true; }
    NL_HIP(hipMemcpyAsync(raw, host, (size_t)count * es, hipMemcpyHostToDevice, c->stream));
    const unsigned int g = grid1d(count);
    switch (dtype) {
        case NL_U8: convert_kernel<uint8_t><<<g, 256, 0, c->stream>>>((const uint8_t *)raw, dst, count); break;
        case NL_I8: convert_kernel<int8_t><<<g, 256, 0, c->stream>>>((const int8_t *)raw, dst, count); break;
        case NL_U16: convert_kernel<uint16_t><<<g, 256, 0, c->stream>>>((const uint16_t *)raw, dst, count); break;
        case NL_I16: convert_kernel<int16_t><<<g, 256, 0, c->stream>>>((const int16_t *)raw, dst, count); break;
        case NL_U32: convert_kernel<uint32_t><<<g, 256, 0, c->stream>>>((const uint32_t *)raw, dst, count); break;
        case NL_I32: convert_kernel<int32_t><<<g, 256, 0, c->stream>>>((const int32_t *)raw, dst, count); break;
        case NL_F64: convert_kernel<double><<<g, 256, 0, c->stream>>>((const double *)raw, dst, count); break;
        case NL_U64: convert_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)raw, dst, count); break;
        case NL_I64: convert_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)raw, dst, count); break;
    }
    NL_CHECK_LAUNCH();
    NL_HIP(hipStreamSynchronize(c->stream));
    if (own) hipFree(raw);
    return NL_OK;
}

extern "C" int nl_filter_load(nl_ctx *c, const void *host, int dtype, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_JOIN_SIDE(c);
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    c->i_gauss = 0; c->i_vmax = 3; c->i_labels = -1; c->frangi_ready = 0;
    c->gauss_ext = nullptr;
    c->ahead_pending = 0;
    c->fsq_cache_valid = 0;
    const i64 plane = c->ny * c->nx;
    int rc = upload_convert(c, host, dtype, c->f[0] + z0 * plane, (z1 - z0) * plane, err, errlen);
    if (rc) return rc;
    // vesselness = zeros, masks = ones (filtering.py:807-808): the first evaluated scale zeroes the vesselness
    // volume and starts the cumulative mask; nl_filter_finish zeroes whatever the masks reject
    c->mask_slots_used = 0;
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// Keep the raw frame resident in HBM (any dtype) so that a timed region can start from device memory.
extern "C" int nl_input_load(nl_ctx *c, const void *host, int dtype, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    const size_t es = dtype_size(dtype);
    if (!es) return nl_fail(err, errlen, NL_EINVAL, "unsupported dtype code %d", dtype);
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    if (c->d_input && c->input_borrowed) { c->d_input = nullptr; c->input_borrowed = 0; }
    if (c->d_input && c->input_dtype != dtype) { hipFree(c->d_input); c->d_input = nullptr; }
    if (!c->d_input) NL_HIP(hipMalloc(&c->d_input, (size_t)c->n * es));
    c->input_dtype = dtype;
    const i64 plane = c->ny * c->nx;
    NL_HIP(hipMemcpyAsync((char *)c->d_input + (size_t)z0 * plane * es, host, (size_t)(z1 - z0) * plane * es, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// frame = xp.asarray(resident input, dtype=float32); vesselness = 0; masks = 1.  Asynchronous.
extern "C" int nl_filter_begin(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_JOIN_SIDE(c);
    if (!c->d_input) return nl_fail(err, errlen, NL_ESTATE, "nl_filter_begin before nl_input_load");
    c->i_gauss = 0; c->i_vmax = 3; c->i_labels = -1; c->frangi_ready = 0;
    c->mask_slots_used = 0;
    c->gauss_ext = nullptr;
    c->ahead_pending = 0;
    c->fsq_cache_valid = 0;
    if (c->input_dtype == NL_F32 && !getenv("NELLIE_COPY_INPUT")) {
        // float32 frames are used where they lie: the cascade never writes its source (ping-pong volumes), so the
        // first Gaussian pass reads the resident input directly (the reference's gauss = frame view, filtering.py:811)
        c->gauss_ext = (float *)c->d_input;
        return NL_OK;
    }
    ProfScope ps(c, "load");
    const unsigned int g = grid1d(c->n);
    switch (c->input_dtype) {
        case NL_U8: convert_kernel<uint8_t><<<g, 256, 0, c->stream>>>((const uint8_t *)c->d_input, c->f[0], c->n); break;
        case NL_I8: convert_kernel<int8_t><<<g, 256, 0, c->stream>>>((const int8_t *)c->d_input, c->f[0], c->n); break;
        case NL_U16: convert_kernel<uint16_t><<<g, 256, 0, c->stream>>>((const uint16_t *)c->d_input, c->f[0], c->n); break;
        case NL_I16: convert_kernel<int16_t><<<g, 256, 0, c->stream>>>((const int16_t *)c->d_input, c->f[0], c->n); break;
        case NL_U32: convert_kernel<uint32_t><<<g, 256, 0, c->stream>>>((const uint32_t *)c->d_input, c->f[0], c->n); break;
        case NL_I32: convert_kernel<int32_t><<<g, 256, 0, c->stream>>>((const int32_t *)c->d_input, c->f[0], c->n); break;
        case NL_F32: convert_kernel<float><<<g, 256, 0, c->stream>>>((const float *)c->d_input, c->f[0], c->n); break;
        case NL_F64: convert_kernel<double><<<g, 256, 0, c->stream>>>((const double *)c->d_input, c->f[0], c->n); break;
        case NL_U64: convert_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)c->d_input, c->f[0], c->n); break;
        case NL_I64: convert_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)c->d_input, c->f[0], c->n); break;
    }
    NL_CHECK_LAUNCH();
    c->mask_slots_used = 0;
    return NL_OK;
}

static int fill_gw(GaussW &gw, const double *w, int r, char *err, size_t errlen) {
    if (r < 0 || r > NL_MAX_RADIUS) return nl_fail(err, errlen, NL_EINVAL, "Gaussian radius %d outside [0,%d]", r, NL_MAX_RADIUS);
    gw.r = r;
    for (int k = 0; k <= r; ++k) gw.w[k] = w[r + k];   // w[] has 2r+1 entries centred at r (symmetric)
    return NL_OK;
}

static bool gm_z2() {
    static int on = -1;
    if (on < 0) { const char *e = getenv("NELLIE_GM_Z2"); on = (e && !atoi(e)) ? 0 : 1; }
    return on != 0;
}
template <int AXIS, int R>
static void launch_gauss_march(nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
    GaussWS ws;
    for (int k = 0; k <= GM_MAX_R; ++k) ws.w[k] = k <= R ? gw.w[k] : 0.0;
    dim3 grid;
    if (AXIS == 0) grid = dim3((unsigned)((c->nx + 63) / 64), (unsigned)((c->ny + 3) / 4), (unsigned)((z1 - z0 + v.chunk - 1) / v.chunk));
    else grid = dim3((unsigned)((c->nx + 63) / 64), (unsigned)((z1 - z0 + 3) / 4), (unsigned)((c->ny + v.chunk - 1) / v.chunk));
    if (AXIS == 0 && R <= 6 && (c->nx & 1) == 0 && ((size_t)src & 7) == 0 && ((size_t)dst & 7) == 0 && gm_z2()) {
        grid.x = (unsigned)((c->nx / 2 + 63) / 64);              // two columns per thread (float2 accesses): see gauss_march_z2_kernel
        gauss_march_z2_kernel<(R <= 6 ? R : 1)><<<grid, 256, 0, c->stream>>>(src, dst, v, z0, z1, ws);
        return;
    }
    gauss_march_kernel<AXIS, R><<<grid, 256, 0, c->stream>>>(src, dst, v, z0, z1, ws);
}
template <int R>
static void launch_gauss_x(nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
    GaussWS ws;
    for (int k = 0; k <= GM_MAX_R; ++k) ws.w[k] = k <= R ? gw.w[k] : 0.0;
    const dim3 grid((unsigned)((c->nx + GX_SEG - 1) / GX_SEG), (unsigned)c->ny, (unsigned)(z1 - z0));
    gauss_x_kernel<R><<<grid, 256, 0, c->stream>>>(src, dst, v, z0, z1, ws, (c->nx % 4 == 0) ? 1 : 0, 0);
}
// returns false when the radius has no specialised kernel
template <int AXIS>
static bool launch_gauss_fast(nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
#define NL_GCASE(RR)                                                                         \
    case RR:                                                                                 \
        if (AXIS == 2) launch_gauss_x<RR>(c, src, dst, v, z0, z1, gw);                       \
        else launch_gauss_march<(AXIS == 2 ? 0 : AXIS), RR>(c, src, dst, v, z0, z1, gw);     \
        return true;
    // the marching kernels reflect at most once: the radius must not exceed the line length
    const i64 n_line = AXIS == 0 ? c->gnz : (AXIS == 1 ? c->ny : c->nx);
    if (AXIS != 2 && gw.r > n_line) return false;
    switch (gw.r) {
        NL_GCASE(1) NL_GCASE(2) NL_GCASE(3) NL_GCASE(4) NL_GCASE(5) NL_GCASE(6) NL_GCASE(7) NL_GCASE(8)
        NL_GCASE(9) NL_GCASE(10) NL_GCASE(11) NL_GCASE(12)
        default: return false;
    }
#undef NL_GCASE
}

extern "C" int nl_gauss_step(nl_ctx *c, const double *wz, int rz, const double *wy, int ry, const double *wx, int rx,
                             int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->ahead_pending) return nl_fail(err, errlen, NL_ESTATE, "nl_gauss_step while a step enqueued ahead is uncommitted");
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    if (c->halo_pending) {                       // ghost planes of the source volume still travelling (nl_halo_exchange_at, async)
        NL_HIP(hipStreamWaitEvent(c->stream, c->ev_x_done, 0));
        c->halo_pending = 0;
    }
    const VolGeom v = geom(c);
    // three ping-pong volumes f[0..2]: the source of a pass is dead once the pass has run,
    // so "the next one" is always a legal destination
    const dim3 blk(256, 1, 1);
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)(z1 - z0));
    int src = c->i_gauss;
    const float *srcp = gauss_cur(c);       // the source of the next pass (the borrowed input before the first one)
    GaussW gw;
    int rc;
    if (wz) {
        if ((rc = fill_gw(gw, wz, rz, err, errlen))) return rc;
        // every tap must land inside the local slab unless it reflects at a true face
        if ((z0 - rz < 0 && c->gz0 > 0) || (z1 - 1 + rz >= c->nzl && c->gz0 + c->nzl < c->gnz))
            return nl_fail(err, errlen, NL_EINVAL, "Z pass of radius %d on planes [%lld,%lld) reaches outside the local slab", rz, (i64)z0, (i64)z1);
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_z");
        if (!launch_gauss_fast<0>(c, srcp, c->f[dst], v, z0, z1, gw))
            gauss_axis_kernel<0><<<grid, blk, 0, c->stream>>>(srcp, c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
    }
    bool fused_yx = false;
    if (wy && wx && ry == rx && ry >= 1 && ry <= GM_MAX_R && ry <= c->ny && !getenv("NELLIE_NO_FUSED_YX")) {
        GaussW gy, gx;
        if ((rc = fill_gw(gy, wy, ry, err, errlen))) return rc;
        if ((rc = fill_gw(gx, wx, rx, err, errlen))) return rc;
        GaussWS wsy, wsx;
        for (int k = 0; k <= GM_MAX_R; ++k) { wsy.w[k] = k <= ry ? gy.w[k] : 0.0; wsx.w[k] = k <= rx ? gx.w[k] : 0.0; }
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_yx");
        const dim3 g2((unsigned)((c->nx + GYX_COLS - 1) / GYX_COLS), (unsigned)((c->ny + v.chunk - 1) / v.chunk), (unsigned)(z1 - z0));
        const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
        if (gyx_tiled()) {
            switch (ry) {
#define NL_YX(RR) case RR: gauss_yx_tile_kernel<RR, false><<<g2.x * g2.y * g2.z, GYX_THREADS, 0, c->stream>>>(srcp, c->f[dst], v, z0, z1, wsy, wsx, vec4, (int)g2.x, (int)g2.y); break;
                NL_YX(1) NL_YX(2) NL_YX(3) NL_YX(4) NL_YX(5) NL_YX(6) NL_YX(7) NL_YX(8) NL_YX(9) NL_YX(10) NL_YX(11) NL_YX(12)
#undef NL_YX
            }
        } else {
            switch (ry) {
#define NL_YX(RR) case RR: gauss_yx_kernel<RR, false><<<g2, GYX_THREADS, 0, c->stream>>>(srcp, c->f[dst], v, z0, z1, wsy, wsx); break;
                NL_YX(1) NL_YX(2) NL_YX(3) NL_YX(4) NL_YX(5) NL_YX(6) NL_YX(7) NL_YX(8) NL_YX(9) NL_YX(10) NL_YX(11) NL_YX(12)
#undef NL_YX
            }
        }
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
        fused_yx = true;
    }
    if (wy && !fused_yx) {
        if ((rc = fill_gw(gw, wy, ry, err, errlen))) return rc;
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_y");
        if (!launch_gauss_fast<1>(c, srcp, c->f[dst], v, z0, z1, gw))
            gauss_axis_kernel<1><<<grid, blk, 0, c->stream>>>(srcp, c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
    }
    if (wx && !fused_yx) {
        if ((rc = fill_gw(gw, wx, rx, err, errlen))) return rc;
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_x");
        if (!launch_gauss_fast<2>(c, srcp, c->f[dst], v, z0, z1, gw))
            gauss_axis_kernel<2><<<grid, blk, 0, c->stream>>>(srcp, c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
    }
    c->i_gauss = src;
    c->fsq_cache_valid = 0;
    if (srcp != c->gauss_ext) c->gauss_ext = nullptr;      // a cascade step ran: the Gaussian now lives in f[src]
    return NL_OK;
}

// The cascade step of scale s+1 only reads the Gaussian of scale s -- like everything else scale s does -- so it can run
// beside it.  nl_gauss_step_ahead enqueues the step on the side stream into the two free ping-pong volumes without
// making it current; nl_gauss_commit (before anything of scale s+1) orders the main stream after it and switches.
// Between the two calls no entry point that uses a free Gaussian volume as scratch may be called (nl_sample_gather,
// nl_mask_volume*, Label); the per-scale calls of Filter do not.
extern "C" int nl_gauss_step_ahead(nl_ctx *c, const double *wz, int rz, const double *wy, int ry, const double *wx, int rx,
                                   int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->ahead_pending) return nl_fail(err, errlen, NL_ESTATE, "a step enqueued ahead is already pending");
    NL_HIP(hipEventRecord(c->ev_main, c->stream));
    hipStream_t ahead_stream = c->side;
    NL_HIP(hipStreamWaitEvent(ahead_stream, c->ev_main, 0));
    const int cur_idx = c->i_gauss;
    float *cur_ext = c->gauss_ext;
    hipStream_t main_stream = c->stream;
    c->stream = ahead_stream;                  // the launch helpers use c->stream
    const int rc = nl_gauss_step(c, wz, rz, wy, ry, wx, rx, z0, z1, err, errlen);
    c->stream = main_stream;
    if (rc) { c->i_gauss = cur_idx; c->gauss_ext = cur_ext; return rc; }
    c->ahead_gauss = c->i_gauss;
    c->i_gauss = cur_idx; c->gauss_ext = cur_ext;
    NL_HIP(hipEventRecord(c->ev_ahead, ahead_stream));
    c->ahead_pending = 1;
    return NL_OK;
}

extern "C" int nl_gauss_commit(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->ahead_pending) return nl_fail(err, errlen, NL_ESTATE, "nl_gauss_commit without nl_gauss_step_ahead");
    NL_HIP(hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
    c->i_gauss = c->ahead_gauss;
    c->gauss_ext = nullptr;
    c->ahead_pending = 0;
    c->fsq_cache_valid = 0;
    return NL_OK;
}

static int make_lattice(const nl_ctx *c, i64 sz, i64 sy, i64 sx, Lattice &L, char *err, size_t errlen) {
    if (sz < 1 || sy < 1 || sx < 1) return nl_fail(err, errlen, NL_EINVAL, "strides must be >= 1");
    L.sz = sz; L.sy = sy; L.sx = sx;
    // owned global planes [g_lo, g_hi): lattice planes are global z = k*sz
    const i64 g_lo = c->gz0 + c->own_lo, g_hi = c->gz0 + c->own_hi;
    const i64 k_lo = (g_lo + sz - 1) / sz, k_hi = (g_hi + sz - 1) / sz;   // k in [k_lo, k_hi)
    L.cz = k_hi > k_lo ? k_hi - k_lo : 0;
    L.zfirst = k_lo * sz - c->gz0;
    L.cy = (c->ny + sy - 1) / sy;
    L.cx = (c->nx + sx - 1) / sx;
    return NL_OK;
}

static int make_field(nl_ctx *c, int field, FieldSrc &fs, char *err, size_t errlen) {
    fs.field = field; fs.hp = hessp(c); fs.max_abs = c->frob_max_abs; fs.max_finite = c->frob_max_finite;
    fs.two_d = c->two_d; fs.bits = nullptr; fs.wpr = 0; fs.fsq_cache = nullptr; fs.norm_dev = nullptr;
    if (field == NL_FIELD_GAUSS) fs.p = gauss_cur(c);
    else if (field == NL_FIELD_FROB) {
        if (!c->have_spacing) return nl_fail(err, errlen, NL_ESTATE, "NL_FIELD_FROB before nl_hessian_stats");
        fs.p = gauss_cur(c);
    } else if (field == NL_FIELD_FRANGI) fs.p = c->f[c->i_vmax];
    else if (field == NL_FIELD_VESSELNESS) {
        if (c->mask_slots_used == 0) return nl_fail(err, errlen, NL_ESTATE, "NL_FIELD_VESSELNESS before any scale was evaluated");
        NL_JOIN_SIDE(c);
        fs.p = c->f[c->i_vmax];
        fs.wpr = (int)((c->nx + 63) / 64);
        fs.bits = (const unsigned long long *)c->m[0] + (i64)((c->mask_slots_used - 1) & 1) * (c->nzl * c->ny * fs.wpr);
    } else return nl_fail(err, errlen, NL_EINVAL, "unknown field %d", field);
    return NL_OK;
}

// NL_FIELD_FROB is sampled up to four times per scale (threshold bracket and exact threshold, min/max and histogram
// each) with different normalisations of the same frob_sq: evaluate the Hessian at the lattice points once.
static int use_fsq_cache(nl_ctx *c, FieldSrc &fs, const Lattice &L, char *err, size_t errlen) {
    if (fs.field != NL_FIELD_FROB) return NL_OK;
    const i64 total = L.cz * L.cy * L.cx;
    if (total == 0) return NL_OK;
    if (!(c->fsq_cache_valid && c->fsq_cache_key[0] == L.sz && c->fsq_cache_key[1] == L.sy && c->fsq_cache_key[2] == L.sx)) {
        if (total > c->fsq_cache_cap) {
            if (c->d_fsq_cache) NL_HIP(hipFree(c->d_fsq_cache));
            c->d_fsq_cache = nullptr; c->fsq_cache_cap = 0;
            NL_HIP(hipMalloc((void **)&c->d_fsq_cache, (size_t)total * 4));
            c->fsq_cache_cap = total;
        }
        ProfScope ps(c, "sample");
        sample_fsq_kernel<<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(fs, geom(c), L, c->d_fsq_cache);
        NL_CHECK_LAUNCH();
        c->fsq_cache_key[0] = L.sz; c->fsq_cache_key[1] = L.sy; c->fsq_cache_key[2] = L.sx;
        c->fsq_cache_valid = 1;
    }
    fs.fsq_cache = c->d_fsq_cache;
    return NL_OK;
}

extern "C" int nl_sample_gather(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, float *out, int64_t cap,
                                int64_t *n, char *err, size_t errlen) {
    NL_ENTER(c);
    Lattice L; FieldSrc fs; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    if (n) *n = total;
    if (total == 0 || (!out && cap == 0)) return NL_OK;   // size query
    if (!out || cap < total) return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld < %lld samples", (i64)cap, total);
    // a free float volume as staging: whichever of f[0..2] is not the current gauss
    float *stage = c->f[(c->i_gauss + 1) % 3];
    if (total > c->n) return nl_fail(err, errlen, NL_EINVAL, "lattice larger than the volume");
    {
        ProfScope ps(c, "sample");
        sample_gather_kernel<<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(fs, geom(c), L, stage);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(out, stage, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// The positive samples of the same lattice, compacted on the device: only they cross PCIe (the consumers take
// arr[arr > 0] first anyway: filtering.py:357, 957-959).  Order unspecified.  cap >= number of lattice points.
// A positive gather leaves its samples in `stage` and their number in *d_n.  Fetching them used to be two round trips (the count,
// then that many samples); the count and the first NL_PREFIX samples now travel together into pinned memory, and only a longer
// list costs a second transfer.  *n = the count; out[0 .. n) = the samples.
#define NL_PREFIX 32768
static int fetch_counted(nl_ctx *c, const float *stage, const unsigned int *d_n, i64 max_count, float *out, i64 cap, int64_t *n, char *err, size_t errlen) {
    if (!c->h_prefix) NL_HIP(hipHostMalloc(&c->h_prefix, (size_t)NL_PREFIX * 4 + 64, hipHostMallocDefault));
    unsigned int *h_n = (unsigned int *)c->h_prefix;
    float *h_s = (float *)((char *)c->h_prefix + 64);
    const i64 first = max_count < NL_PREFIX ? max_count : NL_PREFIX;
    NL_HIP(hipMemcpyAsync(h_n, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    if (first > 0) NL_HIP(hipMemcpyAsync(h_s, stage, (size_t)first * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    const i64 k = (i64)*h_n;
    if (n) *n = 0;
    if (k > cap || (k && !out)) return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld < %lld positive samples", (i64)cap, k);
    if (k > max_count) return nl_fail(err, errlen, NL_ESTATE, "positive gather counted %lld of at most %lld samples", k, max_count);
    if (k) memcpy(out, h_s, (size_t)(k < first ? k : first) * 4);
    if (k > first) {
        NL_HIP(hipMemcpyAsync(out + first, stage + first, (size_t)(k - first) * 4, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
    }
    if (n) *n = k;
    return NL_OK;
}

// The positive lattice samples in two halves, so that the host can do other work (nl_chain_finish: wait for the chain's
// records, repeat its decisions) while the kernel runs: _begin enqueues the kernel and the download of the count, _end waits and
// fetches the samples.  No other call on this context in between except nl_chain_finish / nl_chain_log.
extern "C" int nl_sample_gather_positive_begin(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, int64_t *n_lattice, char *err, size_t errlen) {
    NL_ENTER(c);
    Lattice L; FieldSrc fs; int rc;
    c->gp_total = -1;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    if (n_lattice) *n_lattice = total;
    c->gp_total = total;
    if (total == 0) return NL_OK;
    if (total > c->n) { c->gp_total = -1; return nl_fail(err, errlen, NL_EINVAL, "lattice larger than the volume"); }
    c->gp_stage = c->f[(c->i_gauss + 1) % 3];
    unsigned int *d_n = (unsigned int *)c->d_small;
    NL_HIP(zero_small(d_n, 4, c->stream));
    {
        ProfScope ps(c, "sample");
        sample_gather_pos_kernel<<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(fs, geom(c), L, c->gp_stage, d_n);
        NL_CHECK_LAUNCH();
    }
    return NL_OK;
}
extern "C" int nl_sample_gather_positive_end(nl_ctx *c, float *out, int64_t cap, int64_t *n, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->gp_total < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_sample_gather_positive_end without _begin");
    const i64 total = c->gp_total;
    c->gp_total = -1;
    if (n) *n = 0;
    if (total == 0) return NL_OK;
    return fetch_counted(c, c->gp_stage, (const unsigned int *)c->d_small, total, out, cap, n, err, errlen);
}
extern "C" int nl_sample_gather_positive(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, float *out, int64_t cap,
                                         int64_t *n, char *err, size_t errlen) {
    int64_t total = 0;
    if (n) *n = 0;
    int rc = nl_sample_gather_positive_begin(c, field, sz, sy, sx, &total, err, errlen);
    if (rc) return rc;
    if (total > 0 && (!out || cap < total)) { c->gp_total = -1; return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld < %lld lattice points", (i64)cap, (i64)total); }
    return nl_sample_gather_positive_end(c, out, cap, n, err, errlen);
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
// res = [min bits, max bits, count lo, count hi] of positive float32 samples: unsigned order = float order
static int reduce_range(nl_ctx *c, unsigned int *res, char *err, size_t errlen) {
    NL_NCCL(rccl().GroupStart());
    NL_NCCL(rccl().AllReduce(res, res, 1, ncclUint32, ncclMin, (ncclComm_t)c->comm, c->stream));
    NL_NCCL(rccl().AllReduce(res + 1, res + 1, 1, ncclUint32, ncclMax, (ncclComm_t)c->comm, c->stream));
    NL_NCCL(rccl().AllReduce(res + 2, res + 2, 1, ncclUint64, ncclSum, (ncclComm_t)c->comm, c->stream));
    NL_NCCL(rccl().GroupEnd());
    return NL_OK;
}
static int reduce_u64_sum(nl_ctx *c, unsigned long long *v, size_t n, char *err, size_t errlen) {
    NL_NCCL(rccl().AllReduce(v, v, n, ncclUint64, ncclSum, (ncclComm_t)c->comm, c->stream));
    return NL_OK;
}
static int reduce_u32_max(nl_ctx *c, unsigned int *v, size_t n, char *err, size_t errlen) {
    NL_NCCL(rccl().AllReduce(v, v, n, ncclUint32, ncclMax, (ncclComm_t)c->comm, c->stream));
    return NL_OK;
}

extern "C" int nl_sample_minmax(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, float *mn, float *mx,
                                int64_t *npos, char *err, size_t errlen) {
    NL_ENTER(c);
    Lattice L; FieldSrc fs; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    unsigned int *res = (unsigned int *)c->d_small;
    unsigned int *h = (unsigned int *)c->h_small;
    h[0] = 0xffffffffu; h[1] = 0; h[2] = 0; h[3] = 0;
    NL_HIP(hipMemcpyAsync(res, h, 16, hipMemcpyHostToDevice, c->stream));
    if (total > 0) {
        ProfScope ps(c, "sample");
        sample_minmax_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, 0, c->stream>>>(fs, geom(c), L, res);
        NL_CHECK_LAUNCH();
    }
    if (fused(c) && (rc = reduce_range(c, res, err, errlen))) return rc;
    NL_HIP(hipMemcpyAsync(h, res, 16, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    const unsigned long long cnt = *(unsigned long long *)(h + 2);
    if (npos) *npos = (int64_t)cnt;
    if (cnt) {
        if (mn) memcpy(mn, &h[0], 4);
        if (mx) memcpy(mx, &h[1], 4);
    }
    return NL_OK;
}

extern "C" int nl_sample_hist(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, const float *edges, int nbins,
                              int64_t *counts, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!edges || !counts || nbins < 1 || nbins > 4096) return nl_fail(err, errlen, NL_EINVAL, "bad histogram arguments (nbins=%d)", nbins);
    Lattice L; FieldSrc fs; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    // d_small layout: [0, 32K) counts (u64 x nbins), [32K, 64K) edges (f32 x nbins+1)
    unsigned long long *d_counts = (unsigned long long *)c->d_small;
    float *d_edges = (float *)((char *)c->d_small + (1 << 15));
    NL_HIP(zero_small(d_counts, (size_t)nbins * 8, c->stream));
    memcpy((char *)c->h_small + (1 << 15), edges, (size_t)(nbins + 1) * 4);
    NL_HIP(hipMemcpyAsync(d_edges, (char *)c->h_small + (1 << 15), (size_t)(nbins + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (total > 0) {
        ProfScope ps(c, "sample");
        const size_t sh = (size_t)(nbins + 2) * 4 + (size_t)nbins * 4;
        sample_hist_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, sh, c->stream>>>(fs, geom(c), L, d_edges, nbins, d_counts, nullptr);
        NL_CHECK_LAUNCH();
    }
    if (fused(c) && (rc = reduce_u64_sum(c, d_counts, (size_t)nbins, err, errlen))) return rc;
    NL_HIP(hipMemcpyAsync(c->h_small, d_counts, (size_t)nbins * 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    memcpy(counts, c->h_small, (size_t)nbins * 8);
    return NL_OK;
}

// nl_sample_minmax + nl_sample_hist in one go: the bin edges numpy would build from the range are formed on the
// device, so the two passes need no host round trip in between.  *valid: 0 no positive sample, 1 ok, 2 range not finite
// (the caller raises numpy's ValueError then).  edges (may be NULL) receives the nbins + 1 device-built edges.
// the kernels of one range + edges + histogram chain, working in the `slot`-th half of the small scratch (device and pinned)
#define NL_RH_SLOT 32768
static int range_hist_enqueue_at(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, int nbins, char *d0, char *h0, char *err, size_t errlen);
static int range_hist_enqueue(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, int nbins, int slot, char *err, size_t errlen) {
    return range_hist_enqueue_at(c, field, sz, sy, sx, nbins, (char *)c->d_small + (size_t)slot * NL_RH_SLOT, (char *)c->h_small + (size_t)slot * NL_RH_SLOT, err, errlen);
}
// d0: the record in device memory; h0: its pinned mirror (the initial state of the range words is uploaded from there), or NULL
// when the record was initialised by the caller (chain_init_kernel)
static int range_hist_enqueue_at(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, int nbins, char *d0, char *h0, char *err, size_t errlen) {
    Lattice L; FieldSrc fs; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, field, fs, err, errlen))) return rc;
    if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    // layout (contiguous, one transfer back): counts (u64 x nbins) | edges (f32 x nbins+1, padded) | range, count, flag
    const size_t off_edges = (size_t)nbins * 8, off_res = off_edges + (((size_t)(nbins + 1) * 4 + 15) & ~(size_t)15);
    unsigned long long *d_counts = (unsigned long long *)d0;
    float *d_edges = (float *)(d0 + off_edges);
    unsigned int *res = (unsigned int *)(d0 + off_res);
    if (h0) {
        unsigned int *h = (unsigned int *)(h0 + off_res);
        h[0] = 0xffffffffu; h[1] = 0; h[2] = 0; h[3] = 0; h[4] = 0;
        NL_HIP(hipMemcpyAsync(res, h, 20, hipMemcpyHostToDevice, c->stream));
        NL_HIP(zero_small(d_counts, (size_t)nbins * 8, c->stream));
    }
    if (total > 0 || fused(c)) {
        // fused: a rank without lattice points of its own still takes part in the collectives and builds the same edges
        ProfScope ps(c, "sample");
        if (total > 0) sample_minmax_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, 0, c->stream>>>(fs, geom(c), L, res);
        if (fused(c) && (rc = reduce_range(c, res, err, errlen))) return rc;
        sample_edges_kernel<<<1, 64, 0, c->stream>>>(res, nbins, d_edges, res + 4);
        const size_t sh = (size_t)(nbins + 2) * 4 + (size_t)nbins * 4;
        if (total > 0) sample_hist_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, sh, c->stream>>>(fs, geom(c), L, d_edges, nbins, d_counts, res + 4);
        NL_CHECK_LAUNCH();
        if (fused(c) && (rc = reduce_u64_sum(c, d_counts, (size_t)nbins, err, errlen))) return rc;
    }
    return NL_OK;
}
// The Gaussian and the raw-Frobenius records of one scale in three launches instead of seven: one pass fills the frob_sq
// cache and both ranges, one builds both edge arrays, one bins both.  With fused reductions: two grouped collectives instead of
// four.  hG / hF: pinned mirrors the initial state is uploaded from, or NULL when the caller initialised the records (chain.inc).
static int range_hist_pair_enqueue(nl_ctx *c, int64_t sz, int64_t sy, int64_t sx, int nbins, char *dG, char *dF, char *hG, char *hF,
                                   char *err, size_t errlen) {
    Lattice L; FieldSrc fsG, fsF; int rc;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, NL_FIELD_GAUSS, fsG, err, errlen))) return rc;
    if ((rc = make_field(c, NL_FIELD_FROB, fsF, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    const size_t off_edges = (size_t)nbins * 8, off_res = off_edges + (((size_t)(nbins + 1) * 4 + 15) & ~(size_t)15);
    unsigned int *resG = (unsigned int *)(dG + off_res), *resF = (unsigned int *)(dF + off_res);
    float *edgesG = (float *)(dG + off_edges), *edgesF = (float *)(dF + off_edges);
    for (int k = 0; k < 2; ++k) {
        char *h0 = k ? hF : hG, *d0 = k ? dF : dG;
        if (!h0) continue;
        unsigned int *h = (unsigned int *)(h0 + off_res);
        h[0] = 0xffffffffu; h[1] = 0; h[2] = 0; h[3] = 0; h[4] = 0;
        NL_HIP(hipMemcpyAsync(d0 + off_res, h, 20, hipMemcpyHostToDevice, c->stream));
        NL_HIP(zero_small(d0, (size_t)nbins * 8, c->stream));
    }
    if (total > c->fsq_cache_cap) {
        if (c->d_fsq_cache) NL_HIP(hipFree(c->d_fsq_cache));
        c->d_fsq_cache = nullptr; c->fsq_cache_cap = 0;
        NL_HIP(hipMalloc((void **)&c->d_fsq_cache, (size_t)total * 4));
        c->fsq_cache_cap = total;
    }
    if (total == 0 && !fused(c)) return NL_OK;
    ProfScope ps(c, "sample");
    if (total > 0) {
        sample_minmax2_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, 0, c->stream>>>(fsG, fsF, geom(c), L, resG, resF, c->d_fsq_cache);
        c->fsq_cache_key[0] = L.sz; c->fsq_cache_key[1] = L.sy; c->fsq_cache_key[2] = L.sx;
        c->fsq_cache_valid = 1;
        fsF.fsq_cache = c->d_fsq_cache;
    }
    if (fused(c)) {
        NL_NCCL(rccl().GroupStart());
        for (unsigned int *res : {resG, resF}) {
            NL_NCCL(rccl().AllReduce(res, res, 1, ncclUint32, ncclMin, (ncclComm_t)c->comm, c->stream));
            NL_NCCL(rccl().AllReduce(res + 1, res + 1, 1, ncclUint32, ncclMax, (ncclComm_t)c->comm, c->stream));
            NL_NCCL(rccl().AllReduce(res + 2, res + 2, 1, ncclUint64, ncclSum, (ncclComm_t)c->comm, c->stream));
        }
        NL_NCCL(rccl().GroupEnd());
    }
    sample_edges2_kernel<<<2, 64, 0, c->stream>>>(resG, edgesG, resG + 4, resF, edgesF, resF + 4, nbins);
    const size_t sh = 2 * ((size_t)(nbins + 2) * 4 + (size_t)nbins * 4);
    if (total > 0)
        sample_hist2_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, sh, c->stream>>>(fsG, fsF, geom(c), L, nbins, edgesG, (unsigned long long *)dG, resG + 4,
                                                                                         edgesF, (unsigned long long *)dF, resF + 4);
    NL_CHECK_LAUNCH();
    if (fused(c)) {
        NL_NCCL(rccl().GroupStart());
        NL_NCCL(rccl().AllReduce(dG, dG, (size_t)nbins, ncclUint64, ncclSum, (ncclComm_t)c->comm, c->stream));
        NL_NCCL(rccl().AllReduce(dF, dF, (size_t)nbins, ncclUint64, ncclSum, (ncclComm_t)c->comm, c->stream));
        NL_NCCL(rccl().GroupEnd());
    }
    return NL_OK;
}
static void range_hist_read(const nl_ctx *c, int nbins, int slot, float *mn, float *mx, int64_t *npos, int64_t *counts, float *edges, int *valid) {
    const size_t off_edges = (size_t)nbins * 8, off_res = off_edges + (((size_t)(nbins + 1) * 4 + 15) & ~(size_t)15);
    const char *h0 = (const char *)c->h_small + (size_t)slot * NL_RH_SLOT;
    const unsigned int *hr = (const unsigned int *)(h0 + off_res);
    const unsigned long long cnt = *(const unsigned long long *)(hr + 2);
    if (npos) *npos = (int64_t)cnt;
    *valid = (int)hr[4];
    if (cnt) {
        if (mn) memcpy(mn, &hr[0], 4);
        if (mx) memcpy(mx, &hr[1], 4);
    }
    memcpy(counts, h0, (size_t)nbins * 8);
    if (edges) memcpy(edges, h0 + off_edges, (size_t)(nbins + 1) * 4);
}

extern "C" int nl_sample_range_hist(nl_ctx *c, int field, int64_t sz, int64_t sy, int64_t sx, int nbins, float *mn, float *mx,
                                    int64_t *npos, int64_t *counts, float *edges, int *valid, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!counts || !valid || nbins < 1 || nbins > 2048) return nl_fail(err, errlen, NL_EINVAL, "bad histogram arguments (nbins=%d)", nbins);
    int rc = range_hist_enqueue(c, field, sz, sy, sx, nbins, 0, err, errlen);
    if (rc) return rc;
    const size_t bytes = (size_t)nbins * 8 + (((size_t)(nbins + 1) * 4 + 15) & ~(size_t)15) + 32;
    NL_HIP(hipMemcpyAsync(c->h_small, c->d_small, bytes, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    range_hist_read(c, nbins, 0, mn, mx, npos, counts, edges, valid);
    return NL_OK;
}

// Two independent fields in one round trip (the gamma samples of the Gaussian and the raw Frobenius samples of a scale:
// filtering.py:365-380 and 421-444 need nothing from each other).  Arrays of two: [0] = field_a, [1] = field_b.
extern "C" int nl_sample_range_hist2(nl_ctx *c, int field_a, int field_b, int64_t sz, int64_t sy, int64_t sx, int nbins, float *mn, float *mx,
                                     int64_t *npos, int64_t *counts, float *edges, int *valid, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!counts || !valid || !mn || !mx || !npos || nbins < 1 || nbins > 2048) return nl_fail(err, errlen, NL_EINVAL, "bad histogram arguments (nbins=%d)", nbins);
    int rc;
    const bool fresh_cache = !(c->fsq_cache_valid && c->fsq_cache_key[0] == sz && c->fsq_cache_key[1] == sy && c->fsq_cache_key[2] == sx);
    if (field_a == NL_FIELD_GAUSS && field_b == NL_FIELD_FROB && fresh_cache && nbins <= 1024 && !chain_unfused_sampling()) {
        // the pair of a scale's first round (filtering.py:365-380, 421-444): one pass over the lattice
        if ((rc = range_hist_pair_enqueue(c, sz, sy, sx, nbins, (char *)c->d_small, (char *)c->d_small + NL_RH_SLOT, (char *)c->h_small,
                                          (char *)c->h_small + NL_RH_SLOT, err, errlen))) return rc;
    } else {
        if ((rc = range_hist_enqueue(c, field_a, sz, sy, sx, nbins, 0, err, errlen))) return rc;
        if ((rc = range_hist_enqueue(c, field_b, sz, sy, sx, nbins, 1, err, errlen))) return rc;
    }
    const size_t bytes = (size_t)nbins * 8 + (((size_t)(nbins + 1) * 4 + 15) & ~(size_t)15) + 32;
    for (int k = 0; k < 2; ++k)
        NL_HIP(hipMemcpyAsync((char *)c->h_small + (size_t)k * NL_RH_SLOT, (char *)c->d_small + (size_t)k * NL_RH_SLOT, bytes, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k)
        range_hist_read(c, nbins, k, mn + k, mx + k, npos + k, counts + (size_t)k * nbins, edges ? edges + (size_t)k * (nbins + 1) : nullptr, valid + k);
    return NL_OK;
}

extern "C" int nl_hist_thresholds(const int64_t *counts, const float *edges, int nbins, double *triangle, double *otsu, int *status,
                                  char *err, size_t errlen) {
    if (!counts || !edges || !triangle || !otsu || !status || nbins < 1 || nbins > (1 << 20))
        return nl_fail(err, errlen, NL_EINVAL, "bad histogram arguments (nbins=%d)", nbins);
    hist_thresholds_host<float>(counts, edges, nbins, triangle, otsu, status);
    return NL_OK;
}

static void host_edges(float first, float last, int nbins, float *edges);
// np.histogram(values, bins=nbins, range=(min, max)) of float32 host data + the two thresholds of that histogram, in one call
// (labelling.py:448-455 after the log10: the samples are a few 10^4 values, numpy spends ~0.2-0.6 ms on them while the GPU
// waits).  Same float32 arithmetic as sample_edges_kernel / sample_hist_kernel, which are pinned against numpy.  *status: 0 ok,
// 1 degenerate triangle (numpy's ValueError), 2 range not finite (numpy's ValueError).  counts / edges: optional copies.
extern "C" int nl_host_hist_thresholds_f32(const float *values, int64_t n, int nbins, double *triangle, double *otsu, int *status,
                                           int64_t *counts_out, float *edges_out, char *err, size_t errlen) {
    if (!values || n < 1 || !triangle || !otsu || !status || nbins < 1 || nbins > (1 << 20))
        return nl_fail(err, errlen, NL_EINVAL, "bad histogram arguments (n=%lld, nbins=%d)", (long long)n, nbins);
    float mn = values[0], mx = values[0];
    bool nan = false;
    for (int64_t i = 0; i < n; ++i) {
        const float a = values[i];
        if (a != a) nan = true;
        if (a < mn) mn = a;
        if (a > mx) mx = a;
    }
    *status = 0; *triangle = 0.0; *otsu = 0.0;
    if (nan || !(fabsf(mn) <= 3.402823466e38f) || !(fabsf(mx) <= 3.402823466e38f)) { *status = 2; return NL_OK; }
    std::vector<float> edges((size_t)nbins + 1);
    std::vector<int64_t> counts((size_t)nbins, 0);
    host_edges(mn, mx, nbins, edges.data());
    volatile float first = mn, last = mx;
    if (mn == mx) { first = mn - 0.5f; last = mx + 0.5f; }
    const float f0 = first, f1 = last;
    volatile float denom = f1 - f0;
    const float dn = denom, nb = (float)nbins;
    for (int64_t i = 0; i < n; ++i) {
        const float a = values[i];
        if (!(a >= f0 && a <= f1)) continue;
        const float t = ((a - f0) / dn) * nb;          // float32 throughout (x86-64 SSE, -ffp-contract=off): numpy's expression
        int idx = (int)t;
        if (idx == nbins) idx -= 1;
        if (a < edges[idx]) idx -= 1;
        if (a >= edges[idx + 1] && idx != nbins - 1) idx += 1;
        counts[idx] += 1;
    }
    hist_thresholds_host<float>(counts.data(), edges.data(), nbins, triangle, otsu, status, nullptr);
    if (counts_out) memcpy(counts_out, counts.data(), (size_t)nbins * 8);
    if (edges_out) memcpy(edges_out, edges.data(), ((size_t)nbins + 1) * 4);
    return NL_OK;
}

extern "C" int nl_hist_thresholds_ex(const int64_t *counts, const void *edges, int edges_f64, int nbins, double *triangle, double *otsu,
                                     double *otsu_var, int *status, char *err, size_t errlen) {
    if (!counts || !edges || !triangle || !otsu || !status || nbins < 1 || nbins > (1 << 20))
        return nl_fail(err, errlen, NL_EINVAL, "bad histogram arguments (nbins=%d)", nbins);
    if (edges_f64) hist_thresholds_host<double>(counts, (const double *)edges, nbins, triangle, otsu, status, otsu_var);
    else hist_thresholds_host<float>(counts, (const float *)edges, nbins, triangle, otsu, status, otsu_var);
    return NL_OK;
}

static int set_spacing(nl_ctx *c, const double spacing[3], char *err, size_t errlen) {
    if (!spacing) return nl_fail(err, errlen, NL_EINVAL, "spacing is NULL");
    if ((!c->two_d && c->gnz < 2) || c->ny < 2 || c->nx < 2)
        return nl_fail(err, errlen, NL_EINVAL, "Shape of array too small to calculate a numerical gradient, at least (edge_order + 1) elements are required.");
    if (c->chk_spacing[0] != spacing[0] || c->chk_spacing[1] != spacing[1] || c->chk_spacing[2] != spacing[2]) c->fsq_cache_valid = 0;
    c->hz = (float)spacing[0]; c->hy = (float)spacing[1]; c->hx = (float)spacing[2];
    c->hz2 = (float)(2.0 * spacing[0]); c->hy2 = (float)(2.0 * spacing[1]); c->hx2 = (float)(2.0 * spacing[2]);
    if (!c->have_spacing || c->chk_spacing[0] != spacing[0] || c->chk_spacing[1] != spacing[1] || c->chk_spacing[2] != spacing[2]) {
        int rcx = check_fast_div(c, err, errlen);
        if (rcx) return rcx;
        c->chk_spacing[0] = spacing[0]; c->chk_spacing[1] = spacing[1]; c->chk_spacing[2] = spacing[2];
    }
    c->have_spacing = 1;
    return NL_OK;
}

extern "C" int nl_set_spacing(nl_ctx *c, const double spacing[3], char *err, size_t errlen) {
    NL_ENTER(c);
    return set_spacing(c, spacing, err, errlen);
}

extern "C" int nl_hessian_stats(nl_ctx *c, const double spacing[3], float *max_abs, float *max_frob_sq, int *any_inf,
                                char *err, size_t errlen) {
    NL_ENTER(c);
    { int rcs = set_spacing(c, spacing, err, errlen); if (rcs) return rcs; }
    c->spec_valid = 0;
    if (c->two_d) {
        unsigned int *res2 = (unsigned int *)c->d_small;
        NL_HIP(zero_small(res2, 16, c->stream));
        {
            ProfScope ps(c, "hessian_stats");
            hessian2d_stats_kernel<<<grid2d_rows(c->nx, c->ny), 256, 0, c->stream>>>(gauss_cur(c), geom(c), hessp(c), res2);
            NL_CHECK_LAUNCH();
        }
        unsigned int *h2 = (unsigned int *)c->h_small;
        NL_HIP(hipMemcpyAsync(h2, res2, 16, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
        if (max_abs) memcpy(max_abs, &h2[0], 4);
        if (max_frob_sq) memcpy(max_frob_sq, &h2[1], 4);
        if (any_inf) *any_inf = (int)h2[2];
        return NL_OK;
    }
    unsigned int *res = (unsigned int *)c->d_small;
    NL_HIP(zero_small(res, 16, c->stream));
    {
        ProfScope ps(c, "hessian_stats");
        const int ntx = (int)((c->nx + HM_TX - 1) / HM_TX), nty = (int)((c->ny + 15) / 16);
        const int nzc = (int)((c->own_hi - c->own_lo + HM_ZCHUNK - 1) / HM_ZCHUNK);
        VessP vp{};
#define NL_LAUNCH_STATS(TYV, FASTV, HR)                                                                                   \
        hessian_g_kernel<0, TYV, FASTV><<<(unsigned)(ntx * (int)((c->ny + TYV - 1) / TYV) * nzc), HGCfg<TYV>::NT,         \
                                          HGCfg<TYV>::lds_bytes(), c->stream>>>(                                          \
            gauss_cur(c), nullptr, nullptr, 0, geom(c), HR, vp, VQueue{}, (int)c->own_lo, (int)c->own_hi, ntx,        \
            (int)((c->ny + TYV - 1) / TYV), res, nullptr)
        if (hv_rs(c)) {
            const int rsv = hv_rs(c), ntyv = (int)((c->ny + 2 * rsv - 1) / (2 * rsv));
            NL_HIP(nl_hv_launch(HvLaunch{0, rsv, hv_fastv(c), (unsigned)(ntx * ntyv * nzc), c->stream, gauss_cur(c), nullptr, nullptr, 0, geom(c),
                                         hessp(c), vp, VQueue{}, (int)c->own_lo, (int)c->own_hi, ntx, ntyv, res, nullptr, nullptr}));
        }
        else if (hm_ty() == 8) { if (c->fast_div) NL_LAUNCH_STATS(8, true, hessdv_fast(c)); else NL_LAUNCH_STATS(8, false, hessdv_exact(c)); }
        else { if (c->fast_div) NL_LAUNCH_STATS(16, true, hessdv_fast(c)); else NL_LAUNCH_STATS(16, false, hessdv_exact(c)); }
#undef NL_LAUNCH_STATS
        NL_CHECK_LAUNCH();
    }
    unsigned int *h = (unsigned int *)c->h_small;
    NL_HIP(hipMemcpyAsync(h, res, 16, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (max_abs) memcpy(max_abs, &h[0], 4);
    if (max_frob_sq) memcpy(max_frob_sq, &h[1], 4);
    if (any_inf) *any_inf = (int)h[2];
    return NL_OK;
}

extern "C" int nl_set_frob_norm(nl_ctx *c, float max_abs, float max_finite, char *err, size_t errlen) {
    if (!c) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");
    c->frob_max_abs = max_abs; c->frob_max_finite = max_finite;
    return NL_OK;
}

// h_mask = sqrt(frob_sq)/max_abs > thr (or > 0).  sqrt and the division by a positive constant are monotone, so the
// mask is exactly {frob_sq >= x_min} for the smallest float32 x_min whose image passes the test; it is found by
// bisection over the (ordered) bit patterns of the non-negative floats, with the same two IEEE operations the
// volume op would do.  The kernel then needs one compare per voxel instead of a square root and a division.
static float mask_threshold_on_fsq(float max_abs, int use_thr, float thr) {
    auto pred = [&](float x) -> bool {
        volatile float fr = sqrtf(x);
        fr = fr / max_abs;
        return use_thr ? (fr > thr) : (fr > 0.0f);
    };
    unsigned int lo = 0u, hi = 0x7f800000u;          // hi = +inf bits: "nothing finite passes"
    float fmaxv; { unsigned int b = 0x7f7fffffu; memcpy(&fmaxv, &b, 4); }
    if (!pred(fmaxv)) { float inf; memcpy(&inf, &hi, 4); return inf; }
    hi = 0x7f7fffffu;
    while (lo < hi) {                                 // smallest pattern with pred true
        const unsigned int mid = lo + (hi - lo) / 2;
        float x; memcpy(&x, &mid, 4);
        if (pred(x)) hi = mid; else lo = mid + 1;
    }
    float r; memcpy(&r, &lo, 4);
    return r;
}

static VessP make_vessp(nl_ctx *c, float gamma_sq, float alpha_sq, float beta_sq, int use_thr, float thr) {
    VessP vp{};
    vp.gamma_sq = gamma_sq; vp.alpha_sq = alpha_sq; vp.beta_sq = beta_sq; vp.use_thr = use_thr; vp.thr = thr;
    vp.max_abs = c->frob_max_abs; vp.max_finite = c->frob_max_finite;
    vp.cnt_lo = (int)c->own_lo; vp.cnt_hi = (int)c->own_hi;
    vp.first = c->mask_slots_used == 0 ? 1 : 0;
    vp.fsq_min = mask_threshold_on_fsq(c->frob_max_abs, use_thr, thr);
    vp.m_inf = use_thr ? (c->frob_max_finite > thr) : (c->frob_max_finite > 0.0f);
    c->last_fsq_min = vp.fsq_min;
    return vp;
}

// The walk of a scale in MODE 2, enqueued: statistics into res[0..3] (all "max": bit patterns of non-negative floats, flags),
// h_mask count of the decided voxels into *d_cnt (both zeroed by the caller).  dev_lohi != NULL: the bracket is read from device
// memory by the kernel (chain.inc) and fsq_lo / fsq_hi are ignored.
static int spec_enqueue(nl_ctx *c, const double spacing[3], float fsq_lo, float fsq_hi, int64_t z0, int64_t z1, unsigned int *res,
                        unsigned long long *d_cnt, const float *dev_lohi, char *err, size_t errlen) {
    if (z0 < 0 && z1 < 0) { z0 = c->own_lo; z1 = c->own_hi; }
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    if (z0 > c->own_lo || z1 < c->own_hi) return nl_fail(err, errlen, NL_EINVAL, "the plane range must cover the owned planes");
    NL_JOIN_SIDE(c);
    if (!c->spec_ok) return nl_fail(err, errlen, NL_ESTATE, "one-pass vesselness is not available for this context (queue too large)");
    if (!dev_lohi && !(fsq_lo <= fsq_hi)) return nl_fail(err, errlen, NL_EINVAL, "empty bracket [%g,%g]", (double)fsq_lo, (double)fsq_hi);
    { int rcs = set_spacing(c, spacing, err, errlen); if (rcs) return rcs; }
    VessP vp{};
    vp.cnt_lo = (int)c->own_lo; vp.cnt_hi = (int)c->own_hi;
    vp.have_prev = c->mask_slots_used > 0;
    vp.fsq_lo = fsq_lo; vp.fsq_hi = fsq_hi; vp.qcap = HM_SPEC_CAP;
    {
        ProfScope ps(c, "vesselness");
        const int ntx = (int)((c->nx + HM_TX - 1) / HM_TX);
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 slot_words = c->nzl * c->ny * wpr;
        const int k_scale = c->mask_slots_used;          // committed by nl_vesselness_resolve
        unsigned long long *cm = (unsigned long long *)c->m[0] + (i64)(k_scale & 1) * slot_words;
        const unsigned long long *pm = (unsigned long long *)c->m[0] + (i64)((k_scale + 1) & 1) * slot_words;
        const VQueue vq{(float4 *)c->d_vq, c->d_vq_count};
        const int rs = hv_rs(c);
        if (dev_lohi && !rs) return nl_fail(err, errlen, NL_ESTATE, "the device-resident chain needs the two-voxel walk");
        const int ty = rs ? 2 * rs : hm_ty();
        const int nty = (int)((c->ny + ty - 1) / ty);
        // Planes per workgroup: 64, or 128 where that still leaves thousands of workgroups (a chunk's prologue -- four planes
        // staged, the first gradient tile -- is paid half as often: walk -3 % at 1024^3; a 128-plane frame would run on 256
        // workgroups and lose 30 %).  The queue was sized for 64-plane chunks: twice the planes, twice the entries per region,
        // half the regions -- the same memory when the chunk count halves exactly or rounds the same way.
        int zch = HM_ZCHUNK;
        {
            static int forced = -1;
            if (forced < 0) { const char *e = getenv("NELLIE_HV_ZCHUNK"); forced = e ? atoi(e) : 0; }
            const i64 nz = z1 - z0;
            const i64 c64 = (nz + HM_ZCHUNK - 1) / HM_ZCHUNK, c128 = (nz + 2 * HM_ZCHUNK - 1) / (2 * HM_ZCHUNK);
            const bool fits = 2 * c128 <= c64 && (i64)(2 * HM_ZCHUNK + 4) * c->ny * c->nx * 4 < ((i64)1 << 32);
            if (rs && fits && (forced == 2 * HM_ZCHUNK || (forced == 0 && (i64)ntx * nty * c128 >= 4096))) zch = 2 * HM_ZCHUNK;
        }
        vp.zchunk = zch;
        const int nzc = (int)((z1 - z0 + zch - 1) / zch);
        const unsigned nblocks = (unsigned)(ntx * nty * nzc);
        vp.qcap = HM_SPEC_CAP * (zch / HM_ZCHUNK);
        if (rs) vp.qcap *= 2;                           // a wave owns two row segments
        hipStream_t hs = c->stream;
#define NL_DEV_LOHI dev_lohi
#define NL_LAUNCH_SPEC(TYV, FASTV, HR)                                                                                    \
        hessian_g_kernel<2, TYV, FASTV><<<nblocks, HGCfg<TYV>::NT, HGCfg<TYV>::lds_bytes(), c->stream>>>(                 \
            gauss_cur(c), cm, pm, wpr, geom(c), HR, vp, vq, (int)z0, (int)z1, ntx, nty, res, d_cnt)
        if (rs) NL_HIP(nl_hv_launch(HvLaunch{2, rs, hv_fastv(c), nblocks, hs, gauss_cur(c), cm, pm, wpr, geom(c), hessp(c), vp, vq, (int)z0, (int)z1,
                                             ntx, nty, res, d_cnt, NL_DEV_LOHI}));
        else if (ty == 8) { if (c->fast_div) NL_LAUNCH_SPEC(8, true, hessdv_fast(c)); else NL_LAUNCH_SPEC(8, false, hessdv_exact(c)); }
        else { if (c->fast_div) NL_LAUNCH_SPEC(16, true, hessdv_fast(c)); else NL_LAUNCH_SPEC(16, false, hessdv_exact(c)); }
#undef NL_LAUNCH_SPEC
#undef NL_DEV_LOHI
        NL_CHECK_LAUNCH();
        c->spec_nregions = nblocks * (unsigned)(rs ? rs : ty);
        c->spec_qcap = vp.qcap;
    }
    // fused: max |H|, max frob_sq (bit patterns of non-negative floats), the inf and overflow flags -- all "max"; the count stays local
    if (fused(c)) { int rcr = reduce_u32_max(c, res, 4, err, errlen); if (rcr) return rcr; }
    c->spec_z0 = z0; c->spec_z1 = z1;
    return NL_OK;
}

// ---- one-pass vesselness (MODE 2 of the Hessian kernel + the resolve kernel), see hessian.inc ---------------
extern "C" int nl_vesselness_spec(nl_ctx *c, const double spacing[3], float fsq_lo, float fsq_hi, int64_t z0, int64_t z1,
                                  float *max_abs, float *max_frob_sq, int *any_inf, int *overflow, char *err, size_t errlen) {
    NL_ENTER(c);
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    unsigned int *res = (unsigned int *)c->d_small + 16;
    NL_HIP(zero_small(c->d_small, 128, c->stream));
    { int rc = spec_enqueue(c, spacing, fsq_lo, fsq_hi, z0, z1, res, d_cnt, nullptr, err, errlen); if (rc) return rc; }
    unsigned int *h = (unsigned int *)c->h_small;
    NL_HIP(hipMemcpyAsync(h, c->d_small, 128, hipMemcpyDeviceToHost, c->stream));      // [0] count, [16..19] statistics
    NL_HIP(hipStreamSynchronize(c->stream));
    c->spec_count = *(unsigned long long *)c->h_small;       // d_small is scratch for the sampling calls in between
    h += 16;
    if (max_abs) memcpy(max_abs, &h[0], 4);
    if (max_frob_sq) memcpy(max_frob_sq, &h[1], 4);
    if (any_inf) *any_inf = (int)h[2];
    if (overflow) *overflow = (int)h[3];
    c->spec_lo = fsq_lo; c->spec_hi = fsq_hi;
    c->spec_valid = (h[2] == 0 && h[3] == 0) ? 1 : 0;
    c->last_spec_overflow = (int)h[3];
    return NL_OK;
}

static bool resolve_on_side();
// *hit = 1: the exact threshold lies in the bracket of the pass, the scale is complete (mask_count as
// nl_vesselness_step reports it); *hit = 0: nothing was changed, run nl_vesselness_step.
// h_mask count of the scale completed by the last nl_vesselness_resolve hit (waits for its kernel).
extern "C" int nl_vesselness_count(nl_ctx *c, int64_t *mask_count, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!mask_count) return nl_fail(err, errlen, NL_EINVAL, "mask_count is NULL");
    unsigned long long *d_cnt = (unsigned long long *)((char *)c->d_small + (48 << 10));
    unsigned long long *h_cnt = (unsigned long long *)((char *)c->h_small + (48 << 10));
    hipStream_t st = resolve_on_side() ? c->side : c->stream;        // the stream the resolve kernel ran on
    NL_HIP(hipMemcpyAsync(h_cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
    NL_HIP(hipStreamSynchronize(st));
    *mask_count = (int64_t)(*h_cnt + c->spec_count);
    return NL_OK;
}

// The resolve kernel of a scale, enqueued on the side stream (ordered after everything submitted to the main stream so far);
// commits the scale's mask slot.  dev_params != NULL: gamma_sq, fsq_min and m_inf are read from device memory (chain.inc).
// The resolve kernel runs on the MAIN stream by default (round 4): since the two-voxel walk and the cheaper resolve kernel,
// running it beside the next scale's Gaussian buys nothing (48.8 vs 49.0 ms/step at 1024^3: the Z pass slows from 1.77 to 2.55 ms
// per launch while it shares the GPU) -- and sending it to the side stream only to make the main stream wait for it cost two
// cross-queue dependencies of ~13 us per scale (0.13 ms of a 3.3 ms config-5 frame).  NELLIE_RESOLVE_OVERLAP=1: the side stream,
// beside the next cascade step (every entry point that needs its result joins it: NL_JOIN_SIDE).
static bool resolve_on_side() {
    static int overlap = -1;
    if (overlap < 0) { const char *e = getenv("NELLIE_RESOLVE_OVERLAP"); overlap = (e && atoi(e)) ? 1 : 0; }
    return overlap != 0;
}
static int resolve_enqueue(nl_ctx *c, VessP vp, unsigned long long *d_cnt, const float *dev_params, char *err, size_t errlen) {
    const i64 plane = c->ny * c->nx, z0 = c->spec_z0, z1 = c->spec_z1;
    const bool side = resolve_on_side();
    hipStream_t st = side ? c->side : c->stream;
    if (side) {
        NL_HIP(hipEventRecord(c->ev_main, c->stream));
        NL_HIP(hipStreamWaitEvent(c->side, c->ev_main, 0));
    }
    vp.qcap = c->spec_qcap;
    vp.idx_lo = (c->own_lo - z0) * plane; vp.idx_hi = (c->own_hi - z0) * plane;
    {
        ProfScope ps(c, "vesselness_resolve", st);
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 slot_words = c->nzl * c->ny * wpr;
        const int k_scale = c->mask_slots_used++;
        vp.have_prev = k_scale > 0;
        unsigned long long *cm = (unsigned long long *)c->m[0] + (i64)(k_scale & 1) * slot_words;
        const unsigned long long *pm = (unsigned long long *)c->m[0] + (i64)((k_scale + 1) & 1) * slot_words;
        if (vp.first)
            NL_HIP(hipMemsetAsync(c->f[c->i_vmax] + z0 * plane, 0, (size_t)(z1 - z0) * plane * 4, st));
        vesselness_queue_kernel<true><<<resolve_grid((c->spec_nregions + 3) / 4), 256, 0, st>>>(
            (const float4 *)c->d_vq, c->d_vq_count, c->spec_nregions, c->f[c->i_vmax], z0 * plane, vp, cm, pm, wpr, (int)c->ny, (int)c->nx, z0, d_cnt,
            dev_params);
        NL_CHECK_LAUNCH();
    }
    if (side) {
        NL_HIP(hipEventRecord(c->ev_side, c->side));
        c->side_pending = 1;
    }
    c->spec_valid = 0;
    return NL_OK;
}

extern "C" int nl_vesselness_resolve(nl_ctx *c, float gamma_sq, float alpha_sq, float beta_sq, int use_thr, float thr,
                                     int *hit, int64_t *mask_count, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!hit) return nl_fail(err, errlen, NL_EINVAL, "hit is NULL");
    *hit = 0;
    if (!c->spec_valid) return NL_OK;
    VessP vp = make_vessp(c, gamma_sq, alpha_sq, beta_sq, use_thr, thr);
    if (!(vp.fsq_min >= c->spec_lo && vp.fsq_min <= c->spec_hi)) { c->spec_valid = 0; return NL_OK; }
    // The kernel's counter lives outside the sampling scratch.
    unsigned long long *d_cnt = (unsigned long long *)((char *)c->d_small + (48 << 10));
    if (resolve_on_side()) {
        NL_HIP(hipEventRecord(c->ev_main, c->stream));
        NL_HIP(hipStreamWaitEvent(c->side, c->ev_main, 0));
        NL_HIP(zero_small(d_cnt, 8, c->side));
    } else {
        NL_HIP(zero_small(d_cnt, 8, c->stream));
    }
    { int rc = resolve_enqueue(c, vp, d_cnt, nullptr, err, errlen); if (rc) return rc; }
    *hit = 1;
    if (mask_count) {        // asking for the count here waits for the kernel; nl_vesselness_count can be called later instead
        int rcc = nl_vesselness_count(c, mask_count, err, errlen);
        if (rcc) return rcc;
    }
    return NL_OK;
}


// ---- device-resident threshold chain (chain.inc) -----------------------------------------------------------------------------
static void host_edges(float first, float last, int nbins, float *edges) {       // sample_edges_kernel on the host (verification)
    if (first == last) { first = first - 0.5f; last = last + 0.5f; }
    volatile float delta = last - first;
    const float div = (float)nbins;
    volatile float step = delta / div;
    for (int i = 0; i <= nbins; ++i) {
        volatile float y = (float)i;
        if (step == 0.0f) { y = y / div; y = y * delta; } else y = y * step;
        y = y + first;
        edges[i] = (i == nbins) ? last : y;
    }
}
static inline unsigned int f2u(float x) { unsigned int u; memcpy(&u, &x, 4); return u; }
static inline float u2f(unsigned int u) { float x; memcpy(&x, &u, 4); return x; }
static inline float py_min_f(float a, float b) { return b < a ? b : a; }       // python's min(a, b)

// The host repeats what the chain's kernels decided for one scale, with the code of the synchronous path, and compares bit for
// bit.  Returns the record's flags, NL_CF_VERIFY added on any difference.
static int chain_verify(const ChainScale &s, double division, double margin, double test_scale) {
    if (s.flags) return s.flags;
    int bad = 0;
    double tri, otsu; int st;
    {   // gamma
        hist_thresholds_host<float>((const int64_t *)s.h_gauss.counts, s.h_gauss.edges, NL_CHAIN_BINS, &tri, &otsu, &st);
        const float g = py_min_f((float)tri, (float)otsu);
        const float gamma = g > 0.0f ? g : 1.1920929e-07f;
        const float gamma_sq = (float)(2.0 * std::pow((double)gamma, 2.0));
        if (st || f2u(gamma) != f2u(s.gamma) || f2u(gamma_sq) != f2u(s.gamma_sq)) bad = 1;
        float e[NL_CHAIN_BINS + 1];
        host_edges(u2f(s.h_gauss.res[0]), u2f(s.h_gauss.res[1]), NL_CHAIN_BINS, e);
        if (memcmp(e, s.h_gauss.edges, sizeof(e))) bad = 1;
    }
    {   // bracket
        hist_thresholds_host<float>((const int64_t *)s.h_raw.counts, s.h_raw.edges, NL_CHAIN_BINS, &tri, &otsu, &st);
        const double t = (double)py_min_f((float)tri, (float)otsu) * test_scale / division;
        const float lo = (float)(t * t * (1.0 - margin)), hi = (float)(t * t * (1.0 + margin));
        if (st || f2u(lo) != f2u(s.fsq_lo) || f2u(hi) != f2u(s.fsq_hi)) bad = 1;
        float e[NL_CHAIN_BINS + 1];
        host_edges(u2f(s.h_raw.res[0]), u2f(s.h_raw.res[1]), NL_CHAIN_BINS, e);
        if (memcmp(e, s.h_raw.edges, sizeof(e))) bad = 1;
    }
    {   // statistics -> normalisation -> exact threshold -> mask test
        const float max_abs32 = u2f(s.stats[0]), max_fsq32 = u2f(s.stats[1]);
        const float max_abs = max_abs32 <= 0.0f ? 1.0f : max_abs32;
        volatile float mf = sqrtf(max_fsq32); mf = mf / max_abs;
        volatile float mn = u2f(s.h_raw.res[0]) / max_abs, mx = u2f(s.h_raw.res[1]) / max_abs;
        if (f2u(max_abs) != f2u(s.max_abs) || f2u((float)mf) != f2u(s.max_frob) || f2u((float)mn) != f2u(s.nmn) || f2u((float)mx) != f2u(s.nmx) ||
            f2u(s.norm[0]) != f2u(max_abs) || s.norm[1] != 0.0f) bad = 1;
        float e[NL_CHAIN_BINS + 1];
        host_edges((float)mn, (float)mx, NL_CHAIN_BINS, e);
        if (memcmp(e, s.h_exact.edges, sizeof(e))) bad = 1;
        hist_thresholds_host<float>((const int64_t *)s.h_exact.counts, s.h_exact.edges, NL_CHAIN_BINS, &tri, &otsu, &st);
        const float thr = py_min_f((float)tri, (float)otsu);
        const float thr_cmp = (float)((double)thr / division);
        const float fsq_min = mask_threshold_on_fsq(max_abs, 1, thr_cmp);
        if (st || f2u(thr) != f2u(s.thr) || f2u(thr_cmp) != f2u(s.thr_cmp) || f2u(fsq_min) != f2u(s.fsq_min) || s.m_inf != ((0.0f > thr_cmp) ? 1 : 0)) bad = 1;
        if (!(fsq_min >= s.fsq_lo && fsq_min <= s.fsq_hi) || !((float)mf > thr_cmp)) bad = 1;      // (the kernel would have flagged these)
    }
    return bad ? NL_CF_VERIFY : 0;
}

extern "C" int nl_chain_begin(nl_ctx *c, int n_scales, char *err, size_t errlen) {
    NL_ENTER(c);
    if (n_scales < 1 || n_scales > NL_CHAIN_MAX_SCALES) return nl_fail(err, errlen, NL_EINVAL, "a chain holds 1..%d scales", NL_CHAIN_MAX_SCALES);
    if (c->two_d || !c->spec_ok || !hv_rs(c)) return nl_fail(err, errlen, NL_ESTATE, "the device-resident chain needs the 3-D one-pass walk");
    if (!c->d_chain) {
        NL_HIP(hipMalloc(&c->d_chain, sizeof(ChainScale) * NL_CHAIN_MAX_SCALES));
        NL_HIP(hipHostMalloc(&c->h_chain, sizeof(ChainScale) * NL_CHAIN_MAX_SCALES, hipHostMallocDefault));
        NL_HIP(hipEventCreateWithFlags(&c->ev_chain, hipEventDisableTiming));
    }
    chain_init_kernel<<<16, 256, 0, c->stream>>>((ChainScale *)c->d_chain, n_scales);
    chain_init2_kernel<<<1, 64, 0, c->stream>>>((ChainScale *)c->d_chain, n_scales);
    NL_CHECK_LAUNCH();
    c->chain_n = n_scales; c->chain_k = 0; c->chain_copy_pending = 0;
    return NL_OK;
}

// One scale of the frame, enqueued without a single wait: see chain.inc.  The Gaussian of the scale is current (nl_gauss_step).
extern "C" int nl_chain_scale(nl_ctx *c, const double spacing[3], int64_t sz, int64_t sy, int64_t sx, double alpha_sq, double beta_sq,
                              double division, double margin, double test_scale, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->d_chain || c->chain_k >= c->chain_n) return nl_fail(err, errlen, NL_ESTATE, "nl_chain_scale outside nl_chain_begin .. nl_chain_finish");
    if (!(division != 0.0)) return nl_fail(err, errlen, NL_EINVAL, "the chain needs a non-zero threshold division");
    int rc;
    if ((rc = set_spacing(c, spacing, err, errlen))) return rc;
    ChainScale *cs = (ChainScale *)c->d_chain + c->chain_k;
    c->chain_par[c->chain_k][0] = division; c->chain_par[c->chain_k][1] = margin; c->chain_par[c->chain_k][2] = test_scale;
    ++c->chain_k;
    c->frob_max_abs = 1.0f; c->frob_max_finite = 0.0f;                    // the bracket round: max_abs := 1 (pipeline.py _fsq_bracket)
    if (!chain_unfused_sampling()) {
        if ((rc = range_hist_pair_enqueue(c, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_gauss, (char *)&cs->h_raw, nullptr, nullptr, err, errlen))) return rc;
    } else {
        if ((rc = range_hist_enqueue_at(c, NL_FIELD_GAUSS, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_gauss, nullptr, err, errlen))) return rc;
        if ((rc = range_hist_enqueue_at(c, NL_FIELD_FROB, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_raw, nullptr, err, errlen))) return rc;
    }
    chain_thr1_kernel<<<2, 64, 0, c->stream>>>(cs, division, margin, test_scale);
    NL_CHECK_LAUNCH();
    if ((rc = spec_enqueue(c, spacing, 0.0f, 0.0f, z0, z1, cs->stats, &cs->cnt_walk, &cs->fsq_lo, err, errlen))) return rc;
    chain_post_kernel<<<1, 64, 0, c->stream>>>(cs);
    NL_CHECK_LAUNCH();
    {   // the exact round: edges from the normalised range, histogram of the cached frob_sq under the device's normalisation
        Lattice L; FieldSrc fs;
        if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
        if ((rc = make_field(c, NL_FIELD_FROB, fs, err, errlen))) return rc;
        if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
        fs.norm_dev = cs->norm;
        const i64 total = L.cz * L.cy * L.cx;
        ProfScope ps(c, "sample");
        sample_edges_kernel<<<1, 64, 0, c->stream>>>(cs->h_exact.res, NL_CHAIN_BINS, cs->h_exact.edges, cs->h_exact.res + 4);
        const size_t sh = (size_t)(NL_CHAIN_BINS + 2) * 4 + (size_t)NL_CHAIN_BINS * 4;
        if (total > 0) sample_hist_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, sh, c->stream>>>(fs, geom(c), L, cs->h_exact.edges, NL_CHAIN_BINS, cs->h_exact.counts, cs->h_exact.res + 4);
        NL_CHECK_LAUNCH();
        if (fused(c) && (rc = reduce_u64_sum(c, cs->h_exact.counts, NL_CHAIN_BINS, err, errlen))) return rc;
    }
    chain_thr2_kernel<<<1, 64, 0, c->stream>>>(cs, division);
    NL_CHECK_LAUNCH();
    VessP vp{};
    vp.alpha_sq = (float)alpha_sq; vp.beta_sq = (float)beta_sq; vp.use_thr = 1;
    vp.cnt_lo = (int)c->own_lo; vp.cnt_hi = (int)c->own_hi;
    vp.first = c->mask_slots_used == 0 ? 1 : 0;
    return resolve_enqueue(c, vp, &cs->cnt_resolve, (const float *)cs, err, errlen);
}

// Optional, before nl_chain_finish: start the download of the records now, so that work enqueued after this call (the samples
// of the frame's percentile threshold) runs on the device while nl_chain_finish waits for the records only and repeats the
// decisions on the host.
extern "C" int nl_chain_flush(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->d_chain || c->chain_k < 1) return nl_fail(err, errlen, NL_ESTATE, "nl_chain_flush without scales");
    NL_JOIN_SIDE(c);
    NL_HIP(hipMemcpyAsync(c->h_chain, c->d_chain, sizeof(ChainScale) * (size_t)c->chain_k, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipEventRecord(c->ev_chain, c->stream));
    c->chain_copy_pending = 1;
    return NL_OK;
}

// The one wait of the frame: per scale flags (0 = the chain's result stands), gamma, max |H|, the Frobenius threshold and this
// context's h_mask count.  Any non-zero flag: redo the frame the synchronous way.
extern "C" int nl_chain_finish(nl_ctx *c, int *flags, double *gamma, double *max_abs, double *thr, int64_t *mask_count, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->d_chain || c->chain_k < 1) return nl_fail(err, errlen, NL_ESTATE, "nl_chain_finish without scales");
    const int n = c->chain_k;
    if (c->chain_copy_pending) {
        NL_HIP(hipEventSynchronize(c->ev_chain));
        c->chain_copy_pending = 0;
    } else {
        NL_JOIN_SIDE(c);
        NL_HIP(hipMemcpyAsync(c->h_chain, c->d_chain, sizeof(ChainScale) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
    }
    const ChainScale *h = (const ChainScale *)c->h_chain;
    for (int k = 0; k < n; ++k) {
        const int f = chain_verify(h[k], c->chain_par[k][0], c->chain_par[k][1], c->chain_par[k][2]);
        if (flags) flags[k] = f;
        if (gamma) gamma[k] = (double)h[k].gamma;
        if (max_abs) max_abs[k] = (double)h[k].max_abs;
        if (thr) thr[k] = (double)h[k].thr;
        if (mask_count) mask_count[k] = (int64_t)(h[k].cnt_walk + h[k].cnt_resolve);
    }
    c->chain_n = 0;
    return NL_OK;
}

// Test hook: the logged record of scale k after nl_chain_finish -- which: 0 Gaussian samples, 1 raw Frobenius samples, 2 normalised
// Frobenius samples; counts[256], edges[257], range[2] (min, max of the positive samples), scalars[8] = fsq_lo, fsq_hi, gamma_sq,
// fsq_min, thr_cmp, max_frob, tri, otsu (of that histogram).
extern "C" int nl_chain_log(nl_ctx *c, int k, int which, int64_t *counts, float *edges, float *range, double *scalars, char *err, size_t errlen) {
    if (!c || !c->h_chain || k < 0 || k >= NL_CHAIN_MAX_SCALES || which < 0 || which > 2) return nl_fail(err, errlen, NL_EINVAL, "bad chain log request");
    const ChainScale &s = ((const ChainScale *)c->h_chain)[k];
    const ChainHist &h = which == 0 ? s.h_gauss : (which == 1 ? s.h_raw : s.h_exact);
    if (counts) memcpy(counts, h.counts, sizeof(h.counts));
    if (edges) memcpy(edges, h.edges, sizeof(h.edges));
    if (range) { range[0] = u2f(h.res[0]); range[1] = u2f(h.res[1]); }
    if (scalars) {
        scalars[0] = s.fsq_lo; scalars[1] = s.fsq_hi; scalars[2] = s.gamma_sq; scalars[3] = s.fsq_min; scalars[4] = s.thr_cmp; scalars[5] = s.max_frob;
        scalars[6] = s.tri[which]; scalars[7] = s.otsu[which];
    }
    return NL_OK;
}

extern "C" int nl_vesselness_step(nl_ctx *c, float gamma_sq, float alpha_sq, float beta_sq, int use_thr, float thr,
                                  int64_t z0, int64_t z1, int64_t *mask_count, char *err, size_t errlen) {
    NL_ENTER(c);
    if (z0 < 0 && z1 < 0) { z0 = c->own_lo; z1 = c->own_hi; }
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    if (!c->have_spacing) return nl_fail(err, errlen, NL_ESTATE, "nl_vesselness_step before nl_hessian_stats");
    NL_JOIN_SIDE(c);
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(zero_small(d_cnt, 8, c->stream));
    VessP vp = make_vessp(c, gamma_sq, alpha_sq, beta_sq, use_thr, thr);
    vp.qcap = HM_REGION;
    c->spec_valid = 0;
    if (c->two_d) {
        ProfScope ps(c, "vesselness");
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 slot_words = c->nzl * c->ny * wpr;
        const int k_scale = c->mask_slots_used++;
        vp.have_prev = k_scale > 0;
        unsigned long long *cm = (unsigned long long *)c->m[0] + (i64)(k_scale & 1) * slot_words;
        const unsigned long long *pm = (unsigned long long *)c->m[0] + (i64)((k_scale + 1) & 1) * slot_words;
        if (vp.first) NL_HIP(hipMemsetAsync(c->f[c->i_vmax], 0, (size_t)c->n * 4, c->stream));
        vesselness2d_kernel<<<grid2d_rows((i64)wpr * 64, c->ny), 256, 0, c->stream>>>(gauss_cur(c), c->f[c->i_vmax], cm, pm, wpr, geom(c), hessp(c), vp, d_cnt);
        NL_CHECK_LAUNCH();
    } else {
        ProfScope ps(c, "vesselness");
        const int ntx = (int)((c->nx + HM_TX - 1) / HM_TX);
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 slot_words = c->nzl * c->ny * wpr;
        // cumulative h_mask (AND over the scales so far): scale k reads slot (k-1)&1 and writes slot k&1
        const int k_scale = c->mask_slots_used++;
        vp.have_prev = k_scale > 0;
        unsigned long long *cm = (unsigned long long *)c->m[0] + (i64)(k_scale & 1) * slot_words;
        const unsigned long long *pm = (unsigned long long *)c->m[0] + (i64)((k_scale + 1) & 1) * slot_words;
        // one launch covers as many Z chunks as the queue has regions for
        const i64 plane = c->ny * c->nx;
        const i64 planes_per_launch = (i64)c->vq_chunks * HM_ZCHUNK;
        const VQueue vq{(float4 *)c->d_vq, c->d_vq_count};
        if (vp.first)      // vesselness = zeros (filtering.py:807); only voxels alive in every mask are ever read again
            NL_HIP(hipMemsetAsync(c->f[c->i_vmax] + z0 * plane, 0, (size_t)(z1 - z0) * plane * 4, c->stream));
        const int rs = hv_rs(c);
        const int ty = rs ? 2 * rs : hm_ty();
        const int nty = (int)((c->ny + ty - 1) / ty);
        if (rs) vp.qcap = 2 * HM_REGION;
#define NL_LAUNCH_VESS(TYV, FASTV, HR)                                                                                    \
        hessian_g_kernel<1, TYV, FASTV><<<nblocks, HGCfg<TYV>::NT, HGCfg<TYV>::lds_bytes(), c->stream>>>(                 \
            gauss_cur(c), cm, pm, wpr, geom(c), HR, vp, vq, (int)za, (int)zb, ntx, nty, nullptr, d_cnt)
        for (i64 za = z0; za < z1; za += planes_per_launch) {
            const i64 zb = za + planes_per_launch < z1 ? za + planes_per_launch : z1;
            const int nzc = (int)((zb - za + HM_ZCHUNK - 1) / HM_ZCHUNK);
            const unsigned nblocks = (unsigned)(ntx * nty * nzc);
            if (rs) NL_HIP(nl_hv_launch(HvLaunch{1, rs, hv_fastv(c), nblocks, c->stream, gauss_cur(c), cm, pm, wpr, geom(c), hessp(c), vp, vq, (int)za, (int)zb,
                                                 ntx, nty, nullptr, d_cnt, nullptr}));
            else if (ty == 8) { if (c->fast_div) NL_LAUNCH_VESS(8, true, hessdv_fast(c)); else NL_LAUNCH_VESS(8, false, hessdv_exact(c)); }
            else { if (c->fast_div) NL_LAUNCH_VESS(16, true, hessdv_fast(c)); else NL_LAUNCH_VESS(16, false, hessdv_exact(c)); }
            NL_CHECK_LAUNCH();
            const unsigned nregions = nblocks * (unsigned)(rs ? rs : ty);
            vesselness_queue_kernel<false><<<resolve_grid((nregions + 3) / 4), 256, 0, c->stream>>>(vq.ent, vq.count, nregions, c->f[c->i_vmax], za * plane, vp,
                                                                                      nullptr, nullptr, wpr, (int)c->ny, (int)c->nx, za, nullptr, nullptr);
            NL_CHECK_LAUNCH();
        }
#undef NL_LAUNCH_VESS
        NL_CHECK_LAUNCH();
    }
    if (mask_count) {
        NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
        *mask_count = (int64_t)(*(unsigned long long *)c->h_small);
    }
    return NL_OK;
}

// ---- 2-D images (im_info.no_z) -----------------------------------------------------------------------------
extern "C" int nl_set_ndim(nl_ctx *c, int ndim, char *err, size_t errlen) {
    NL_ENTER(c);
    if (ndim != 2 && ndim != 3) return nl_fail(err, errlen, NL_EINVAL, "ndim must be 2 or 3");
    if (ndim == 2 && (c->nzl != 1 || c->gnz != 1)) return nl_fail(err, errlen, NL_EINVAL, "a 2-D context has exactly one plane");
    c->two_d = ndim == 2;
    c->fsq_cache_valid = 0;
    return NL_OK;
}

// One sigma of filtering.py:779-789.  w?2 / w?0 = scipy's order-2 / order-0 Gaussian kernels (2r+1 float64 weights,
// symmetric) for the Y and X axes; s2 = float32(sigma**2).
extern "C" int nl_log2d_step(nl_ctx *c, const double *wy2, const double *wy0, const double *wx2, const double *wx0, int r,
                             float s2, int first, int use_mask, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->two_d) return nl_fail(err, errlen, NL_ESTATE, "nl_log2d_step on a 3-D context");
    if (!wy2 || !wy0 || !wx2 || !wx0) return nl_fail(err, errlen, NL_EINVAL, "weights are NULL");
    NL_JOIN_SIDE(c);
    for (int k = 0; k < 4; ++k)
        if (!c->d_2d[k]) NL_HIP(hipMalloc((void **)&c->d_2d[k], (size_t)c->n * 4));
    GaussW gy2, gy0, gx2, gx0;
    int rc;
    if ((rc = fill_gw(gy2, wy2, r, err, errlen)) || (rc = fill_gw(gy0, wy0, r, err, errlen)) ||
        (rc = fill_gw(gx2, wx2, r, err, errlen)) || (rc = fill_gw(gx0, wx0, r, err, errlen))) return rc;
    const VolGeom v = geom(c);
    const dim3 blk(256, 1, 1);
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, 1);
    ProfScope ps(c, "log2d");
    const float *src = gauss_cur(c);
    float *t = c->d_2d[0], *A = c->d_2d[1], *B = c->d_2d[2], *lap = c->d_2d[3];
    // gaussian_laplace: second derivative along Y (then plain Gaussian along X), plus the one along X.  Both terms in one walk over
    // the image (the pair kernel of Markers' LoG: A = XY(gy2, gx0) + XY(gy0, gx2), the float32 sum the combine kernel forms) when
    // the radius allows; four one-axis passes otherwise.
    bool summed = false;
    if (gyx_tiled() && r >= 1 && r <= GM_MAX_R && r <= c->ny && r <= c->nx) {
        auto ws_of = [](const GaussW &g) { GaussWS w; for (int k = 0; k <= GM_MAX_R; ++k) w.w[k] = k <= g.r ? g.w[k] : 0.0; return w; };
        const GaussWS wya = ws_of(gy2), wxa = ws_of(gx0), wyb = ws_of(gy0), wxb = ws_of(gx2);
        const dim3 g2((unsigned)((c->nx + GYX_COLS - 1) / GYX_COLS), (unsigned)((c->ny + v.chunk - 1) / v.chunk), 1);
        const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
        const unsigned nb = g2.x * g2.y;
        switch (r) {
#define NL_L2D(RR) case RR: gauss_yx_dual_kernel<RR, false><<<nb, GYX_THREADS, 0, c->stream>>>(src, A, v, 0, 1, wya, wxa, wyb, wxb, vec4, (int)g2.x, (int)g2.y); break;
            NL_L2D(1) NL_L2D(2) NL_L2D(3) NL_L2D(4) NL_L2D(5) NL_L2D(6) NL_L2D(7) NL_L2D(8) NL_L2D(9) NL_L2D(10) NL_L2D(11) NL_L2D(12)
#undef NL_L2D
        }
        summed = true;
    } else {
        gauss_axis_kernel<1><<<grid, blk, 0, c->stream>>>(src, t, v, 0, 1, gy2);
        gauss_axis_kernel<2><<<grid, blk, 0, c->stream>>>(t, A, v, 0, 1, gx0);
        gauss_axis_kernel<1><<<grid, blk, 0, c->stream>>>(src, t, v, 0, 1, gy0);
        gauss_axis_kernel<2><<<grid, blk, 0, c->stream>>>(t, B, v, 0, 1, gx2);
    }
    const int wpr = (int)((c->nx + 63) / 64);
    const i64 slot_words = c->nzl * c->ny * wpr;
    const unsigned long long *mask = c->mask_slots_used > 0
        ? (const unsigned long long *)c->m[0] + (i64)((c->mask_slots_used - 1) & 1) * slot_words : nullptr;
    log2d_combine_kernel<<<grid2d_rows(c->nx, c->ny), 256, 0, c->stream>>>(A, summed ? nullptr : B, s2, use_mask ? mask : nullptr, wpr, v, first, lap);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

// filtering.py:792-795 and 928-930: scale the blob response to [0, 0.1] and take the maximum with NL_FIELD_FRANGI
// (call after nl_filter_finish).
extern "C" int nl_log2d_finish(nl_ctx *c, int64_t *n_positive, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->two_d || !c->d_2d[3]) return nl_fail(err, errlen, NL_ESTATE, "nl_log2d_finish before nl_log2d_step");
    unsigned int *res = (unsigned int *)c->d_small;
    unsigned long long *d_pos = (unsigned long long *)c->d_small + 2;
    NL_HIP(zero_small(res, 32, c->stream));
    ProfScope ps(c, "log2d");
    log2d_clip_max_kernel<<<grid1d(c->n), 256, 0, c->stream>>>(c->d_2d[3], c->n, res);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(c->h_small, res, 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    float mx;
    memcpy(&mx, c->h_small, 4);
    const float denom = mx + 1e-12f;                      // float32 + weak python float
    log2d_apply_kernel<<<grid1d(c->n), 256, 0, c->stream>>>(c->d_2d[3], denom, c->f[c->i_vmax], c->n, d_pos);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(c->h_small, d_pos, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_positive) *n_positive = (int64_t)(*(unsigned long long *)c->h_small);
    return NL_OK;
}

extern "C" int nl_filter_finish(nl_ctx *c, int64_t z0, int64_t z1, int64_t *n_positive, char *err, size_t errlen) {
    NL_ENTER(c);
    if (z0 < 0 && z1 < 0) { z0 = c->own_lo; z1 = c->own_hi; }
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    NL_JOIN_SIDE(c);
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(zero_small(d_cnt, 8, c->stream));
    const i64 plane = c->ny * c->nx;
    if (c->mask_slots_used == 0)       // every scale was skipped: vesselness was never written
        NL_HIP(hipMemsetAsync(c->f[c->i_vmax] + z0 * plane, 0, (size_t)(z1 - z0) * plane * 4, c->stream));
    {
        ProfScope ps(c, "finish");
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 quads = (z1 - z0) * c->ny * ((c->nx + 3) / 4);
        const i64 slot_words = c->nzl * c->ny * wpr;
        const int last = c->mask_slots_used > 0 ? ((c->mask_slots_used - 1) & 1) : 0;      // the slot of the last scale = AND of all
        finish_kernel<<<grid1d(quads, 256, 256 * 16), 256, 0, c->stream>>>(c->f[c->i_vmax], (const unsigned long long *)c->m[0] + last * slot_words,
                                                               c->mask_slots_used > 0 ? 1 : 0, slot_words, wpr, geom(c), z0, z1, c->own_lo, c->own_hi, d_cnt);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_positive) *n_positive = (int64_t)(*(unsigned long long *)c->h_small);
    c->frangi_ready = 1;
    return NL_OK;
}

// filtering.py:931-932, 969-1000: `_remove_edges` on the resident `vesselness * masks` frame (after nl_filter_finish),
// owned planes; *n_positive = values > 0 left on them.
extern "C" int nl_remove_edges(nl_ctx *c, int margin, int64_t *n_positive, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_JOIN_SIDE(c);
    if (!c->frangi_ready) return nl_fail(err, errlen, NL_ESTATE, "nl_remove_edges before a Frangi frame exists");
    if (margin < 0) return nl_fail(err, errlen, NL_EINVAL, "margin %d is negative", margin);
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(zero_small(d_cnt, 8, c->stream));
    {
        ProfScope ps(c, "finish");
        // the same planes nl_filter_finish materialises: nl_mask_volume thresholds and opens own +- 2, so the ghost planes
        // of a Z slab must lose their edge rows too (each plane is trimmed on its own: no cross-plane dependency)
        const i64 r0 = c->own_lo - 2 > 0 ? c->own_lo - 2 : 0, r1 = c->own_hi + 2 < c->nzl ? c->own_hi + 2 : c->nzl;
        remove_edges_kernel<<<(unsigned)(r1 - r0), 256, 0, c->stream>>>(c->f[c->i_vmax], geom(c), r0, margin, c->own_lo, c->own_hi, d_cnt);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_positive) *n_positive = (int64_t)(*(unsigned long long *)c->h_small);
    return NL_OK;
}

// filtering.py:964-966 on the finished frame; thr_dev != NULL: the threshold is read from device memory (nl_mask_volume_dev)
static int mask_volume_enqueue(nl_ctx *c, float thr, const float *thr_dev, int *dst_out, char *err, size_t errlen) {
    // result goes to a free gauss volume, which then becomes the Frangi volume
    int dst = -1;
    for (int k = 0; k < 3; ++k) if (k != c->i_gauss) { dst = k; break; }
    *dst_out = dst;
    ProfScope ps(c, "mask_volume");
    const int wpr = (int)((c->nx + 63) / 64);
    const VolGeom v = geom(c);
    // planes whose bits exist: own +-2 clipped to the slab (ghost planes beyond a true face do not exist)
    const i64 m0 = c->own_lo - 2 > 0 ? c->own_lo - 2 : 0, m1 = c->own_hi + 2 < c->nzl ? c->own_hi + 2 : c->nzl;
    const i64 e0 = c->own_lo - 1 > 0 ? c->own_lo - 1 : 0, e1 = c->own_hi + 1 < c->nzl ? c->own_hi + 1 : c->nzl;
    unsigned long long *bM = (unsigned long long *)c->m[1], *bE = (unsigned long long *)c->m[2], *bD = (unsigned long long *)c->m[0];
    rl_threshold_pack_kernel<<<grid1d((m1 - m0) * c->ny * 64, 256, (i64)1 << 22), 256, 0, c->stream>>>(
        c->f[c->i_vmax] + m0 * c->ny * c->nx, nullptr, bM + m0 * c->ny * wpr, 1, thr, (int)c->nx, (m1 - m0) * c->ny, wpr, thr_dev);
    NL_CHECK_LAUNCH();
    bits_morph6_kernel<0><<<(unsigned)(((e1 - e0) * c->ny * wpr + 255) / 256), 256, 0, c->stream>>>(bM, bE, v, wpr, e0, e1, c->two_d);
    NL_CHECK_LAUNCH();
    bits_morph6_kernel<1><<<(unsigned)(((c->own_hi - c->own_lo) * c->ny * wpr + 255) / 256), 256, 0, c->stream>>>(bE, bD, v, wpr, c->own_lo, c->own_hi, c->two_d);
    NL_CHECK_LAUNCH();
    apply_bits_kernel<<<grid1d((c->own_hi - c->own_lo) * c->ny * ((c->nx + 3) / 4), 256, 256 * 32), 256, 0, c->stream>>>(
        c->f[c->i_vmax], bD, c->f[dst], v, wpr, c->own_lo, c->own_hi);
    NL_CHECK_LAUNCH();
    return NL_OK;
}
static void mask_volume_commit(nl_ctx *c, int dst) {       // swap roles: the old vmax volume joins the gauss ping-pong set
    float *tmp = c->f[c->i_vmax];
    c->f[c->i_vmax] = c->f[dst];
    c->f[dst] = tmp;
}

extern "C" int nl_mask_volume(nl_ctx *c, float thr, char *err, size_t errlen) {
    NL_ENTER(c);
    int dst, rc;
    if ((rc = mask_volume_enqueue(c, thr, nullptr, &dst, err, errlen))) return rc;
    mask_volume_commit(c, dst);
    return NL_OK;
}

// The kernels of the fused epilogue; thr_dev != NULL: the threshold is read from device memory (nl_tail_enqueue).
// Writes the masked frame into the free volume *dst_out; the caller commits it (swap with i_vmax) or not.
static int mask_volume_fused_enqueue(nl_ctx *c, float thr, const float *thr_dev, unsigned long long *d_cnt, int *dst_out, char *err, size_t errlen) {
    int dst = -1;
    for (int k = 0; k < 3; ++k) if (k != c->i_gauss) { dst = k; break; }
    *dst_out = dst;
    NL_HIP(zero_small(d_cnt, 8, c->stream));
    ProfScope ps(c, "mask_volume");
    const int wpr = (int)((c->nx + 63) / 64);
    const VolGeom v = geom(c);
    const i64 slot_words = c->nzl * c->ny * wpr;
    const int last = (c->mask_slots_used - 1) & 1;
    const unsigned long long *alive = (const unsigned long long *)c->m[0] + (i64)last * slot_words;
    // the threshold bits may not overwrite the mask slot they are computed from: m[1] / m[2] hold them, the free
    // slot of m[0] takes the opened mask
    const i64 m0 = c->own_lo - 2 > 0 ? c->own_lo - 2 : 0, m1 = c->own_hi + 2 < c->nzl ? c->own_hi + 2 : c->nzl;
    const i64 e0 = c->own_lo - 1 > 0 ? c->own_lo - 1 : 0, e1 = c->own_hi + 1 < c->nzl ? c->own_hi + 1 : c->nzl;
    unsigned long long *bM = (unsigned long long *)c->m[1], *bE = (unsigned long long *)c->m[2];
    unsigned long long *bD = (unsigned long long *)c->m[0] + (i64)(last ^ 1) * slot_words;
    pack_masked_kernel<<<grid1d((m1 - m0) * c->ny * 64, 256, 256 * 16), 256, 0, c->stream>>>(    // one atomic per workgroup: small grid
        c->f[c->i_vmax], alive, bM, thr, thr_dev, (int)c->nx, m0 * c->ny, m1 * c->ny, wpr, c->own_lo * c->ny, c->own_hi * c->ny, d_cnt);
    NL_CHECK_LAUNCH();
    bits_morph6_kernel<0><<<(unsigned)(((e1 - e0) * c->ny * wpr + 255) / 256), 256, 0, c->stream>>>(bM, bE, v, wpr, e0, e1, c->two_d);
    NL_CHECK_LAUNCH();
    bits_morph6_kernel<1><<<(unsigned)(((c->own_hi - c->own_lo) * c->ny * wpr + 255) / 256), 256, 0, c->stream>>>(bE, bD, v, wpr, c->own_lo, c->own_hi, c->two_d);
    NL_CHECK_LAUNCH();
    apply_bits_pos_kernel<<<grid1d(((c->own_hi - c->own_lo) * c->ny + APPLY_ROWS - 1) / APPLY_ROWS * 64, 256, (i64)1 << 22), 256, 0, c->stream>>>(    // a wave takes APPLY_ROWS rows
        c->f[c->i_vmax], bD, c->f[dst], v, wpr, c->own_lo, c->own_hi);
    NL_CHECK_LAUNCH();
    // a fused communicator: the count comes back GLOBAL (one collective on the stream instead of a host-level all-reduce behind the call)
    if (fused(c)) NL_NCCL(rccl().AllReduce(d_cnt, d_cnt, 1, ncclUint64, ncclSum, (ncclComm_t)c->comm, c->stream));
    return NL_OK;
}
static void mask_volume_fused_commit(nl_ctx *c, int dst) {
    float *tmp = c->f[c->i_vmax];
    c->f[c->i_vmax] = c->f[dst];
    c->f[dst] = tmp;
    c->frangi_ready = 1;
    // valid as long as only NL_KEEP_SUPPORT entry points follow; on a slab it describes the OWNED planes (all nl_slab_label_pack reads)
    c->d_support = (const unsigned long long *)c->m[0] + (i64)(((c->mask_slots_used - 1) & 1) ^ 1) * (c->nzl * c->ny * (i64)((c->nx + 63) / 64));
    c->support_epoch = c->epoch.load();
}

extern "C" int nl_mask_volume_fused(nl_ctx *c, float thr, int64_t *n_positive, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->mask_slots_used == 0) return nl_fail(err, errlen, NL_ESTATE, "nl_mask_volume_fused before any scale was evaluated");
    NL_JOIN_SIDE(c);
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    int dst, rc;
    if ((rc = mask_volume_fused_enqueue(c, thr, nullptr, d_cnt, &dst, err, errlen))) return rc;
    NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_positive) *n_positive = (int64_t)(*(unsigned long long *)c->h_small);
    mask_volume_fused_commit(c, dst);
    return NL_OK;
}

// ---- the frame's epilogue without a host decision in it (percentile.inc) ------------------------------------------------------
// layout of d_pct: [PctRec 64 B][sample counter 4 B, pad][voxel counter 8 B][hist 2 x PCT_BINS u32]
static int pct_buffers(nl_ctx *c, char *err, size_t errlen) {
    if (!c->d_pct) {
        NL_HIP(hipMalloc(&c->d_pct, 128 + (size_t)2 * PCT_BINS * 4));
        NL_HIP(hipHostMalloc(&c->h_pct, 128, hipHostMallocDefault));
    }
    return NL_OK;
}
// numpy.percentile(samples[0 .. *d_n), q) -> rec->thr, selected on the device (all-reduced across a fused communicator)
static int pct_enqueue(nl_ctx *c, const float *samples, const unsigned int *d_n, i64 max_n, float q, char *err, size_t errlen) {
    PctRec *rec = (PctRec *)c->d_pct;
    unsigned int *hist = (unsigned int *)((char *)c->d_pct + 128);
    NL_HIP(hipMemcpyAsync(&rec->n, d_n, 4, hipMemcpyDeviceToDevice, c->stream));
    NL_HIP(hipMemcpyAsync(&rec->n_local, d_n, 4, hipMemcpyDeviceToDevice, c->stream));
    if (fused(c)) NL_NCCL(rccl().AllReduce(&rec->n, &rec->n, 1, ncclUint32, ncclSum, (ncclComm_t)c->comm, c->stream));
    ProfScope ps(c, "sample");
    pct_begin_kernel<<<1, 64, 0, c->stream>>>(rec, q);
    const unsigned grid = grid1d(max_n > 0 ? max_n : 1, 256, 256 * 256);
#define NL_PCT_LEVEL(L)                                                                                            \
    NL_HIP(hipMemsetAsync(hist, 0, (size_t)2 * PCT_BINS * 4, c->stream));                                           \
    pct_hist_kernel<L><<<grid, 256, 0, c->stream>>>(samples, d_n, rec, hist);                                       \
    if (fused(c)) NL_NCCL(rccl().AllReduce(hist, hist, (size_t)2 * PCT_BINS, ncclUint32, ncclSum, (ncclComm_t)c->comm, c->stream)); \
    pct_select_kernel<L><<<1, 256, 0, c->stream>>>(rec, hist);
    NL_PCT_LEVEL(0) NL_PCT_LEVEL(1) NL_PCT_LEVEL(2)
#undef NL_PCT_LEVEL
    NL_CHECK_LAUNCH();
    return NL_OK;
}

/* filtering.py:926 + 952-967 in one enqueue, no host decision inside: the positive lattice samples of `vesselness * masks`
   (strides sz, sy, sx), their q-th percentile (numpy's float32 'linear' rule, selected on the device), the percentile mask, its
   opening and the product.  Nothing is committed yet: nl_tail_finish waits, reports and (commit != 0) makes the result the frame. */
extern "C" int nl_tail_enqueue(nl_ctx *c, int64_t sz, int64_t sy, int64_t sx, double q, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->mask_slots_used == 0) return nl_fail(err, errlen, NL_ESTATE, "nl_tail_enqueue before any scale was evaluated");
    if (c->two_d) return nl_fail(err, errlen, NL_EINVAL, "nl_tail_enqueue is the 3-D epilogue");
    int rc;
    if ((rc = pct_buffers(c, err, errlen))) return rc;
    Lattice L; FieldSrc fs;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, NL_FIELD_VESSELNESS, fs, err, errlen))) return rc;       // (joins the side stream)
    const i64 total = L.cz * L.cy * L.cx;
    if (total > c->n) return nl_fail(err, errlen, NL_EINVAL, "lattice larger than the volume");
    float *stage = c->f[(c->i_gauss + 1) % 3];
    if (stage == c->f[c->i_vmax]) return nl_fail(err, errlen, NL_ESTATE, "no free volume for the samples");
    unsigned int *d_n = (unsigned int *)((char *)c->d_pct + 64);
    unsigned long long *d_cnt = (unsigned long long *)((char *)c->d_pct + 72);
    NL_HIP(zero_small(d_n, 4, c->stream));
    if (total > 0) {
        ProfScope ps(c, "sample");
        sample_gather_pos_kernel<<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(fs, geom(c), L, stage, d_n);
        NL_CHECK_LAUNCH();
    }
    if ((rc = pct_enqueue(c, stage, d_n, total, (float)q, err, errlen))) return rc;
    // the samples sit in a volume the epilogue may write (dst): the selection above is complete before it does (stream order)
    int dst;
    if ((rc = mask_volume_fused_enqueue(c, 0.0f, &((PctRec *)c->d_pct)->thr, d_cnt, &dst, err, errlen))) return rc;
    NL_HIP(hipMemcpyAsync(c->h_pct, c->d_pct, 80, hipMemcpyDeviceToHost, c->stream));
    c->tail_pending = 1; c->tail_dst = dst;
    return NL_OK;
}

extern "C" int nl_tail_finish(nl_ctx *c, int commit, int64_t *n_samples, float *a, float *b, float *gamma, float *thr, int64_t *n_positive,
                              char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->tail_pending) return nl_fail(err, errlen, NL_ESTATE, "nl_tail_finish without nl_tail_enqueue");
    NL_HIP(hipStreamSynchronize(c->stream));
    c->tail_pending = 0;
    const PctRec *rec = (const PctRec *)c->h_pct;
    if (n_samples) *n_samples = rec->n;
    if (a) *a = rec->a;
    if (b) *b = rec->b;
    if (gamma) *gamma = rec->gamma;
    if (thr) *thr = rec->thr;
    if (n_positive) *n_positive = (int64_t)(*(const unsigned long long *)((const char *)c->h_pct + 72));
    if (commit && rec->n > 0) mask_volume_fused_commit(c, c->tail_dst);
    return NL_OK;
}

/* _mask_volume (filtering.py:952-967) on the finished frame with the percentile selected on the device: the positive lattice samples
   of the Frangi frame, their q-th percentile, `frame > thr`, the opening and the product, one wait.  n_samples = 0: nothing was
   changed (the reference returns the frame as it is).  The epilogue of 2-D images, of remove_edges runs, of slabs without the
   fused epilogue; 3-D frames normally take nl_tail_enqueue. */
extern "C" int nl_mask_volume_dev(nl_ctx *c, int64_t sz, int64_t sy, int64_t sx, double q, int64_t *n_samples, float *a, float *b, float *gamma,
                                  float *thr, char *err, size_t errlen) {
    NL_ENTER(c);
    int rc;
    if ((rc = pct_buffers(c, err, errlen))) return rc;
    Lattice L; FieldSrc fs;
    if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
    if ((rc = make_field(c, NL_FIELD_FRANGI, fs, err, errlen))) return rc;
    const i64 total = L.cz * L.cy * L.cx;
    if (total > c->n) return nl_fail(err, errlen, NL_EINVAL, "lattice larger than the volume");
    float *stage = nullptr;                                  // a free volume: neither the frame nor the current Gaussian
    for (int k = 0; k < 3; ++k) if (k != c->i_gauss && c->f[k] != c->f[c->i_vmax]) stage = c->f[k];
    if (!stage) return nl_fail(err, errlen, NL_ESTATE, "no free volume for the samples");
    unsigned int *d_n = (unsigned int *)((char *)c->d_pct + 64);
    NL_HIP(zero_small(d_n, 4, c->stream));
    if (total > 0) {
        ProfScope ps(c, "sample");
        sample_gather_pos_kernel<<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(fs, geom(c), L, stage, d_n);
        NL_CHECK_LAUNCH();
    }
    if ((rc = pct_enqueue(c, stage, d_n, total, (float)q, err, errlen))) return rc;
    int dst;
    if ((rc = mask_volume_enqueue(c, 0.0f, &((PctRec *)c->d_pct)->thr, &dst, err, errlen))) return rc;
    NL_HIP(hipMemcpyAsync(c->h_pct, c->d_pct, 64, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    const PctRec *rec = (const PctRec *)c->h_pct;
    if (n_samples) *n_samples = rec->n;
    if (a) *a = rec->a;
    if (b) *b = rec->b;
    if (gamma) *gamma = rec->gamma;
    if (thr) *thr = rec->thr;
    if (rec->n > 0) mask_volume_commit(c, dst);
    return NL_OK;
}

/* numpy.percentile(values, q) by the device's selection (tests): values > 0, n < the context's voxel count */
extern "C" int nl_debug_percentile(nl_ctx *c, const float *values, int64_t n, double q, float *thr, float *a, float *b, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!values || n < 1 || n > c->n) return nl_fail(err, errlen, NL_EINVAL, "bad sample array");
    int rc;
    if ((rc = pct_buffers(c, err, errlen))) return rc;
    float *stage = c->f[(c->i_gauss + 1) % 3];
    unsigned int *d_n = (unsigned int *)((char *)c->d_pct + 64);
    const unsigned int un = (unsigned int)n;
    NL_HIP(hipMemcpyAsync(stage, values, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipMemcpyAsync(d_n, &un, 4, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    const int keep = c->fuse_reduce; c->fuse_reduce = 0;                 // a local array: no collective
    rc = pct_enqueue(c, stage, d_n, n, (float)q, err, errlen);
    c->fuse_reduce = keep;
    if (rc) return rc;
    NL_HIP(hipMemcpyAsync(c->h_pct, c->d_pct, 64, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    const PctRec *rec = (const PctRec *)c->h_pct;
    if (thr) *thr = rec->thr;
    if (a) *a = rec->a;
    if (b) *b = rec->b;
    return NL_OK;
}

static int store_planes(nl_ctx *c, const void *dev_base, void *host, size_t elem, int64_t z0, int64_t z1, char *err, size_t errlen) {
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    const i64 plane = c->ny * c->nx;
    NL_HIP(hipMemcpyAsync(host, (const char *)dev_base + (size_t)z0 * plane * elem, (size_t)(z1 - z0) * plane * elem,
                          hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

extern "C" int nl_filter_store(nl_ctx *c, float *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_SUPPORT(c);
    return store_planes(c, c->f[c->i_vmax], host, 4, z0, z1, err, errlen);
}
extern "C" int nl_gauss_store(nl_ctx *c, float *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    return store_planes(c, gauss_cur(c), host, 4, z0, z1, err, errlen);
}

// ------------------------------------------------------------------------------ slab helpers ----
static float *field_ptr(nl_ctx *c, int field) {
    if (field == NL_FIELD_GAUSS) return gauss_cur(c);
    if (field == NL_FIELD_FRANGI) return c->f[c->i_vmax];
    return nullptr;
}

extern "C" int nl_planes_get(nl_ctx *c, int field, int64_t z0, int64_t z1, float *host, char *err, size_t errlen) {
    NL_ENTER(c);
    float *p = field_ptr(c, field);
    if (!p) return nl_fail(err, errlen, NL_EINVAL, "nl_planes_get: field %d has no volume", field);
    return store_planes(c, p, host, 4, z0, z1, err, errlen);
}

extern "C" int nl_planes_put(nl_ctx *c, int field, int64_t z0, int64_t z1, const float *host, char *err, size_t errlen) {
    NL_ENTER(c);
    float *p = field_ptr(c, field);
    c->fsq_cache_valid = 0;
    if (!p || !host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "nl_planes_put: bad field or plane range");
    const i64 plane = c->ny * c->nx;
    NL_HIP(hipMemcpyAsync(p + z0 * plane, host, (size_t)(z1 - z0) * plane * 4, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}


extern "C" int nl_comm_unique_id(char *id128, char *err, size_t errlen) {
    if (!id128) return nl_fail(err, errlen, NL_EINVAL, "id buffer is NULL");
    ncclUniqueId id;
    {
        ncclResult_t r_ = rccl().GetUniqueId(&id);
        if (r_ != ncclSuccess) return nl_fail(err, errlen, NL_ECOMM, "ncclGetUniqueId: %s", rccl().GetErrorString(r_));
    }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return NL_OK;
}

// An id of the loopback transport (loopback.inc): `world` contexts of THIS process, one host thread per rank, exchange
// through device-to-device copies on the very streams, with the very offsets and counts RCCL would be given.
extern "C" int nl_comm_loopback_id(char *id128, char *err, size_t errlen) {
    if (!id128) return nl_fail(err, errlen, NL_EINVAL, "id buffer is NULL");
    lb::get_unique_id(id128);
    return NL_OK;
}

// RCCL communicators outlive their context: a context that closes hands its communicators to a per-process pool, and the next
// context of the same (device, world, rank, role) takes them from there instead of creating new ones (every rank does the
// same, so the pool's state is the same everywhere; the id the caller brings is then not used -- the CONSTRAINT: the ranks of a
// job open and close their contexts in the same order, which the SPMD stage classes do; a rank that restarts alone, or a context
// that failed in a collective (its communicators are destroyed instead, `comm_poisoned`), needs fresh ids on every rank).  Why: a process in which an RCCL
// communicator has been destroyed -- or created beside an older one -- runs every later slab step 9-18 % slower (measured at
// world 1 on a 128 x 2048 x 2048 slab: 29.9 -> 32.7 ms synchronous, 30.1 -> 35.3 ms with the device chain; with the earlier
// communicators neither destroyed nor replaced: 30.1), and the stages of a run (Filter, then Label) each open a context.
// Loopback communicators are plain host objects and are destroyed with their context.
struct PooledComm { int device, world, rank, role; ncclComm_t comm; };
static std::mutex g_comm_pool_mu;
static std::vector<PooledComm> g_comm_pool;
static ncclComm_t comm_pool_take(int device, int world, int rank, int role) {
    std::lock_guard<std::mutex> lk(g_comm_pool_mu);
    for (size_t i = 0; i < g_comm_pool.size(); ++i) {
        const PooledComm &p = g_comm_pool[i];
        if (p.device == device && p.world == world && p.rank == rank && p.role == role) {
            ncclComm_t c = p.comm;
            g_comm_pool.erase(g_comm_pool.begin() + (long)i);
            return c;
        }
    }
    return nullptr;
}
static void comm_release(nl_ctx *c, void *comm, int role) {
    if (!comm) return;
    // a communicator whose context saw a collective fail may be out of step with its peers: never hand it to a later context
    if (lb::is_ours(comm) || c->comm_poisoned || getenv("NELLIE_DESTROY_COMMS")) { rccl().CommDestroy((ncclComm_t)comm); return; }
    std::lock_guard<std::mutex> lk(g_comm_pool_mu);
    g_comm_pool.push_back(PooledComm{c->device, c->world, c->rank, role, (ncclComm_t)comm});
}
static int comm_acquire(nl_ctx *c, int world, int rank, const char *id128, int role, ncclComm_t *out, char *err, size_t errlen) {
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    if (!lb::is_loopback_id(id128)) {
        ncclComm_t pooled = comm_pool_take(c->device, world, rank, role);
        if (pooled) { *out = pooled; return NL_OK; }
    }
    NL_NCCL(rccl().CommInitRank(out, world, id, rank));
    return NL_OK;
}

extern "C" int nl_comm_init(nl_ctx *c, int world, int rank, const char *id128, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!id128 || world < 1 || rank < 0 || rank >= world) return nl_fail(err, errlen, NL_EINVAL, "bad communicator arguments");
    if (c->comm) return nl_fail(err, errlen, NL_ESTATE, "the context already has a communicator");
    ncclComm_t comm;
    int rc = comm_acquire(c, world, rank, id128, 1, &comm, err, errlen);
    if (rc) return rc;
    c->comm = comm; c->world = world; c->rank = rank;
    return NL_OK;
}

// Ghost-plane exchange with the Z neighbours over RCCL (xGMI).  The `depth` owned planes that start `offset` planes inside
// this rank's boundary go to the neighbour's ghost planes at the same distance from the interface, and the neighbours'
// come into ours: low side  send [own_lo + offset, +depth)  recv [own_lo - offset - depth, own_lo - offset),
//                 high side send [own_hi - offset - depth, own_hi - offset)  recv [own_hi + offset, +depth).
// offset 0 = the classic halo.  Asynchronous on the context stream; with `async` != 0 (and a second communicator,
// nl_comm_init2) it runs on a stream and a communicator of its own, ordered after everything submitted so far, and the next
// nl_gauss_step waits for it: the exchange for cascade step s+1 then travels while scale s is being evaluated.
static int halo_exchange_impl(nl_ctx *c, int field, int64_t offset, int64_t depth, int async, char *err, size_t errlen) {
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_halo_exchange before nl_comm_init");
    c->fsq_cache_valid = 0;
    float *p = field_ptr(c, field);
    if (!p) return nl_fail(err, errlen, NL_EINVAL, "nl_halo_exchange: field %d has no volume", field);
    const i64 plane = c->ny * c->nx;
    const bool has_lo = c->rank > 0, has_hi = c->rank + 1 < c->world;
    if (depth < 1 || offset < 0 || offset + depth > c->own_hi - c->own_lo || (has_lo && offset + depth > c->own_lo) ||
        (has_hi && offset + depth > c->nzl - c->own_hi))
        return nl_fail(err, errlen, NL_EINVAL, "halo planes [%lld, %lld) from the interface do not fit the slab (own %lld, ghosts %lld/%lld)", (i64)offset,
                       (i64)(offset + depth), (i64)(c->own_hi - c->own_lo), (i64)c->own_lo, (i64)(c->nzl - c->own_hi));
    const bool side = async && c->comm2;
    ncclComm_t comm = (ncclComm_t)(side ? c->comm2 : c->comm);
    hipStream_t st = c->stream;
    if (side) {
        if (!c->xstream) {
            NL_HIP(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
            NL_HIP(hipEventCreateWithFlags(&c->ev_x_main, hipEventDisableTiming));
            NL_HIP(hipEventCreateWithFlags(&c->ev_x_done, hipEventDisableTiming));
        }
        if (c->halo_pending) NL_HIP(hipStreamWaitEvent(c->stream, c->ev_x_done, 0));      // one exchange in flight at a time
        NL_HIP(hipEventRecord(c->ev_x_main, c->stream));
        NL_HIP(hipStreamWaitEvent(c->xstream, c->ev_x_main, 0));
        st = c->xstream;
    }
    ProfScope ps(c, "halo", st);
    NL_NCCL(rccl().GroupStart());
    if (has_lo) {
        NL_NCCL(rccl().Send(p + (c->own_lo + offset) * plane, (size_t)(depth * plane), ncclFloat, c->rank - 1, comm, st));
        NL_NCCL(rccl().Recv(p + (c->own_lo - offset - depth) * plane, (size_t)(depth * plane), ncclFloat, c->rank - 1, comm, st));
    }
    if (has_hi) {
        NL_NCCL(rccl().Send(p + (c->own_hi - offset - depth) * plane, (size_t)(depth * plane), ncclFloat, c->rank + 1, comm, st));
        NL_NCCL(rccl().Recv(p + (c->own_hi + offset) * plane, (size_t)(depth * plane), ncclFloat, c->rank + 1, comm, st));
    }
    NL_NCCL(rccl().GroupEnd());
    if (side) {
        NL_HIP(hipEventRecord(c->ev_x_done, c->xstream));
        c->halo_pending = 1;
    }
    return NL_OK;
}
extern "C" int nl_halo_exchange(nl_ctx *c, int field, int64_t depth, char *err, size_t errlen) {
    NL_ENTER(c);
    return halo_exchange_impl(c, field, 0, depth, 0, err, errlen);
}
extern "C" int nl_halo_exchange_at(nl_ctx *c, int field, int64_t offset, int64_t depth, int async, char *err, size_t errlen) {
    NL_ENTER(c);
    return halo_exchange_impl(c, field, offset, depth, async, err, errlen);
}
// second communicator (its own unique id): carries the asynchronous ghost-plane exchanges, so that they do not serialise
// with the reductions of the first one
extern "C" int nl_comm_init2(nl_ctx *c, int world, int rank, const char *id128, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!id128 || world != c->world || rank != c->rank || !c->comm) return nl_fail(err, errlen, NL_EINVAL, "nl_comm_init2 needs the world / rank of nl_comm_init");
    if (c->comm2) return nl_fail(err, errlen, NL_ESTATE, "the context already has a second communicator");
    ncclComm_t comm;
    int rc = comm_acquire(c, world, rank, id128, 2, &comm, err, errlen);
    if (rc) return rc;
    c->comm2 = comm;
    return NL_OK;
}

// Small all-reduce of host values through RCCL: dtype 0 = int64, 1 = float32; op 0 = sum, 1 = min, 2 = max.
extern "C" int nl_allreduce(nl_ctx *c, void *host_inout, int64_t count, int dtype, int op, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_SUPPORT(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_allreduce before nl_comm_init");
    const size_t es = dtype == 0 ? 8 : 4;
    if (!host_inout || count < 1 || (size_t)count * es > (1 << 15) || dtype < 0 || dtype > 1 || op < 0 || op > 2)
        return nl_fail(err, errlen, NL_EINVAL, "bad all-reduce arguments");
    memcpy(c->h_small, host_inout, (size_t)count * es);
    NL_HIP(hipMemcpyAsync(c->d_small, c->h_small, (size_t)count * es, hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
    NL_NCCL(rccl().AllReduce(c->d_small, c->d_small, (size_t)count, dtype == 0 ? ncclInt64 : ncclFloat, ops[op], (ncclComm_t)c->comm, c->stream));
    NL_HIP(hipMemcpyAsync(c->h_small, c->d_small, (size_t)count * es, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    memcpy(host_inout, c->h_small, (size_t)count * es);
    return NL_OK;
}

extern "C" int nl_comm_fuse(nl_ctx *c, int on, char *err, size_t errlen) {
    NL_ENTER(c);
    if (on && !c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_comm_fuse before nl_comm_init");
    c->fuse_reduce = on ? 1 : 0;
    return NL_OK;
}

// The positive samples of ALL ranks in one call with one wait (round 4): every rank compacts its samples into a block
// [count | samples ...] of block_items + 1 floats (block_items: a bound on any rank's sample points that the callers derive from
// the global geometry, identical everywhere), the blocks are all-gathered over RCCL on the context stream and land in page-locked
// memory.  mode 0: the lattice arr[::a, ::b, ::c] of `field` (filtering.py:348-363), mode 1: flat[a::b] (labelling.py:418-433).
// out receives the samples rank by rank, counts[r] how many rank r contributed.  Before: a download, then nl_allgather_var's two
// collectives with a wait each.
extern "C" int nl_positive_samples_world(nl_ctx *c, int field, int mode, int64_t a, int64_t b, int64_t cc, int64_t block_items,
                                         float *out, int64_t cap, int64_t *counts, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_SUPPORT(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_positive_samples_world before nl_comm_init");
    if (block_items < 0 || !counts || (mode != 0 && mode != 1)) return nl_fail(err, errlen, NL_EINVAL, "bad arguments");
    const int W = c->world;
    const size_t blk = (size_t)block_items + 1;                       // floats per rank
    const size_t need = blk * (size_t)(W + 1) * 4;
    if (need > c->ag_cap) {
        if (c->d_ag) hipFree(c->d_ag);
        c->d_ag = nullptr; c->ag_cap = 0;
        NL_HIP(hipMalloc(&c->d_ag, need + need / 2));
        c->ag_cap = need + need / 2;
    }
    if (blk * W * 4 > c->h_ag_cap) {
        if (c->h_ag) hipHostFree(c->h_ag);
        c->h_ag = nullptr; c->h_ag_cap = 0;
        NL_HIP(hipHostMalloc(&c->h_ag, blk * W * 4 * 3 / 2, hipHostMallocDefault));
        c->h_ag_cap = blk * W * 4 * 3 / 2;
    }
    float *d_send = (float *)c->d_ag, *d_recv = d_send + blk;
    NL_HIP(zero_small(d_send, 4, c->stream));
    int rc;
    i64 points = 0;
    if (mode == 0) {
        Lattice L; FieldSrc fs;
        if ((rc = make_lattice(c, a, b, cc, L, err, errlen))) return rc;
        if ((rc = make_field(c, field, fs, err, errlen))) return rc;
        if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
        points = L.cz * L.cy * L.cx;
        if (points > block_items) return nl_fail(err, errlen, NL_EINVAL, "%lld lattice points in this slab, block of %lld", (long long)points, (long long)block_items);
        if (points) {
            ProfScope ps(c, "sample");
            sample_gather_pos_kernel<<<(unsigned)((points + 255) / 256), 256, 0, c->stream>>>(fs, geom(c), L, d_send + 1, (unsigned int *)d_send);
            NL_CHECK_LAUNCH();
        }
    } else {
        if (b < 1 || a < 0) return nl_fail(err, errlen, NL_EINVAL, "bad offset/step");
        if (field != NL_FIELD_FRANGI && field != NL_FIELD_GAUSS) return nl_fail(err, errlen, NL_EINVAL, "flat sampling supports GAUSS/FRANGI");
        const i64 plane = c->ny * c->nx;
        const i64 g_begin = (c->gz0 + c->own_lo) * plane, g_end = (c->gz0 + c->own_hi) * plane;
        const i64 k0 = g_begin > a ? (g_begin - a + b - 1) / b : 0;
        const i64 k1 = g_end > a ? (g_end - a + b - 1) / b : 0;
        points = k1 > k0 ? k1 - k0 : 0;
        if (points > block_items) return nl_fail(err, errlen, NL_EINVAL, "%lld sample points in this slab, block of %lld", (long long)points, (long long)block_items);
        if (points) {
            const float *src = (field == NL_FIELD_FRANGI) ? c->f[c->i_vmax] : gauss_cur(c);
            ProfScope ps(c, "sample");
            flat_gather_pos_kernel<<<(unsigned)((points + 255) / 256), 256, 0, c->stream>>>(src, -c->gz0 * plane, a + k0 * b, b, points, d_send + 1, (unsigned int *)d_send);
            NL_CHECK_LAUNCH();
        }
    }
    {
        ProfScope ps(c, "halo");
        NL_NCCL(rccl().AllGather(d_send, d_recv, blk, ncclFloat, (ncclComm_t)c->comm, c->stream));
    }
    NL_HIP(hipMemcpyAsync(c->h_ag, d_recv, blk * W * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    i64 total = 0;
    const float *h = (const float *)c->h_ag;
    for (int r = 0; r < W; ++r) {
        unsigned int k; memcpy(&k, h + (size_t)r * blk, 4);
        if ((i64)k > block_items) return nl_fail(err, errlen, NL_ESTATE, "rank %d reports %u samples in a block of %lld", r, k, (long long)block_items);
        counts[r] = (int64_t)k;
        if (total + (i64)k > cap || (k && !out)) return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld too small", (long long)cap);
        if (k) memcpy(out + total, h + (size_t)r * blk + 1, (size_t)k * 4);
        total += (i64)k;
    }
    return NL_OK;
}

// Variable-size all-gather of host bytes (see include/nellie_amd.h).  Two collectives: the sizes, then the padded blocks.
extern "C" int nl_allgather_bytes(nl_ctx *c, const void *send, int64_t nbytes, void *recv, int64_t max_bytes, int64_t *bytes_of,
                                  char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_SUPPORT(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_allgather_bytes before nl_comm_init");
    if (nbytes < 0 || max_bytes < 1 || nbytes > max_bytes || !recv || !bytes_of || (nbytes && !send))
        return nl_fail(err, errlen, NL_EINVAL, "bad all-gather arguments");
    const int W = c->world;
    // sizes through the small scratch
    long long *hs = (long long *)c->h_small;
    hs[0] = nbytes;
    NL_HIP(hipMemcpyAsync(c->d_small, hs, 8, hipMemcpyHostToDevice, c->stream));
    NL_NCCL(rccl().AllGather(c->d_small, (char *)c->d_small + 64, 1, ncclInt64, (ncclComm_t)c->comm, c->stream));
    NL_HIP(hipMemcpyAsync(hs, (char *)c->d_small + 64, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    for (int r = 0; r < W; ++r) { bytes_of[r] = hs[r]; if (hs[r] > max_bytes) return nl_fail(err, errlen, NL_EINVAL, "rank %d sends %lld bytes, more than max_bytes = %lld", r, hs[r], (long long)max_bytes); }
    // blocks through a staging buffer that grows on demand
    const size_t need = (size_t)max_bytes * (size_t)(W + 1);
    if (need > c->ag_cap) {
        if (c->d_ag) hipFree(c->d_ag);
        c->d_ag = nullptr; c->ag_cap = 0;
        NL_HIP(hipMalloc(&c->d_ag, need));
        c->ag_cap = need;
    }
    char *d_send = (char *)c->d_ag, *d_recv = d_send + max_bytes;
    if (nbytes) NL_HIP(hipMemcpyAsync(d_send, send, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    NL_NCCL(rccl().AllGather(d_send, d_recv, (size_t)max_bytes, ncclChar, (ncclComm_t)c->comm, c->stream));
    NL_HIP(hipMemcpyAsync(recv, d_recv, (size_t)max_bytes * W, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// The same without a size negotiated by the caller: the block size is the largest of the gathered sizes, and the blocks land
// in a page-locked buffer the context owns (*recv, valid until the next call; rank r's block at r * *stride).
extern "C" int nl_allgather_var(nl_ctx *c, const void *send, int64_t nbytes, void **recv, int64_t *stride, int64_t *bytes_of,
                                char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_SUPPORT(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_allgather_var before nl_comm_init");
    if (nbytes < 0 || !recv || !stride || !bytes_of || (nbytes && !send)) return nl_fail(err, errlen, NL_EINVAL, "bad all-gather arguments");
    const int W = c->world;
    long long *hs = (long long *)c->h_small;
    hs[0] = nbytes;
    NL_HIP(hipMemcpyAsync(c->d_small, hs, 8, hipMemcpyHostToDevice, c->stream));
    NL_NCCL(rccl().AllGather(c->d_small, (char *)c->d_small + 64, 1, ncclInt64, (ncclComm_t)c->comm, c->stream));
    NL_HIP(hipMemcpyAsync(hs, (char *)c->d_small + 64, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    long long mx = 16;
    for (int r = 0; r < W; ++r) { bytes_of[r] = hs[r]; if (hs[r] > mx) mx = hs[r]; }
    mx = (mx + 15) & ~15ll;
    const size_t need = (size_t)mx * (size_t)(W + 1);
    if (need > c->ag_cap) {
        if (c->d_ag) hipFree(c->d_ag);
        c->d_ag = nullptr; c->ag_cap = 0;
        NL_HIP(hipMalloc(&c->d_ag, need + need / 2));          // head room: the tables of the next phase / frame differ a little
        c->ag_cap = need + need / 2;
    }
    if ((size_t)mx * W > c->h_ag_cap) {
        if (c->h_ag) hipHostFree(c->h_ag);
        c->h_ag = nullptr; c->h_ag_cap = 0;
        const size_t cap = (size_t)mx * W * 3 / 2;
        NL_HIP(hipHostMalloc(&c->h_ag, cap, hipHostMallocDefault));
        c->h_ag_cap = cap;
    }
    char *d_send = (char *)c->d_ag, *d_recv = d_send + mx;
    if (nbytes) NL_HIP(hipMemcpyAsync(d_send, send, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    NL_NCCL(rccl().AllGather(d_send, d_recv, (size_t)mx, ncclChar, (ncclComm_t)c->comm, c->stream));
    NL_HIP(hipMemcpyAsync(c->h_ag, d_recv, (size_t)mx * W, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    *recv = c->h_ag; *stride = mx;
    return NL_OK;
}

// ---------------------------------------------------------------------------------- Markers ------
// Stage after Label (nellie/segmentation/mocap_marking.py:648-703, use_im = 'distance', 3-D).  Volumes:
//   f[0] distance (float32, the stage product), f[1] / f[2] scratch (squared distances, then the Z-filtered volume and
//   the Laplacian accumulator), f[3] best response over the scales; the float32 intensities live in the eigen queue
//   buffer (idle outside Filter); m[1] = mask bits (labels > 0), m[2] = border bits, m[0] = peak bits | kept bits.
static float *mk_intensity(nl_ctx *c) { return c->d_vq; }

extern "C" int nl_markers_begin(nl_ctx *c, const int *labels_host, const void *intensity_host, int dtype, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_JOIN_SIDE(c);
    if (c->own_lo != 0 || c->own_hi != c->nzl || c->gnz != c->nzl) return nl_fail(err, errlen, NL_EINVAL, "the Markers stage runs on a whole volume (no Z slabs yet)");
    const int wpr = (int)((c->nx + 63) / 64);
    const i64 nrows = c->nzl * c->ny;
    if (vq_alloc_entries(c->nzl, c->ny, c->nx) * 32 < c->n * 4) return nl_fail(err, errlen, NL_ENOMEM, "scratch too small for the intensity volume [out of memory]");
    ProfScope ps(c, "markers_begin");
    // mask bits first: the device labels may live in one of the volumes reused below
    const int *lab = nullptr;
    if (labels_host) {
        NL_HIP(hipMemcpyAsync(c->f[1], labels_host, (size_t)c->n * 4, hipMemcpyHostToDevice, c->stream));
        lab = (const int *)c->f[1];
    } else {
        if (c->i_labels < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_markers_begin(labels = NULL) before nl_label_run");
        lab = (const int *)c->f[c->i_labels];
    }
    mk_pack_labels_kernel<<<grid1d(nrows * wpr * 64, 256, (i64)1 << 20), 256, 0, c->stream>>>(lab, (unsigned long long *)c->m[1], (int)c->nx, nrows, wpr);
    NL_CHECK_LAUNCH();
    // intensities as float32 (score_img[...] = intensity_im[...], mocap_marking.py:595-596)
    if (intensity_host) {
        NL_HIP(hipStreamSynchronize(c->stream));
        const int keep_vmax = c->i_vmax;
        c->i_vmax = -1;                                  // every f[] volume is free to stage raw bytes now
        int rc = upload_convert(c, intensity_host, dtype, mk_intensity(c), c->n, err, errlen);
        c->i_vmax = keep_vmax;
        if (rc) return rc;
    } else {
        if (!c->d_input) return nl_fail(err, errlen, NL_ESTATE, "nl_markers_begin(intensity = NULL) without a resident input");
        const unsigned int g = grid1d(c->n);
        switch (c->input_dtype) {
            case NL_U8: convert_kernel<uint8_t><<<g, 256, 0, c->stream>>>((const uint8_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_I8: convert_kernel<int8_t><<<g, 256, 0, c->stream>>>((const int8_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_U16: convert_kernel<uint16_t><<<g, 256, 0, c->stream>>>((const uint16_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_I16: convert_kernel<int16_t><<<g, 256, 0, c->stream>>>((const int16_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_U32: convert_kernel<uint32_t><<<g, 256, 0, c->stream>>>((const uint32_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_I32: convert_kernel<int32_t><<<g, 256, 0, c->stream>>>((const int32_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_F32: convert_kernel<float><<<g, 256, 0, c->stream>>>((const float *)c->d_input, mk_intensity(c), c->n); break;
            case NL_F64: convert_kernel<double><<<g, 256, 0, c->stream>>>((const double *)c->d_input, mk_intensity(c), c->n); break;
            case NL_U64: convert_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_I64: convert_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)c->d_input, mk_intensity(c), c->n); break;
        }
        NL_CHECK_LAUNCH();
    }
    c->i_labels = -1; c->frangi_ready = 0; c->gauss_ext = nullptr; c->fsq_cache_valid = 0;
    c->mk_state = 1; c->mk_first_scale = 1; c->mk_use = nullptr;
    return NL_OK;
}

// use_im = 'frangi' (mocap_marking.py:675-679): the LoG runs on this float32 image instead of the distance image
extern "C" int nl_markers_use_image(nl_ctx *c, const float *host, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->mk_state < 1) return nl_fail(err, errlen, NL_ESTATE, "nl_markers_use_image before nl_markers_begin");
    if (!host) { c->mk_use = nullptr; return NL_OK; }
    if (vq_alloc_entries(c->nzl, c->ny, c->nx) * 32 < c->n * 8) return nl_fail(err, errlen, NL_ENOMEM, "scratch too small for the LoG source image [out of memory]");
    float *dst = mk_intensity(c) + c->n;
    NL_HIP(hipMemcpyAsync(dst, host, (size_t)c->n * 4, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    c->mk_use = dst;
    return NL_OK;
}

// distance_transform_edt(mask) clamped at `clamp` (= 2 * max_radius_px) and the border shell (mocap_marking.py:419-450)
extern "C" int nl_markers_distance(nl_ctx *c, float clamp, int64_t *n_mask, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->mk_state < 1) return nl_fail(err, errlen, NL_ESTATE, "nl_markers_distance before nl_markers_begin");
    if (!(clamp > 0.0f) || clamp > 1000.0f) return nl_fail(err, errlen, NL_EINVAL, "clamp %g out of range", (double)clamp);
    const int wpr = (int)((c->nx + 63) / 64);
    const VolGeom v = geom(c);
    const int W = (int)clamp;                                   // background farther than this cannot matter
    unsigned long long *mask = (unsigned long long *)c->m[1], *border = (unsigned long long *)c->m[2];
    ProfScope ps(c, "markers_distance");
    const i64 nw = c->nzl * c->ny * wpr;
    mk_border_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, c->stream>>>(mask, border, v, wpr);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemsetAsync(c->f[0], 0, (size_t)c->n * 4, c->stream));            // distance = 0 on the background
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(zero_small(d_cnt, 8, c->stream));
    const unsigned gw_ = grid1d(nw, 256, (i64)1 << 20);               // a wave scans 64 mask words per trip
    mk_edt_x_kernel<<<grid1d(nw, 256, 256 * 32), 256, 0, c->stream>>>(mask, (int *)c->f[1], v, wpr, W, d_cnt);
    mk_edt_axis_kernel<1, 0><<<gw_, 256, 0, c->stream>>>(mask, (const int *)c->f[1], (int *)c->f[2], nullptr, clamp, v, wpr, W);
    mk_edt_axis_kernel<0, 1><<<gw_, 256, 0, c->stream>>>(mask, (const int *)c->f[2], nullptr, c->f[0], clamp, v, wpr, W);
    NL_CHECK_LAUNCH();
    // best response = 0, no peaks yet (mocap_marking.py:483-484)
    NL_HIP(hipMemsetAsync(c->f[3], 0, (size_t)c->n * 4, c->stream));
    NL_HIP(hipMemsetAsync(c->m[0], 0, (size_t)nw * 8 * 2, c->stream));
    NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_mask) *n_mask = (int64_t)(*(unsigned long long *)c->h_small);
    c->i_gauss = 0; c->gauss_ext = nullptr;
    c->mk_state = 2; c->mk_first_scale = 1;
    return NL_OK;
}

// One sigma of mocap_marking.py:488-508: -gaussian_laplace(distance, (s/z_ratio, s, s)) * s^2, clamped at 0, local maxima on
// the valid voxels, best response across scales.  w?2 / w?0: scipy's order-2 / order-0 `_gaussian_kernel1d` (truncate
// 4.0), 2r+1 float64 weights; Y and X share sigma and radius.
extern "C" int nl_markers_log_step(nl_ctx *c, const double *wz2, const double *wz0, int rz, const double *wy2, const double *wy0,
                                   const double *wx2, const double *wx0, int ryx, float s2, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->mk_state < 2) return nl_fail(err, errlen, NL_ESTATE, "nl_markers_log_step before nl_markers_distance");
    const bool flat = !wz2 && !wz0;                    // 2-D image: sigma_vec = (s, s), no Z terms (mocap_marking.py:323-324)
    if (flat && c->nzl != 1) return nl_fail(err, errlen, NL_EINVAL, "Z weights are NULL on a 3-D context");
    if ((!flat && (!wz2 || !wz0)) || !wy2 || !wy0 || !wx2 || !wx0) return nl_fail(err, errlen, NL_EINVAL, "weights are NULL");
    if ((!flat && rz < 1) || ryx < 1 || rz > NL_MAX_RADIUS || ryx > NL_MAX_RADIUS) return nl_fail(err, errlen, NL_EINVAL, "LoG radii (%d, %d) outside the supported range [1, %d]", rz, ryx, NL_MAX_RADIUS);
    // The tiled in-plane kernels hold 2R+1 values per thread (R <= GM_MAX_R) and reflect at most once (R <= ny).  Finer pixels
    // (0.065 um: sigma up to 5.1 px, radius 21) and images thinner than a kernel take the one-thread-per-voxel passes: any radius,
    // scipy's multiple reflection, the same summation order; the volume between the Y and the X pass is allocated on first use.
    const bool generic = ryx > GM_MAX_R || ryx > c->ny;
    if (generic && !c->mk_scratch) NL_HIP(hipMalloc((void **)&c->mk_scratch, (size_t)c->n * 4));
    GaussW gz2, gz0, gy2, gy0, gx2, gx0;
    int rc;
    if ((!flat && ((rc = fill_gw(gz2, wz2, rz, err, errlen)) || (rc = fill_gw(gz0, wz0, rz, err, errlen)))) || (rc = fill_gw(gy2, wy2, ryx, err, errlen)) ||
        (rc = fill_gw(gy0, wy0, ryx, err, errlen)) || (rc = fill_gw(gx2, wx2, ryx, err, errlen)) || (rc = fill_gw(gx0, wx0, ryx, err, errlen))) return rc;
    const VolGeom v = geom(c);
    const i64 z0 = 0, z1 = c->nzl;
    const dim3 blk(256, 1, 1);
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)c->nzl);
    const dim3 g2((unsigned)((c->nx + GYX_COLS - 1) / GYX_COLS), (unsigned)((c->ny + v.chunk - 1) / v.chunk), (unsigned)c->nzl);
    float *dist = c->f[0], *tz = c->f[1], *lap = c->f[2];
    const float *use = c->mk_use ? c->mk_use : dist;            // the image the LoG runs on
    auto ws_of = [](const GaussW &g) { GaussWS w; for (int k = 0; k <= GM_MAX_R; ++k) w.w[k] = k <= g.r ? g.w[k] : 0.0; return w; };
    auto zpass = [&](const GaussW &gz) {
        if (!launch_gauss_fast<0>(c, use, tz, v, z0, z1, gz)) gauss_axis_kernel<0><<<grid, blk, 0, c->stream>>>(use, tz, v, z0, z1, gz);
    };
    // large radii: the fused Y+X kernel turns compute-bound (one output per thread reads 2R+1 LDS values); a marching Y
    // pass plus the stand-alone X kernel (four outputs per thread) through one more scratch volume is faster there
    static int split_from = -1;
    if (split_from < 0) { const char *e = getenv("NELLIE_MK_SPLIT_R"); split_from = e ? atoi(e) : (gyx_tiled() ? 99 : 8); }
    float *tmp2 = mk_intensity(c) + c->n;
    const bool can_split = !c->mk_use && vq_alloc_entries(c->nzl, c->ny, c->nx) * 32 >= c->n * 8;     // mk_use lives in tmp2's place
    const float *yx_src = flat ? use : tz;
    auto yx = [&](const GaussW &gy, const GaussW &gx, bool acc) {
        const GaussWS wy = ws_of(gy), wx = ws_of(gx);
        if (can_split && ryx >= split_from) {
            const dim3 gym((unsigned)((c->nx + 63) / 64), (unsigned)((c->nzl + 3) / 4), (unsigned)((c->ny + v.chunk - 1) / v.chunk));
            const dim3 gxk((unsigned)((c->nx + GX_SEG - 1) / GX_SEG), (unsigned)c->ny, (unsigned)c->nzl);
            const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
            switch (ryx) {
#define NL_MKS(RR) case RR: gauss_march_kernel<1, RR><<<gym, 256, 0, c->stream>>>(yx_src, tmp2, v, z0, z1, wy);                      \
                            gauss_x_kernel<RR><<<gxk, 256, 0, c->stream>>>(tmp2, lap, v, z0, z1, wx, vec4, acc ? 1 : 0); break;
                NL_MKS(1) NL_MKS(2) NL_MKS(3) NL_MKS(4) NL_MKS(5) NL_MKS(6) NL_MKS(7) NL_MKS(8) NL_MKS(9) NL_MKS(10) NL_MKS(11) NL_MKS(12)
#undef NL_MKS
            }
            return;
        }
        if (gyx_tiled()) {
            const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
            switch (ryx) {
#define NL_MKYT(RR) case RR: if (acc) gauss_yx_tile_kernel<RR, true><<<g2.x * g2.y * g2.z, GYX_THREADS, 0, c->stream>>>(yx_src, lap, v, z0, z1, wy, wx, vec4, (int)g2.x, (int)g2.y); \
                             else gauss_yx_tile_kernel<RR, false><<<g2.x * g2.y * g2.z, GYX_THREADS, 0, c->stream>>>(yx_src, lap, v, z0, z1, wy, wx, vec4, (int)g2.x, (int)g2.y); break;
                NL_MKYT(1) NL_MKYT(2) NL_MKYT(3) NL_MKYT(4) NL_MKYT(5) NL_MKYT(6) NL_MKYT(7) NL_MKYT(8) NL_MKYT(9) NL_MKYT(10) NL_MKYT(11) NL_MKYT(12)
#undef NL_MKYT
            }
            return;
        }
        switch (ryx) {
#define NL_MKYX(RR) case RR: if (acc) gauss_yx_kernel<RR, true><<<g2, GYX_THREADS, 0, c->stream>>>(yx_src, lap, v, z0, z1, wy, wx); \
                             else gauss_yx_kernel<RR, false><<<g2, GYX_THREADS, 0, c->stream>>>(yx_src, lap, v, z0, z1, wy, wx); break;
            NL_MKYX(1) NL_MKYX(2) NL_MKYX(3) NL_MKYX(4) NL_MKYX(5) NL_MKYX(6) NL_MKYX(7) NL_MKYX(8) NL_MKYX(9) NL_MKYX(10) NL_MKYX(11) NL_MKYX(12)
#undef NL_MKYX
        }
    };
    // the two in-plane terms in one walk over their common input
    static int dual = -1;
    if (dual < 0) { const char *e = getenv("NELLIE_MK_DUAL"); dual = (e && !atoi(e)) ? 0 : 1; }
    auto yx_dual = [&](bool acc) {
        const GaussWS wya = ws_of(gy2), wxa = ws_of(gx0), wyb = ws_of(gy0), wxb = ws_of(gx2);
        const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
        const unsigned nb = g2.x * g2.y * g2.z;
        switch (ryx) {
#define NL_MKD(RR) case RR: if (acc) gauss_yx_dual_kernel<RR, true><<<nb, GYX_THREADS, 0, c->stream>>>(yx_src, lap, v, z0, z1, wya, wxa, wyb, wxb, vec4, (int)g2.x, (int)g2.y); \
                            else gauss_yx_dual_kernel<RR, false><<<nb, GYX_THREADS, 0, c->stream>>>(yx_src, lap, v, z0, z1, wya, wxa, wyb, wxb, vec4, (int)g2.x, (int)g2.y); break;
            NL_MKD(1) NL_MKD(2) NL_MKD(3) NL_MKD(4) NL_MKD(5) NL_MKD(6) NL_MKD(7) NL_MKD(8) NL_MKD(9) NL_MKD(10) NL_MKD(11) NL_MKD(12)
#undef NL_MKD
        }
    };
    const bool use_dual = dual && gyx_tiled();
    auto yx_generic = [&](const GaussW &gy, const GaussW &gx, bool acc) {
        gauss_axis_kernel<1><<<grid, blk, 0, c->stream>>>(yx_src, c->mk_scratch, v, z0, z1, gy);
        if (acc) gauss_axis_kernel<2, true><<<grid, blk, 0, c->stream>>>(c->mk_scratch, lap, v, z0, z1, gx);
        else gauss_axis_kernel<2><<<grid, blk, 0, c->stream>>>(c->mk_scratch, lap, v, z0, z1, gx);
    };
    {
        ProfScope ps(c, "markers_log");
        // generic_laplace: output = d2/dz2 term; output += d2/dy2 term; output += d2/dx2 term (float32 adds, in this order)
        if (generic) {
            if (flat) { yx_generic(gy2, gx0, false); yx_generic(gy0, gx2, true); }
            else { zpass(gz2); yx_generic(gy0, gx0, false); zpass(gz0); yx_generic(gy2, gx0, true); yx_generic(gy0, gx2, true); }
        } else if (flat) {
            if (use_dual) yx_dual(false);
            else { yx(gy2, gx0, false); yx(gy0, gx2, true); }
        } else {
            zpass(gz2); yx(gy0, gx0, false);
            zpass(gz0);
            if (use_dual) yx_dual(true);
            else { yx(gy2, gx0, true); yx(gy0, gx2, true); }
        }
        NL_CHECK_LAUNCH();
    }
    {
        ProfScope ps(c, "markers_peaks");
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 nw = c->nzl * c->ny * wpr;
        mk_peak_kernel<<<grid1d(nw, 256, (i64)1 << 20), 256, 0, c->stream>>>(lap, s2, (const unsigned long long *)c->m[1], dist, c->f[3],
                                                                                 (unsigned long long *)c->m[0], v, wpr);
        NL_CHECK_LAUNCH();
    }
    return NL_OK;
}

// mocap_marking.py:569-606 + 692-695: intensity-based non-maximum suppression of the peaks; *n_markers = markers kept.
extern "C" int nl_markers_finish(nl_ctx *c, int peak_min_distance, int64_t *n_markers, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->mk_state < 2) return nl_fail(err, errlen, NL_ESTATE, "nl_markers_finish before nl_markers_distance");
    if (peak_min_distance < 0 || peak_min_distance > 31) return nl_fail(err, errlen, NL_EINVAL, "peak_min_distance %d out of range", peak_min_distance);
    const int wpr = (int)((c->nx + 63) / 64);
    const i64 nw = c->nzl * c->ny * wpr;
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(zero_small(d_cnt, 8, c->stream));
    {
        ProfScope ps(c, "markers_nms");
        mk_nms_kernel<<<grid1d(nw, 256, 256 * 32), 256, 0, c->stream>>>((const unsigned long long *)c->m[0], mk_intensity(c), peak_min_distance,
                                                                             (unsigned long long *)c->m[0] + nw, geom(c), wpr, d_cnt);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_markers) *n_markers = (int64_t)(*(unsigned long long *)c->h_small);
    c->mk_state = 3;
    return NL_OK;
}

// D2H of the three products (any of the pointers may be NULL): marker uint8, distance float32, border uint8
extern "C" int nl_markers_store(nl_ctx *c, uint8_t *marker, float *distance, uint8_t *border, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->mk_state < 2 || (marker && c->mk_state < 3)) return nl_fail(err, errlen, NL_ESTATE, "nl_markers_store before the products exist");
    const int wpr = (int)((c->nx + 63) / 64);
    const i64 nrows = c->nzl * c->ny, nw = nrows * wpr;
    uint8_t *stage = (uint8_t *)c->f[1];
    if (distance) NL_HIP(hipMemcpyAsync(distance, c->f[0], (size_t)c->n * 4, hipMemcpyDeviceToHost, c->stream));
    if (border) {
        mk_bits_to_u8_kernel<<<grid1d(nw * 64, 256, (i64)1 << 20), 256, 0, c->stream>>>((const unsigned long long *)c->m[2], stage, (int)c->nx, nrows, wpr);
        NL_CHECK_LAUNCH();
        NL_HIP(hipMemcpyAsync(border, stage, (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    }
    if (marker) {
        uint8_t *stage2 = stage + c->n;
        mk_bits_to_u8_kernel<<<grid1d(nw * 64, 256, (i64)1 << 20), 256, 0, c->stream>>>((const unsigned long long *)c->m[0] + nw, stage2, (int)c->nx, nrows, wpr);
        NL_CHECK_LAUNCH();
        NL_HIP(hipMemcpyAsync(marker, stage2, (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    }
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// ---------------------------------------------------------------------------------- Label -------
extern "C" int nl_label_load_frangi(nl_ctx *c, const float *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    const i64 plane = c->ny * c->nx;
    c->i_vmax = 3; c->i_gauss = 0; c->i_labels = -1;
    NL_HIP(hipMemcpyAsync(c->f[c->i_vmax] + z0 * plane, host, (size_t)(z1 - z0) * plane * 4, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    c->frangi_ready = 1;
    return NL_OK;
}

// planes [z0, z1) of the context's frame against `host_original` = those planes of the original image (a Z slab masks the planes it owns)
extern "C" int nl_label_intensity_mask_planes(nl_ctx *c, const void *host_original, int dtype, double thresh, int64_t z0, int64_t z1,
                                              char *err, size_t errlen) {
    NL_ENTER(c);
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    const i64 count = (z1 - z0) * c->ny * c->nx;
    const size_t es = dtype_size(dtype);
    if (!es || !host_original) return nl_fail(err, errlen, NL_EINVAL, "bad original image (dtype code %d)", dtype);
    void *raw = nullptr;
    NL_HIP(hipMalloc(&raw, (size_t)count * es));
    hipError_t e = hipMemcpyAsync(raw, host_original, (size_t)count * es, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { hipFree(raw); return nl_fail(err, errlen, NL_EHIP, "upload of the original image failed: %s", hipGetErrorString(e)); }
    float *fr = c->f[c->i_vmax] + z0 * c->ny * c->nx;
    const unsigned int g = grid1d(count);
    switch (dtype) {
        case NL_U8: intensity_mask_kernel<uint8_t><<<g, 256, 0, c->stream>>>((const uint8_t *)raw, fr, thresh, count); break;
        case NL_I8: intensity_mask_kernel<int8_t><<<g, 256, 0, c->stream>>>((const int8_t *)raw, fr, thresh, count); break;
        case NL_U16: intensity_mask_kernel<uint16_t><<<g, 256, 0, c->stream>>>((const uint16_t *)raw, fr, thresh, count); break;
        case NL_I16: intensity_mask_kernel<int16_t><<<g, 256, 0, c->stream>>>((const int16_t *)raw, fr, thresh, count); break;
        case NL_U32: intensity_mask_kernel<uint32_t><<<g, 256, 0, c->stream>>>((const uint32_t *)raw, fr, thresh, count); break;
        case NL_I32: intensity_mask_kernel<int32_t><<<g, 256, 0, c->stream>>>((const int32_t *)raw, fr, thresh, count); break;
        case NL_F32: intensity_mask_kernel<float><<<g, 256, 0, c->stream>>>((const float *)raw, fr, thresh, count); break;
        case NL_F64: intensity_mask_kernel<double><<<g, 256, 0, c->stream>>>((const double *)raw, fr, thresh, count); break;
        case NL_U64: intensity_mask_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)raw, fr, thresh, count); break;
        case NL_I64: intensity_mask_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)raw, fr, thresh, count); break;
    }
    e = hipGetLastError();
    hipStreamSynchronize(c->stream);
    hipFree(raw);
    if (e != hipSuccess) return nl_fail(err, errlen, NL_EHIP, "intensity mask kernel: %s", hipGetErrorString(e));
    return NL_OK;
}
extern "C" int nl_label_intensity_mask(nl_ctx *c, const void *host_original, int dtype, double thresh, char *err, size_t errlen) {
    if (!c) return nl_fail(err, errlen, NL_EINVAL, "ctx is NULL");
    return nl_label_intensity_mask_planes(c, host_original, dtype, thresh, 0, c->nzl, err, errlen);
}

extern "C" int nl_flat_sample_gather(nl_ctx *c, int field, int64_t offset, int64_t step, float *out, int64_t cap, int64_t *n,
                                     char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_SUPPORT(c);
    if (step < 1 || offset < 0) return nl_fail(err, errlen, NL_EINVAL, "bad offset/step");
    if (field != NL_FIELD_FRANGI && field != NL_FIELD_GAUSS) return nl_fail(err, errlen, NL_EINVAL, "flat sampling supports GAUSS/FRANGI");
    // flat index runs over the GLOBAL volume; this rank contributes indices inside its owned planes
    const i64 plane = c->ny * c->nx;
    const i64 g_begin = (c->gz0 + c->own_lo) * plane, g_end = (c->gz0 + c->own_hi) * plane;
    i64 k0 = 0;
    if (g_begin > offset) k0 = (g_begin - offset + step - 1) / step;
    i64 k1 = (g_end > offset) ? (g_end - offset + step - 1) / step : 0;    // k in [k0,k1)
    const i64 count = k1 > k0 ? k1 - k0 : 0;
    if (n) *n = count;
    if (count == 0 || (!out && cap == 0)) return NL_OK;   // size query
    if (!out || cap < count) return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld < %lld samples", (i64)cap, count);
    const float *src = (field == NL_FIELD_FRANGI) ? c->f[c->i_vmax] : gauss_cur(c);
    float *stage = nullptr;
    for (int k = 0; k < 3; ++k) if (k != c->i_gauss && c->f[k] != src) { stage = c->f[k]; break; }
    {
        ProfScope ps(c, "sample");
        // local flat index = global - gz0*plane
        flat_gather_kernel<<<(unsigned)((count + 255) / 256), 256, 0, c->stream>>>(src, -c->gz0 * plane, offset + k0 * step, step, count, stage);
        NL_CHECK_LAUNCH();
    }
    NL_HIP(hipMemcpyAsync(out, stage, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// nl_flat_sample_gather restricted to the positive samples, compacted on the device (labelling.py:426-433 takes
// values[values > 0]); order unspecified.  cap >= the count nl_flat_sample_gather reports.
extern "C" int nl_flat_sample_gather_positive(nl_ctx *c, int field, int64_t offset, int64_t step, float *out, int64_t cap,
                                              int64_t *n, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_SUPPORT(c);
    if (step < 1 || offset < 0) return nl_fail(err, errlen, NL_EINVAL, "bad offset/step");
    if (field != NL_FIELD_FRANGI && field != NL_FIELD_GAUSS) return nl_fail(err, errlen, NL_EINVAL, "flat sampling supports GAUSS/FRANGI");
    const i64 plane = c->ny * c->nx;
    const i64 g_begin = (c->gz0 + c->own_lo) * plane, g_end = (c->gz0 + c->own_hi) * plane;
    i64 k0 = 0;
    if (g_begin > offset) k0 = (g_begin - offset + step - 1) / step;
    i64 k1 = (g_end > offset) ? (g_end - offset + step - 1) / step : 0;
    const i64 count = k1 > k0 ? k1 - k0 : 0;
    if (n) *n = 0;
    if (count == 0) return NL_OK;
    if (!out || cap < count) return nl_fail(err, errlen, NL_EINVAL, "output capacity %lld < %lld samples", (i64)cap, count);
    const float *src = (field == NL_FIELD_FRANGI) ? c->f[c->i_vmax] : gauss_cur(c);
    float *stage = nullptr;
    for (int k = 0; k < 3; ++k) if (k != c->i_gauss && c->f[k] != src) { stage = c->f[k]; break; }
    unsigned int *d_n = (unsigned int *)c->d_small;
    NL_HIP(zero_small(d_n, 4, c->stream));
    {
        ProfScope ps(c, "sample");
        flat_gather_pos_kernel<<<(unsigned)((count + 255) / 256), 256, 0, c->stream>>>(src, -c->gz0 * plane, offset + k0 * step, step, count, stage, d_n);
        NL_CHECK_LAUNCH();
    }
    return fetch_counted(c, stage, d_n, count, out, cap, n, err, errlen);
}

template <int FG, int CONN>
static int run_ccl(nl_ctx *c, const uint8_t *mask, int *L, char *err, size_t errlen) {
    const i64 nrows = c->nzl * c->ny;
    const i64 waves = nrows * ((c->nx + 63) / 64);
    ccl_init_kernel<FG><<<(unsigned)((waves * 64 + 255) / 256), 256, 0, c->stream>>>(mask, L, c->nx, nrows);
    NL_CHECK_LAUNCH();
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)c->nzl);
    ccl_merge_kernel<FG, CONN><<<grid, 256, 0, c->stream>>>(mask, L, c->nzl, c->ny, c->nx);
    NL_CHECK_LAUNCH();
    ccl_flatten_kernel<<<grid1d(c->n), 256, 0, c->stream>>>(L, c->n);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

// Voxel-level variant (first implementation): kept as the fallback for rows longer than 65535 voxels or
// pathological masks with more than N/2 runs, and as an A/B reference (NELLIE_LABEL_VOXEL=1).
static int label_run_voxels(nl_ctx *c, int has_thr, float thr, int64_t min_area, int fill_holes, int64_t *n_labels,
                            char *err, size_t errlen) {
    // buffers: frangi = f[i_vmax]; the other three float volumes serve as int32 scratch
    int free_idx[3], nf = 0;
    for (int k = 0; k < 4; ++k) if (k != c->i_vmax) free_idx[nf++] = k;
    int *L = (int *)c->f[free_idx[0]];
    int *aux = (int *)c->f[free_idx[1]];
    int *out = (int *)c->f[free_idx[2]];
    uint8_t *mA = c->m[1], *mB = c->m[2], *flag = c->m[0];
    const i64 n = c->n;
    const i64 nrows = c->nzl * c->ny;
    const i64 waves = nrows * ((c->nx + 63) / 64);
    int rc;
    ProfScope ps(c, "label");
    threshold_kernel<<<grid1d(n), 256, 0, c->stream>>>(c->f[c->i_vmax], mA, has_thr, thr, n);
    NL_CHECK_LAUNCH();
    if (fill_holes) {
        if ((rc = run_ccl<0, 6>(c, mA, L, err, errlen))) return rc;
        clear_root_flags_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, flag, n);
        NL_CHECK_LAUNCH();
        border_mark_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, flag, geom(c));
        NL_CHECK_LAUNCH();
        fill_holes_apply_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, flag, mA, n);
        NL_CHECK_LAUNCH();
    }
    if ((rc = run_ccl<1, 26>(c, mA, L, err, errlen))) return rc;
    zero_at_roots_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, aux, n);
    NL_CHECK_LAUNCH();
    area_count_kernel<<<grid1d(waves * 64, 256, 256 * 16), 256, 0, c->stream>>>(L, aux, c->nx, nrows);
    NL_CHECK_LAUNCH();
    const int ma = (int)(min_area > 0x7fffffff ? 0x7fffffff : min_area);
    keep_large_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, aux, mB, ma, n);
    NL_CHECK_LAUNCH();
    {
        const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)c->nzl);
        majority_kernel<<<grid, 256, 0, c->stream>>>(mB, mA, geom(c));
        NL_CHECK_LAUNCH();
    }
    if ((rc = run_ccl<1, 26>(c, mA, L, err, errlen))) return rc;
    const i64 nblk = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    unsigned int *blk = (unsigned int *)c->d_blk;
    unsigned long long *d_total = (unsigned long long *)c->d_small;
    root_count_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(L, n, nullptr, n, blk);
    NL_CHECK_LAUNCH();
    blk_scan_kernel<<<1, 1024, 0, c->stream>>>(blk, nblk, d_total);
    NL_CHECK_LAUNCH();
    root_assign_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(L, n, nullptr, n, blk, aux);
    NL_CHECK_LAUNCH();
    relabel_kernel<<<grid1d(n), 256, 0, c->stream>>>(L, aux, out, n);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(c->h_small, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_labels) *n_labels = (int64_t)(*(unsigned long long *)c->h_small);
    c->i_labels = free_idx[2];
    return NL_OK;
}


// exclusive scan of n u32 values (in -> out); returns nothing, total = out[n-1] + in[n-1]
static int scan_excl_u32(nl_ctx *c, const unsigned int *in, unsigned int *out, i64 n, char *err, size_t errlen) {
    const i64 nblk = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    unsigned int *blk = (unsigned int *)c->d_blk;
    unsigned long long *d_total = (unsigned long long *)c->d_small + 16;
    chunk_sum_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(in, n, blk);
    NL_CHECK_LAUNCH();
    blk_scan_kernel<<<1, 1024, 0, c->stream>>>(blk, nblk, d_total);
    NL_CHECK_LAUNCH();
    chunk_scan_kernel<<<(unsigned)nblk, 256, 0, c->stream>>>(in, out, n, blk);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

struct RunSet { RunRec *runs; int *parent; unsigned int *row_off; i64 nruns; int *proot = nullptr; int *link = nullptr; bool proot_valid = false;
                // nruns < 0: the host did not wait for the count (it lies at row_off[nrows]); kernels then read it there
                bool host_count = true; i64 cap = 0;
                RunN rn() const { return RunN{host_count ? nullptr : n_ptr, nruns, cap}; }
                const unsigned int *n_ptr = nullptr; };
// workgroups of a kernel that walks the runs: exact when the host knows the count, else a grid the kernels stride over
static unsigned run_blocks(const RunSet &rs, i64 nrows, int per_block = 256) {
    if (rs.host_count) return (unsigned)((rs.nruns + per_block - 1) / per_block > 0 ? (rs.nruns + per_block - 1) / per_block : 1);
    i64 b = nrows / 64;                                   // ~4 runs per row at one thread per run: more only means grid-stride trips
    if (b < 256) b = 256;
    if (b > 8192) b = 8192;
    return (unsigned)b;
}

// Geometry the run-level Label works on: the whole (global) volume as rows of bit-packed words.
struct LabelGeo {
    i64 nz, ny, nx;            // volume the masks describe (the global one for a Z-slab run)
    i64 nrows; int wpr; i64 nwords;
    unsigned int *rows;        // 2 x (nrows + 2) u32: run counts, run offsets
    unsigned long long *bitsA, *bitsB;
    i64 paint_row0, paint_row1;   // rows this context paints ...
    int *paint_out;               // ... into this int32 buffer (row paint_row0 first)
    int *link_scratch = nullptr;  // >= one int per possible run, free until the paint (enables the two-level union-find)
    int zf_lo = 0, zf_hi = -2;    // planes of this run set that are true Z faces of the volume (-1: none; set by label_geo_faces)
    i64 gz0 = 0, gnz = 0;         // placement of plane 0 of the run set in the global volume (boundary rules)
};
static void label_geo_whole(LabelGeo &g) { g.zf_lo = 0; g.zf_hi = (int)g.nz - 1; g.gz0 = 0; g.gnz = g.nz; }

// runs of `bits` (or of its complement) + union-find over them, flattened
template <int CONN>
static int build_components(nl_ctx *c, const LabelGeo &g, const unsigned long long *bits, int invert, RunSet &rs, i64 cap,
                            bool *overflow, char *err, size_t errlen) {
    unsigned int *counts = g.rows, *row_off = g.rows + (g.nrows + 2);
    NL_HIP(zero_small(counts + g.nrows, 4, c->stream));
    if (g.wpr <= 64) {
        int P = 1;
        while (P < g.wpr) P <<= 1;
        const i64 groups = (g.nrows + 64 / P - 1) / (64 / P);
        rl_count_wave_kernel<<<grid1d(groups * 64, 256, 16384), 256, 0, c->stream>>>(bits, invert, counts, g.nrows, g.wpr, (int)g.nx, P);
    }
    else rl_count_kernel<<<(unsigned)((g.nrows + 255) / 256), 256, 0, c->stream>>>(bits, invert, counts, g.nrows, g.wpr, (int)g.nx);
    NL_CHECK_LAUNCH();
    int rc = scan_excl_u32(c, counts, row_off, g.nrows + 1, err, errlen);
    if (rc) return rc;
    rs.row_off = row_off; rs.n_ptr = row_off + g.nrows; rs.cap = cap;
    rs.proot_valid = false;
    *overflow = false;
    unsigned int *d_ovf = (unsigned int *)c->d_small + 60;           // sticky within a labelling: zeroed by the caller, read at its end
    if (rs.host_count) {
        NL_HIP(hipMemcpyAsync(c->h_small, row_off + g.nrows, 4, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
        rs.nruns = (i64)(*(unsigned int *)c->h_small);
        *overflow = rs.nruns > cap;
        if (*overflow || rs.nruns == 0) return NL_OK;
    } else {
        rs.nruns = -1;                                                 // kernels read row_off[nrows]; beyond `cap` they are no-ops
    }
    // two levels (see label_runs.inc): planes in LDS, then component pairs across planes; NELLIE_UF_PLANES=0: one level
    static int two_level = -1;
    if (two_level < 0) { const char *e = getenv("NELLIE_UF_PLANES"); two_level = (e && !atoi(e)) ? 0 : 1; }
    // segments per plane: at least ~1024 workgroups for the in-LDS level (a 136-plane slab would otherwise use half of the CUs)
    int seg_shift = 5;
    {
        static int seg_target = -1;                                   // NELLIE_UF_SEG_WGS=1: one workgroup per plane (round 3)
        if (seg_target < 0) { const char *e = getenv("NELLIE_UF_SEG_WGS"); seg_target = (e && atoi(e) > 0) ? atoi(e) : 1024; }
        const i64 want = (seg_target + g.nz - 1) / g.nz;
        while (((g.ny + ((i64)1 << seg_shift) - 1) >> seg_shift) > want) ++seg_shift;
        while (g.nz * ((g.ny + ((i64)1 << seg_shift) - 1) >> seg_shift) > 8192 && ((i64)1 << seg_shift) < g.ny) ++seg_shift;
    }
    const int nseg = (int)((g.ny + ((i64)1 << seg_shift) - 1) >> seg_shift);
    const bool lvl2 = two_level && g.nz * nseg <= 8192 && rs.proot && rs.link;
    int *link = lvl2 ? rs.link : nullptr;                            // rl_emit_kernel fills the pair filter's slots with -1
    if (g.wpr <= 30)
        rl_emit_kernel<true><<<(unsigned)((g.nrows + 255) / 256), 256, (size_t)256 * (g.wpr + 1) * 8, c->stream>>>(bits, invert, row_off, rs.runs, rs.parent, g.nrows, g.wpr, (int)g.nx, link, cap, d_ovf);
    else
        rl_emit_kernel<false><<<(unsigned)((g.nrows + 255) / 256), 256, 0, c->stream>>>(bits, invert, row_off, rs.runs, rs.parent, g.nrows, g.wpr, (int)g.nx, link, cap, d_ovf);
    NL_CHECK_LAUNCH();
    const unsigned gr = run_blocks(rs, g.nrows);
    const RunN rn = rs.rn();
    if (lvl2) {
        uint8_t *seg_done = (uint8_t *)c->d_small + (52 << 10);
        rl_union_plane_kernel<CONN><<<(unsigned)(g.nz * nseg), 256, 0, c->stream>>>(rs.runs, row_off, rs.parent, rs.proot, (int)g.ny,
                                                                         CONN == 6 ? 1 : 0, g.zf_lo, g.zf_hi, (int)g.nx, seg_done, seg_shift, nseg, rn);
        rl_union_cross_kernel<CONN><<<gr, 256, 0, c->stream>>>(rs.runs, row_off, rs.parent, rs.proot, rs.link, rn, (int)g.ny,
                                                               CONN == 6 ? 1 : 0, g.zf_lo, g.zf_hi, (int)g.nx, seg_done, seg_shift, nseg);
        rs.proot_valid = true;
    } else {
        rl_union_kernel<CONN><<<gr, 256, 0, c->stream>>>(rs.runs, row_off, rs.parent, rn, (int)g.ny,
                                                         CONN == 6 ? 1 : 0, g.zf_lo, g.zf_hi, (int)g.nx);
    }
    NL_CHECK_LAUNCH();
    ccl_flatten_kernel<<<gr, 256, 0, c->stream>>>(rs.parent, rs.nruns, rn.p, cap);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

// ids 1..K in raster order of each component's first voxel (scipy.ndimage.label numbering), painted as int32
static int number_and_paint(nl_ctx *c, const LabelGeo &g, const RunSet &rs, int *aux, int64_t *n_labels, char *err, size_t errlen,
                            bool *overflow = nullptr) {
    unsigned long long total = 0;
    unsigned int *blk = (unsigned int *)c->d_blk;
    unsigned long long *d_total = (unsigned long long *)c->d_small;
    const bool any = !rs.host_count || rs.nruns > 0;
    if (any) {
        const RunN rn = rs.rn();
        const unsigned nb = rs.host_count ? (unsigned)((rs.nruns + SCAN_CHUNK - 1) / SCAN_CHUNK) : run_blocks(rs, g.nrows / 16 + 1);
        root_count_kernel<<<nb, 256, 0, c->stream>>>(rs.parent, rs.nruns, rn.p, rs.cap, blk);
        NL_CHECK_LAUNCH();
        blk_scan_kernel<<<1, 1024, 0, c->stream>>>(blk, rs.host_count ? (rs.nruns + SCAN_CHUNK - 1) / SCAN_CHUNK : 0, d_total, rn.p, rs.cap);
        NL_CHECK_LAUNCH();
        root_assign_kernel<<<nb, 256, 0, c->stream>>>(rs.parent, rs.nruns, rn.p, rs.cap, blk, aux);
        NL_CHECK_LAUNCH();
        NL_HIP(hipMemcpyAsync(c->h_small, d_total, 8, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipMemcpyAsync((char *)c->h_small + 8, (unsigned int *)c->d_small + 60, 4, hipMemcpyDeviceToHost, c->stream));   // the overflow flag
    }
    rl_paint_kernel<<<grid1d((g.paint_row1 - g.paint_row0) * 64, 256, (i64)1 << 22), 256, 0, c->stream>>>(
        g.bitsA, rs.row_off, rs.parent, aux, g.paint_out, g.paint_row0, g.paint_row1, g.wpr, (int)g.nx);
    NL_CHECK_LAUNCH();
    NL_HIP(hipStreamSynchronize(c->stream));
    if (any) total = *(unsigned long long *)c->h_small;
    if (overflow) *overflow = any && !rs.host_count && *(unsigned int *)((char *)c->h_small + 8) != 0;
    if (n_labels) *n_labels = (int64_t)total;
    return NL_OK;
}

// labelling.py:484-509 on a bit-packed mask (bitsA holds `frame > thr` on entry).  *overflow: more runs than scratch.
static int label_core(nl_ctx *c, const LabelGeo &g, int64_t min_area, int fill_holes, int64_t *n_labels, bool *overflow,
                      char *err, size_t errlen) {
    int free_idx[3], nf = 0;
    for (int k = 0; k < 4; ++k) if (k != c->i_vmax) free_idx[nf++] = k;
    const i64 cap = c->n / 2;                                 // runs that fit the scratch volumes
    RunSet rs;
    rs.runs = (RunRec *)c->f[free_idx[0]];                    // 8 B x cap  = 4N bytes
    rs.parent = (int *)c->f[free_idx[1]];                     // 4 B x cap  = 2N bytes
    int *aux = rs.parent + cap;                               // 4 B x cap  = 2N bytes (areas, then new ids)
    rs.proot = aux;                                           // in-plane roots during the unions (aux is idle until the areas)
    rs.link = g.link_scratch;
    uint8_t *flag = c->m[0];
    const VolGeom vg{g.nz, g.ny, g.nx, g.gz0, g.gnz};        // boundary rules: true faces of the global volume only
    int rc;
    *overflow = false;
    // No launch below waits for a run count (round 4): the kernels read it from device memory and stride over the runs; a run
    // set beyond the scratch volumes makes them no-ops and raises a flag that the single wait at the end returns.
    static int dev_count = -1;
    if (dev_count < 0) { const char *e = getenv("NELLIE_LABEL_HOST_COUNTS"); dev_count = (e && atoi(e)) ? 0 : 1; }
    rs.host_count = !dev_count;
    NL_HIP(zero_small((unsigned int *)c->d_small + 60, 4, c->stream));
    auto known_empty = [&]() { return rs.host_count && rs.nruns == 0; };
    if (fill_holes) {
        // binary_fill_holes: 6-connected background components that reach no face become foreground
        if ((rc = build_components<6>(c, g, g.bitsA, 1, rs, cap, overflow, err, errlen))) return rc;
        if (*overflow) return NL_OK;
        if (!known_empty()) {
            const unsigned gr = run_blocks(rs, g.nrows);
            rl_fill_u8_kernel<<<gr, 256, 0, c->stream>>>(flag, rs.rn());
            rl_border_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, flag, rs.rn(), vg);
            NL_CHECK_LAUNCH();
            rl_fill_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, flag, g.bitsA, rs.rn(), g.wpr);
            NL_CHECK_LAUNCH();
        }
    }
    // first labelling + small-object removal
    if ((rc = build_components<26>(c, g, g.bitsA, 0, rs, cap, overflow, err, errlen))) return rc;
    if (*overflow) return NL_OK;
    NL_HIP(hipMemsetAsync(g.bitsB, 0, (size_t)g.nwords * 8, c->stream));
    if (!known_empty()) {
        const unsigned gr = run_blocks(rs, g.nrows);
        rl_fill_u32_kernel<<<gr, 256, 0, c->stream>>>((unsigned int *)aux, 0u, rs.rn());
        rl_area_kernel<<<run_blocks(rs, g.nrows / 16 + 1, RL_CHUNK), 256, 0, c->stream>>>(rs.runs, rs.parent, aux, rs.rn());
        NL_CHECK_LAUNCH();
        const int ma = (int)(min_area > 0x7fffffff ? 0x7fffffff : min_area);
        rl_keep_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, aux, ma, g.bitsB, rs.rn(), g.wpr);
        NL_CHECK_LAUNCH();
    }
    // majority smoothing, second labelling
    majority_bits_kernel<<<(unsigned)((g.nwords + 255) / 256), 256, 0, c->stream>>>(g.bitsB, g.bitsA, vg, g.wpr, 0, g.nz);
    NL_CHECK_LAUNCH();
    if ((rc = build_components<26>(c, g, g.bitsA, 0, rs, cap, overflow, err, errlen))) return rc;
    if (*overflow) return NL_OK;
    if ((rc = number_and_paint(c, g, rs, aux, n_labels, err, errlen, overflow))) return rc;
    if (*overflow) return NL_OK;
    c->i_labels = free_idx[2];
    return NL_OK;
}

static int label_out_index(const nl_ctx *c) {
    int last = -1;
    for (int k = 0; k < 4; ++k) if (k != c->i_vmax) last = k;
    return last;
}

extern "C" int nl_label_run(nl_ctx *c, int has_thr, float thr, int64_t min_area, int fill_holes, int64_t *n_labels,
                            char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->frangi_ready) return nl_fail(err, errlen, NL_ESTATE, "nl_label_run before a Frangi volume exists");
    if (c->nzl != c->gnz) return nl_fail(err, errlen, NL_EINVAL, "nl_label_run works on a whole volume (Z-slabs: nl_label_pack / nl_label_run_global)");
    static int force_voxel = -1;
    if (force_voxel < 0) { const char *e = getenv("NELLIE_LABEL_VOXEL"); force_voxel = (e && atoi(e)) ? 1 : 0; }
    if (force_voxel || c->nx > 65535) return label_run_voxels(c, has_thr, thr, min_area, fill_holes, n_labels, err, errlen);
    LabelGeo g;
    g.nz = c->nzl; g.ny = c->ny; g.nx = c->nx;
    g.nrows = c->nzl * c->ny; g.wpr = (int)((c->nx + 63) / 64); g.nwords = g.nrows * g.wpr;
    g.rows = c->d_rows;
    g.bitsA = (unsigned long long *)c->m[1]; g.bitsB = (unsigned long long *)c->m[2];
    g.paint_row0 = 0; g.paint_row1 = g.nrows; g.paint_out = (int *)c->f[label_out_index(c)];
    g.link_scratch = g.paint_out;                 // the whole label volume (4N bytes) is idle until the paint
    label_geo_whole(g);
    ProfScope ps(c, "label");
    const unsigned long long *support = (c->support_epoch + 1 == c->epoch.load() && c->d_support && has_thr && thr >= 0.0f) ? c->d_support : nullptr;
    c->last_label_sparse = support ? 1 : 0;
    // dense: a pure streaming read wants one wave per row; sparse: most rows end after one load, several rows per wave
    rl_threshold_pack_kernel<<<grid1d(g.nrows * 64, 256, support ? (i64)256 * 64 : (i64)1 << 22), 256, 0, c->stream>>>(c->f[c->i_vmax], support, g.bitsA, has_thr, thr,
                                                                                          (int)c->nx, g.nrows, g.wpr);
    NL_CHECK_LAUNCH();
    bool overflow = false;
    int rc = label_core(c, g, min_area, fill_holes, n_labels, &overflow, err, errlen);
    if (rc) return rc;
    if (overflow) return label_run_voxels(c, has_thr, thr, min_area, fill_holes, n_labels, err, errlen);
    return NL_OK;
}

// ---- Z-slab Label: every rank packs the mask bits of its own planes into a GLOBAL bit mask (1 bit/voxel,
// gnz*ny*nx/8 bytes), the bit planes are all-gathered, and the run-level labelling (cheap: it scales with the
// number of runs, not voxels) runs redundantly on the global mask on every rank, which then paints only its own
// planes.  Exact by construction: it IS the single-volume algorithm.
static int ensure_global_label_buffers(nl_ctx *c, char *err, size_t errlen) {
    const i64 grows = c->gnz * c->ny;
    const int wpr = (int)((c->nx + 63) / 64);
    if (!c->gbits[0]) {
        NL_HIP(hipMalloc((void **)&c->gbits[0], (size_t)grows * wpr * 8));
        NL_HIP(hipMalloc((void **)&c->gbits[1], (size_t)grows * wpr * 8));
        NL_HIP(hipMalloc((void **)&c->grows, ((size_t)grows + 2) * 2 * 4));
        if ((grows + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK + 1 > c->blk_cap) {
            hipFree(c->d_blk);
            c->blk_cap = (grows + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
            NL_HIP(hipMalloc(&c->d_blk, (size_t)c->blk_cap * 4));
        }
    }
    return NL_OK;
}

extern "C" int nl_label_pack(nl_ctx *c, int has_thr, float thr, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->frangi_ready) return nl_fail(err, errlen, NL_ESTATE, "nl_label_pack before a Frangi volume exists");
    if (c->nx > 65535) return nl_fail(err, errlen, NL_EINVAL, "rows longer than 65535 voxels are not supported on Z-slabs");
    int rc = ensure_global_label_buffers(c, err, errlen);
    if (rc) return rc;
    const int wpr = (int)((c->nx + 63) / 64);
    const i64 own_rows = (c->own_hi - c->own_lo) * c->ny;
    const i64 row0 = (c->gz0 + c->own_lo) * c->ny;
    ProfScope ps(c, "label");
    rl_threshold_pack_kernel<<<grid1d(own_rows * 64, 256, (i64)1 << 22), 256, 0, c->stream>>>(
        c->f[c->i_vmax] + c->own_lo * c->ny * c->nx, nullptr, c->gbits[0] + row0 * wpr, has_thr, thr, (int)c->nx, own_rows, wpr);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

// host access to bit-mask rows [row0, row0+nrows) of the global mask (tests / communicators without RCCL)
extern "C" int nl_label_bits_get(nl_ctx *c, int64_t row0, int64_t nrows, uint64_t *host, char *err, size_t errlen) {
    NL_ENTER(c);
    const int wpr = (int)((c->nx + 63) / 64);
    if (!c->gbits[0] || !host || row0 < 0 || nrows < 1 || row0 + nrows > c->gnz * c->ny) return nl_fail(err, errlen, NL_EINVAL, "bad bit-mask row range");
    NL_HIP(hipMemcpyAsync(host, c->gbits[0] + row0 * wpr, (size_t)nrows * wpr * 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}
extern "C" int nl_label_bits_put(nl_ctx *c, int64_t row0, int64_t nrows, const uint64_t *host, char *err, size_t errlen) {
    NL_ENTER(c);
    const int wpr = (int)((c->nx + 63) / 64);
    if (!c->gbits[0] || !host || row0 < 0 || nrows < 1 || row0 + nrows > c->gnz * c->ny) return nl_fail(err, errlen, NL_EINVAL, "bad bit-mask row range");
    NL_HIP(hipMemcpyAsync(c->gbits[0] + row0 * wpr, host, (size_t)nrows * wpr * 8, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// all-gather of the mask bit planes over RCCL: rank r broadcasts the rows of its own planes (slab_plane0[r] ..)
extern "C" int nl_label_bits_allgather(nl_ctx *c, const int64_t *slab_plane0, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_label_bits_allgather before nl_comm_init");
    if (!c->gbits[0] || !slab_plane0) return nl_fail(err, errlen, NL_ESTATE, "nl_label_bits_allgather before nl_label_pack");
    const int wpr = (int)((c->nx + 63) / 64);
    ProfScope ps(c, "halo");
    NL_NCCL(rccl().GroupStart());
    for (int r = 0; r < c->world; ++r) {
        const i64 p0 = slab_plane0[r], p1 = slab_plane0[r + 1];        // world + 1 entries, last = gnz
        unsigned long long *ptr = c->gbits[0] + p0 * c->ny * wpr;
        NL_NCCL(rccl().Broadcast(ptr, ptr, (size_t)((p1 - p0) * c->ny * wpr), ncclUint64, r, (ncclComm_t)c->comm, c->stream));
    }
    NL_NCCL(rccl().GroupEnd());
    return NL_OK;
}

extern "C" int nl_label_run_global(nl_ctx *c, int64_t min_area, int fill_holes, int64_t *n_labels, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->gbits[0]) return nl_fail(err, errlen, NL_ESTATE, "nl_label_run_global before nl_label_pack");
    LabelGeo g;
    g.nz = c->gnz; g.ny = c->ny; g.nx = c->nx;
    g.nrows = c->gnz * c->ny; g.wpr = (int)((c->nx + 63) / 64); g.nwords = g.nrows * g.wpr;
    g.rows = c->grows;
    g.bitsA = c->gbits[0]; g.bitsB = c->gbits[1];
    g.paint_row0 = (c->gz0 + c->own_lo) * c->ny; g.paint_row1 = (c->gz0 + c->own_hi) * c->ny;
    g.paint_out = (int *)c->f[label_out_index(c)] + c->own_lo * c->ny * c->nx;
    label_geo_whole(g);
    ProfScope ps(c, "label");
    bool overflow = false;
    int rc = label_core(c, g, min_area, fill_holes, n_labels, &overflow, err, errlen);
    if (rc) return rc;
    if (overflow) return nl_fail(err, errlen, NL_ENOMEM, "the global mask has more runs than this slab's scratch volumes hold [out of memory]");
    return NL_OK;
}

// ---- Z-slab Label without replication (see label_runs.inc "Z-slab Label" and nellie_amd/sharded.py) -----------------
struct SlabGeo { LabelGeo g; RunSet rs; int *aux; unsigned int *sel, *scan; int *stage; i64 cap; int row_lo, row_hi; bool has_lo, has_hi; int out_idx; };
static int slab_geo(nl_ctx *c, SlabGeo &sg, char *err, size_t errlen) {
    if (c->nx > 65535) return nl_fail(err, errlen, NL_EINVAL, "rows longer than 65535 voxels are not supported on Z-slabs");
    sg.has_lo = c->gz0 + c->own_lo > 0; sg.has_hi = c->gz0 + c->own_hi < c->gnz;
    if ((sg.has_lo && c->own_lo < 1) || (sg.has_hi && c->own_hi > c->nzl - 1))
        return nl_fail(err, errlen, NL_EINVAL, "the slab holds no ghost plane next to an interior interface");
    c->sl_e0 = c->own_lo - (sg.has_lo ? 1 : 0); c->sl_e1 = c->own_hi + (sg.has_hi ? 1 : 0);
    const int wpr = (int)((c->nx + 63) / 64);
    LabelGeo &g = sg.g;
    g.nz = c->sl_e1 - c->sl_e0; g.ny = c->ny; g.nx = c->nx;
    g.nrows = g.nz * c->ny; g.wpr = wpr; g.nwords = g.nrows * wpr;
    g.rows = c->d_rows;
    g.bitsA = (unsigned long long *)c->m[1] + c->sl_e0 * c->ny * wpr;
    g.bitsB = (unsigned long long *)c->m[2] + c->sl_e0 * c->ny * wpr;
    g.gz0 = c->gz0 + c->sl_e0; g.gnz = c->gnz;
    g.zf_lo = (g.gz0 == 0) ? 0 : -1; g.zf_hi = (g.gz0 + g.nz == c->gnz) ? (int)g.nz - 1 : -1;
    int free_idx[3], nf = 0;
    for (int k = 0; k < 4; ++k) if (k != c->i_vmax) free_idx[nf++] = k;
    sg.cap = c->n / 2;
    sg.rs.runs = (RunRec *)c->f[free_idx[0]];
    sg.rs.parent = (int *)c->f[free_idx[1]];
    sg.aux = sg.rs.parent + sg.cap;
    sg.rs.proot = sg.aux;
    sg.rs.link = (int *)c->f[free_idx[2]];
    sg.rs.row_off = g.rows + (g.nrows + 2);
    sg.rs.nruns = c->sl_nruns; sg.rs.cap = sg.cap;
    sg.sel = (unsigned int *)c->f[free_idx[2]]; sg.scan = sg.sel + sg.cap;
    sg.stage = (int *)c->f[free_idx[2]];
    sg.out_idx = free_idx[2];
    sg.row_lo = (int)((c->own_lo - c->sl_e0) * c->ny); sg.row_hi = (int)((c->own_hi - c->sl_e0) * c->ny);
    g.paint_row0 = sg.row_lo; g.paint_row1 = sg.row_hi;
    g.paint_out = (int *)c->f[free_idx[2]] + c->own_lo * c->ny * c->nx;
    g.link_scratch = sg.rs.link;
    return NL_OK;
}

/* mask bits of the owned planes: frangi > thr (labelling.py:478) */
extern "C" int nl_slab_label_pack(nl_ctx *c, int has_thr, float thr, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->frangi_ready) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_label_pack before a Frangi volume exists");
    SlabGeo sg; int rc = slab_geo(c, sg, err, errlen); if (rc) return rc;
    const int wpr = sg.g.wpr;
    const i64 own_rows = (c->own_hi - c->own_lo) * c->ny;
    ProfScope ps(c, "label");
    // right after the fused epilogue the frame is known to be <= 0 outside the opened mask: read it only there
    const unsigned long long *support = (c->support_epoch + 1 == c->epoch.load() && c->d_support && has_thr && thr >= 0.0f)
                                            ? c->d_support + c->own_lo * c->ny * wpr : nullptr;
    c->last_label_sparse = support ? 1 : 0;
    rl_threshold_pack_kernel<<<grid1d(own_rows * 64, 256, support ? (i64)256 * 64 : (i64)1 << 22), 256, 0, c->stream>>>(
        c->f[c->i_vmax] + c->own_lo * c->ny * c->nx, support, (unsigned long long *)c->m[1] + c->own_lo * c->ny * wpr, has_thr, thr, (int)c->nx, own_rows, wpr);
    NL_CHECK_LAUNCH();
    c->sl_phase = -1; c->sl_nruns = 0; c->sl_numbered = 0;
    return NL_OK;
}

/* one bit plane (local plane index) of mask `which` (0: the working mask, 1: the kept-objects mask) to / from the host */
extern "C" int nl_slab_bits_get(nl_ctx *c, int which, int64_t plane, uint64_t *host, char *err, size_t errlen) {
    NL_ENTER(c);
    const int wpr = (int)((c->nx + 63) / 64);
    if (!host || plane < 0 || plane >= c->nzl || which < 0 || which > 1) return nl_fail(err, errlen, NL_EINVAL, "bad bit-plane request");
    const unsigned long long *b = (const unsigned long long *)c->m[1 + which] + plane * c->ny * wpr;
    NL_HIP(hipMemcpyAsync(host, b, (size_t)c->ny * wpr * 8, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}
extern "C" int nl_slab_bits_put(nl_ctx *c, int which, int64_t plane, const uint64_t *host, char *err, size_t errlen) {
    NL_ENTER(c);
    const int wpr = (int)((c->nx + 63) / 64);
    if (!host || plane < 0 || plane >= c->nzl || which < 0 || which > 1) return nl_fail(err, errlen, NL_EINVAL, "bad bit-plane request");
    unsigned long long *b = (unsigned long long *)c->m[1 + which] + plane * c->ny * wpr;
    NL_HIP(hipMemcpyAsync(b, host, (size_t)c->ny * wpr * 8, hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}
/* the same exchange with the Z neighbours over RCCL: my first / last owned bit plane -> their ghost plane, theirs -> mine */
extern "C" int nl_slab_bits_exchange(nl_ctx *c, int which, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_bits_exchange before nl_comm_init");
    if (which < 0 || which > 1) return nl_fail(err, errlen, NL_EINVAL, "bad mask selector");
    const int wpr = (int)((c->nx + 63) / 64);
    const size_t words = (size_t)c->ny * wpr;
    unsigned long long *b = (unsigned long long *)c->m[1 + which];
    const bool has_lo = c->rank > 0, has_hi = c->rank + 1 < c->world;
    if ((has_lo && c->own_lo < 1) || (has_hi && c->own_hi > c->nzl - 1)) return nl_fail(err, errlen, NL_EINVAL, "no ghost plane to receive into");
    ncclComm_t comm = (ncclComm_t)c->comm;
    ProfScope ps(c, "halo");
    NL_NCCL(rccl().GroupStart());
    if (has_lo) {
        NL_NCCL(rccl().Send(b + c->own_lo * words, words, ncclUint64, c->rank - 1, comm, c->stream));
        NL_NCCL(rccl().Recv(b + (c->own_lo - 1) * words, words, ncclUint64, c->rank - 1, comm, c->stream));
    }
    if (has_hi) {
        NL_NCCL(rccl().Send(b + (c->own_hi - 1) * words, words, ncclUint64, c->rank + 1, comm, c->stream));
        NL_NCCL(rccl().Recv(b + c->own_hi * words, words, ncclUint64, c->rank + 1, comm, c->stream));
    }
    NL_NCCL(rccl().GroupEnd());
    return NL_OK;
}

/* Page-locked staging for the lists the host hands back (patches, selections): they are copied here first, so the H2D copy
   can stay asynchronous -- the area is rewritten only by a later call, and every phase waits for the stream in between. */
static int slab_host_stage(nl_ctx *c, size_t ints, char *err, size_t errlen) {
    if (ints > c->h_sl_ints) {
        NL_HIP(hipStreamSynchronize(c->stream));
        if (c->h_sl) hipHostFree(c->h_sl);
        c->h_sl = nullptr; c->h_sl_ints = 0;
        const size_t cap = ints + ints / 2 + 4096;
        NL_HIP(hipHostMalloc((void **)&c->h_sl, cap * 4, hipHostMallocDefault));
        c->h_sl_ints = cap;
    }
    return NL_OK;
}

/* One phase of the slab protocol up to the tables, in ONE call with one wait of its own (plus the run count inside
   build_components): components of the owned planes + ghost planes (SL_FILL: 6-connected background, SL_AREA / SL_NUMBER:
   26-connected foreground of the working mask), the phase's per-tree quantity, and the COMPACT tables of the four planes the
   neighbours also see (label_runs.inc "the tables the ranks exchange").  gather != 0: the blobs of all ranks, all-gathered over
   RCCL on the context stream in fixed blocks of block_ints int32 (no size negotiation, no host round trip in between);
   out receives world (gather) or 1 blocks.  *need_ints = the largest blob of any rank: if it exceeds block_ints the caller calls
   again with phase = -1 and a larger block (the device tables are still there; nothing is recomputed). */
extern "C" int nl_slab_phase(nl_ctx *c, int phase, int gather, int64_t block_ints, int32_t *out, int64_t *need_ints, int64_t *nruns,
                             char *err, size_t errlen) {
    NL_ENTER(c);
    if (phase < -1 || phase > SL_NUMBER || !out || !need_ints || block_ints < 8) return nl_fail(err, errlen, NL_EINVAL, "bad phase / buffers");
    if (gather && !c->comm) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_phase(gather) before nl_comm_init");
    if (phase < 0 && c->sl_phase < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_phase(-1) before a phase ran");
    SlabGeo sg; int rc = slab_geo(c, sg, err, errlen); if (rc) return rc;
    LabelGeo &g = sg.g; RunSet &rs = sg.rs;
    const int W = gather ? c->world : 1;
    // entries a plane can hold: at most one per run, a row has at most (nx + 1) / 2 runs
    const i64 max_runs_plane = c->ny * ((c->nx + 1) / 2);
    const int capE = (int)(max_runs_plane < ((i64)1 << 18) ? max_runs_plane : ((i64)1 << 18));
    const size_t blob_cap = 8 + (size_t)8 * capE;
    // [header 16 | entry indices 4 capE | this rank's blob (at least one block: the all-gather sends a whole block) | gathered blocks]
    const size_t o_bidx = 16, o_blob = o_bidx + (size_t)4 * capE;
    const size_t o_gath = o_blob + (blob_cap > (size_t)block_ints ? blob_cap : (size_t)block_ints);
    const size_t need_dev = o_gath + (size_t)W * (size_t)block_ints;
    if (need_dev > c->d_sl_ints || capE != c->sl_capE) {
        if (phase < 0 && capE != c->sl_capE) return nl_fail(err, errlen, NL_ESTATE, "slab tables of another geometry");
        int *nb = nullptr;
        NL_HIP(hipMalloc((void **)&nb, (need_dev + need_dev / 4) * 4));
        if (c->d_sl) {
            const size_t keep = o_blob + blob_cap < c->d_sl_ints ? o_blob + blob_cap : c->d_sl_ints;
            if (phase < 0) NL_HIP(hipMemcpyAsync(nb, c->d_sl, keep * 4, hipMemcpyDeviceToDevice, c->stream));
            NL_HIP(hipStreamSynchronize(c->stream));
            hipFree(c->d_sl);
        }
        c->d_sl = nb; c->d_sl_ints = need_dev + need_dev / 4; c->sl_capE = capE;
    }
    if ((rc = slab_host_stage(c, (size_t)W * (size_t)block_ints, err, errlen))) return rc;
    int *hdr = c->d_sl, *bidx = c->d_sl + o_bidx, *blob = c->d_sl + o_blob, *gath = c->d_sl + o_gath;
    if (phase >= 0) {
        ProfScope ps(c, "label");
        bool overflow = false;
        if (phase == SL_FILL) rc = build_components<6>(c, g, g.bitsA, 1, rs, sg.cap, &overflow, err, errlen);
        else rc = build_components<26>(c, g, g.bitsA, 0, rs, sg.cap, &overflow, err, errlen);
        if (rc) return rc;
        if (overflow) return nl_fail(err, errlen, NL_ENOMEM, "the slab's mask has more runs than its scratch volumes hold [out of memory]");
        c->sl_nruns = rs.nruns; c->sl_phase = phase; c->sl_numbered = 0;
        // the four planes the neighbours also see; their segment components BEFORE the per-tree quantity takes proot's memory
        const i64 ny = c->ny;
        SlabPlanes pl;
        // (a plane only matters towards a side that has a neighbour: the ghost plane and the owned plane next to it)
        pl.row[0] = sg.has_lo ? 0 : -1;
        pl.row[1] = sg.has_lo ? (int)((c->own_lo - c->sl_e0) * ny) : -1;
        pl.row[2] = sg.has_hi ? (int)((c->own_hi - 1 - c->sl_e0) * ny) : -1;
        pl.row[3] = sg.has_hi ? (int)((c->own_hi - c->sl_e0) * ny) : -1;
        sl_boundary_kernel<<<4, 1024, 0, c->stream>>>(rs.row_off, (rs.nruns && rs.proot_valid) ? rs.proot : nullptr, pl, (int)ny, capE, bidx, hdr);
        NL_CHECK_LAUNCH();
        if (rs.nruns) {
            const unsigned gr = (unsigned)((rs.nruns + 255) / 256);
            if (phase == SL_FILL) {
                NL_HIP(hipMemsetAsync(c->m[0], 0, (size_t)rs.nruns, c->stream));
                rl_border_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, c->m[0], rs.rn(), VolGeom{g.nz, g.ny, g.nx, g.gz0, g.gnz});
            } else if (phase == SL_AREA) {
                NL_HIP(hipMemsetAsync(sg.aux, 0, (size_t)rs.nruns * 4, c->stream));
                sl_area_kernel<<<(unsigned)((rs.nruns + RL_CHUNK - 1) / RL_CHUNK), 256, 0, c->stream>>>(rs.runs, rs.parent, sg.aux, rs.nruns, sg.row_lo, sg.row_hi);
            } else {
                NL_HIP(hipMemsetD32Async((hipDeviceptr_t)sg.aux, 0x7fffffff, (size_t)rs.nruns, c->stream));
                sl_first_own_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, sg.aux, rs.nruns, sg.row_lo, sg.row_hi, rs.row_off + sg.row_lo);
            }
            NL_CHECK_LAUNCH();
        }
        sl_table2_kernel<<<4, 256, 0, c->stream>>>(rs.parent, sg.aux, phase == SL_FILL ? c->m[0] : nullptr, bidx, capE, hdr, blob, rs.row_off + g.nrows);
        NL_CHECK_LAUNCH();
    }
    const size_t send = (size_t)block_ints < blob_cap ? (size_t)block_ints : blob_cap;
    if (gather && c->world > 1) {
        ProfScope ps(c, "halo");
        NL_NCCL(rccl().AllGather(blob, gath, (size_t)block_ints, ncclInt32, (ncclComm_t)c->comm, c->stream));
        NL_HIP(hipMemcpyAsync(c->h_sl, gath, (size_t)W * (size_t)block_ints * 4, hipMemcpyDeviceToHost, c->stream));
    } else {
        NL_HIP(hipMemcpyAsync(c->h_sl, blob, send * 4, hipMemcpyDeviceToHost, c->stream));
    }
    NL_HIP(hipStreamSynchronize(c->stream));
    i64 need = 0;
    for (int r = 0; r < W; ++r) {
        const int *b = c->h_sl + (size_t)r * (size_t)block_ints;
        if (b[5]) return nl_fail(err, errlen, NL_ENOMEM, "rank %d: a boundary plane holds more than %d components [out of memory]", gather ? r : c->rank, capE);
        if (b[4] > need) need = b[4];
    }
    *need_ints = need;
    if (nruns) *nruns = c->sl_nruns;
    if (need <= block_ints) memcpy(out, c->h_sl, (size_t)W * (size_t)block_ints * 4);
    return NL_OK;
}

/* quantity[roots[i]] = values[i]: what the host learned about trees that continue on other ranks */
extern "C" int nl_slab_patch(nl_ctx *c, int64_t n, const int32_t *roots, const int32_t *values, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->sl_phase < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_patch before nl_slab_phase");
    if (n == 0) return NL_OK;
    if (n < 0 || !roots || !values || 2 * n > c->n) return nl_fail(err, errlen, NL_EINVAL, "bad patch arguments");
    SlabGeo sg; int rc = slab_geo(c, sg, err, errlen); if (rc) return rc;
    if ((rc = slab_host_stage(c, (size_t)2 * n, err, errlen))) return rc;
    memcpy(c->h_sl, roots, (size_t)n * 4); memcpy(c->h_sl + n, values, (size_t)n * 4);
    int *d_idx = sg.stage, *d_val = sg.stage + n;
    NL_HIP(hipMemcpyAsync(d_idx, c->h_sl, (size_t)2 * n * 4, hipMemcpyHostToDevice, c->stream));
    sl_patch_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(sg.aux, c->sl_phase == SL_FILL ? c->m[0] : nullptr, d_idx, d_val, (int)n);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

/* SL_FILL: enclosed background of the owned planes -> working mask (labelling.py:486); SL_AREA: runs of objects with
   >= min_area voxels -> kept-objects mask of the owned planes (labelling.py:495-501) */
extern "C" int nl_slab_apply(nl_ctx *c, int64_t min_area, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->sl_phase != SL_FILL && c->sl_phase != SL_AREA) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_apply outside the fill / area phases");
    SlabGeo sg; int rc = slab_geo(c, sg, err, errlen); if (rc) return rc;
    LabelGeo &g = sg.g; RunSet &rs = sg.rs;
    ProfScope ps(c, "label");
    const unsigned gr = (unsigned)((rs.nruns + 255) / 256);
    if (c->sl_phase == SL_FILL) {
        if (rs.nruns) sl_fill_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, c->m[0], g.bitsA, rs.nruns, g.wpr, sg.row_lo, sg.row_hi);
    } else {
        NL_HIP(hipMemsetAsync(g.bitsB + (i64)sg.row_lo * g.wpr, 0, (size_t)(sg.row_hi - sg.row_lo) * g.wpr * 8, c->stream));
        const int ma = (int)(min_area > 0x7fffffff ? 0x7fffffff : min_area);
        if (rs.nruns) sl_keep_kernel<<<gr, 256, 0, c->stream>>>(rs.runs, rs.parent, sg.aux, ma, g.bitsB, rs.nruns, g.wpr, sg.row_lo, sg.row_hi);
    }
    NL_CHECK_LAUNCH();
    return NL_OK;
}

/* working mask (owned planes) = majority filter of the kept-objects mask, ghost planes included (labelling.py:503-505) */
extern "C" int nl_slab_majority(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    SlabGeo sg; int rc = slab_geo(c, sg, err, errlen); if (rc) return rc;
    LabelGeo &g = sg.g;
    ProfScope ps(c, "label");
    const i64 z_lo = c->own_lo - c->sl_e0, z_hi = c->own_hi - c->sl_e0;
    majority_bits_kernel<<<(unsigned)(((z_hi - z_lo) * g.ny * g.wpr + 255) / 256), 256, 0, c->stream>>>(g.bitsB, g.bitsA, VolGeom{g.nz, g.ny, g.nx, g.gz0, g.gnz}, g.wpr, z_lo, z_hi);
    NL_CHECK_LAUNCH();
    return NL_OK;
}

/* SL_NUMBER: ranks the trees this rank numbers, in raster order of their first run: the roots on the owned planes, minus
   `clear` (trees that continue on other ranks), plus `set` (those of them this rank owns).  *n_local = their count;
   ids_of_set[i] = 1-based local rank of set[i] (what the other ranks need to know about the trees this rank owns). */
extern "C" int nl_slab_number(nl_ctx *c, int64_t n_clear, const int32_t *clear, int64_t n_set, const int32_t *set, int64_t *n_local,
                              int32_t *ids_of_set, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->sl_phase != SL_NUMBER) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_number outside the numbering phase");
    if (n_clear < 0 || n_set < 0 || (n_clear && !clear) || (n_set && (!set || !ids_of_set)) || n_clear + n_set > c->n / 4) return nl_fail(err, errlen, NL_EINVAL, "bad selection lists");
    SlabGeo sg; int rc = slab_geo(c, sg, err, errlen); if (rc) return rc;
    RunSet &rs = sg.rs;
    ProfScope ps(c, "label");
    unsigned long long total = 0;
    if (rs.nruns) {
        sl_select_kernel<<<(unsigned)((rs.nruns + 255) / 256), 256, 0, c->stream>>>(rs.runs, rs.parent, sg.sel, rs.nruns, sg.row_lo, sg.row_hi);
        int *d_idx = (int *)(sg.scan + rs.nruns);                    // behind the scan array (cap >= nruns + the lists: checked below)
        if (rs.nruns + n_clear + 2 * n_set > sg.cap) return nl_fail(err, errlen, NL_ENOMEM, "selection lists do not fit the scratch volume [out of memory]");
        if ((rc = slab_host_stage(c, (size_t)(n_clear + n_set) + 2 + (size_t)n_set, err, errlen))) return rc;
        if (n_clear) memcpy(c->h_sl, clear, (size_t)n_clear * 4);
        if (n_set) memcpy(c->h_sl + n_clear, set, (size_t)n_set * 4);
        if (n_clear + n_set) NL_HIP(hipMemcpyAsync(d_idx, c->h_sl, (size_t)(n_clear + n_set) * 4, hipMemcpyHostToDevice, c->stream));
        if (n_clear) sl_set_u32_kernel<<<(unsigned)((n_clear + 255) / 256), 256, 0, c->stream>>>(sg.sel, d_idx, (int)n_clear, 0u);
        if (n_set) sl_set_u32_kernel<<<(unsigned)((n_set + 255) / 256), 256, 0, c->stream>>>(sg.sel, d_idx + n_clear, (int)n_set, 1u);
        NL_CHECK_LAUNCH();
        if ((rc = scan_excl_u32(c, sg.sel, sg.scan, rs.nruns, err, errlen))) return rc;
        int *h_back = c->h_sl + n_clear + n_set;                     // [scan of the last run, its flag, ids of `set`]
        int *d_out = d_idx + n_clear + n_set;
        if (n_set) {
            sl_gather_kernel<<<(unsigned)((n_set + 255) / 256), 256, 0, c->stream>>>(sg.scan, d_idx + n_clear, d_out, (int)n_set, 1);
            NL_CHECK_LAUNCH();
            NL_HIP(hipMemcpyAsync(h_back + 2, d_out, (size_t)n_set * 4, hipMemcpyDeviceToHost, c->stream));
        }
        NL_HIP(hipMemcpyAsync(h_back, sg.scan + rs.nruns - 1, 4, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipMemcpyAsync(h_back + 1, sg.sel + rs.nruns - 1, 4, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
        total = (unsigned long long)(unsigned int)h_back[0] + (unsigned int)h_back[1];
        if (n_set) memcpy(ids_of_set, h_back + 2, (size_t)n_set * 4);
    } else if (n_set) {
        return nl_fail(err, errlen, NL_EINVAL, "selection on an empty run set");
    }
    if (n_local) *n_local = (int64_t)total;
    c->sl_numbered = 1;
    return NL_OK;
}

/* int32 labels of the owned planes (labelling.py:507): a selected tree gets base + its local rank, the trees listed in
   `roots` (they continue on other ranks) get `labels` */
extern "C" int nl_slab_paint(nl_ctx *c, int64_t base, int64_t n, const int32_t *roots, const int32_t *labels, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->sl_numbered) return nl_fail(err, errlen, NL_ESTATE, "nl_slab_paint before nl_slab_number");
    if (n < 0 || (n && (!roots || !labels))) return nl_fail(err, errlen, NL_EINVAL, "bad label patch");
    SlabGeo sg; int rc = slab_geo(c, sg, err, errlen); if (rc) return rc;
    LabelGeo &g = sg.g; RunSet &rs = sg.rs;
    ProfScope ps(c, "label");
    if (rs.nruns) {
        sl_ids_kernel<<<(unsigned)((rs.nruns + 255) / 256), 256, 0, c->stream>>>(sg.sel, sg.scan, (int)base, sg.aux, rs.nruns);
        if (n) {
            if (rs.nruns + 2 * n > sg.cap) return nl_fail(err, errlen, NL_ENOMEM, "label patch does not fit the scratch volume [out of memory]");
            int *d_idx = (int *)(sg.scan + rs.nruns), *d_val = d_idx + n;
            if ((rc = slab_host_stage(c, (size_t)2 * n, err, errlen))) return rc;
            memcpy(c->h_sl, roots, (size_t)n * 4); memcpy(c->h_sl + n, labels, (size_t)n * 4);
            NL_HIP(hipMemcpyAsync(d_idx, c->h_sl, (size_t)2 * n * 4, hipMemcpyHostToDevice, c->stream));
            sl_patch_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(sg.aux, nullptr, d_idx, d_val, (int)n);   // stream order: before the paint
        }
        NL_CHECK_LAUNCH();
    }
    rl_paint_kernel<<<grid1d((g.paint_row1 - g.paint_row0) * 64, 256, (i64)1 << 22), 256, 0, c->stream>>>(
        g.bitsA, rs.row_off, rs.parent, sg.aux, g.paint_out, g.paint_row0, g.paint_row1, g.wpr, (int)g.nx);
    NL_CHECK_LAUNCH();
    NL_HIP(hipStreamSynchronize(c->stream));
    c->i_labels = sg.out_idx;
    c->sl_numbered = 0; c->sl_phase = -1;
    return NL_OK;
}

// Host side of the slab protocol: the graph of (rank, tree) nodes joined through the shared planes (see include/nellie_amd.h).
// Plain C++ on a few thousand entries; numpy needed ~1 ms per rank for the same (sharded.join_slab_tables, kept as the model).
extern "C" int nl_host_slab_join(int world, const int32_t *blobs, int64_t block_ints, int64_t cap, int64_t *n_nodes, int64_t *n_comp,
                                 int64_t *node_rank, int32_t *node_root, int64_t *node_val, int64_t *node_comp, char *err, size_t errlen) {
    if (world < 1 || !blobs || block_ints < 8 || !n_nodes || !n_comp) return nl_fail(err, errlen, NL_EINVAL, "bad join arguments");
    struct Tab { const int32_t *root[4], *val[4]; int n[4]; };
    std::vector<Tab> tabs((size_t)world);
    std::vector<std::vector<int32_t>> uniq((size_t)world);          // a rank's trees, ascending root
    std::vector<i64> base((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) {
        const int32_t *b = blobs + (size_t)r * (size_t)block_ints;
        i64 total = 0;
        for (int k = 0; k < 4; ++k) { if (b[k] < 0) return nl_fail(err, errlen, NL_EINVAL, "negative table size"); total += b[k]; }
        if (8 + 2 * total > block_ints) return nl_fail(err, errlen, NL_EINVAL, "rank %d: table of %lld entries exceeds the block", r, (long long)total);
        i64 off = 8;
        for (int k = 0; k < 4; ++k) { tabs[r].n[k] = b[k]; tabs[r].root[k] = b + off; tabs[r].val[k] = b + total + off; off += b[k]; }
        auto &u = uniq[r];
        u.assign(b + 8, b + 8 + total);
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        base[r + 1] = base[r] + (i64)u.size();
    }
    const i64 n = base[world];
    *n_nodes = n;
    if (n > cap) { *n_comp = 0; return NL_OK; }                     // the caller sizes its arrays from *n_nodes and calls again
    auto node_of = [&](int r, int32_t root) -> i64 {
        const auto &u = uniq[r];
        return base[r] + (i64)(std::lower_bound(u.begin(), u.end(), root) - u.begin());
    };
    std::vector<i64> par((size_t)n);
    for (i64 i = 0; i < n; ++i) par[i] = i;
    auto find = [&](i64 i) -> i64 { while (par[i] != i) { par[i] = par[par[i]]; i = par[i]; } return i; };
    std::vector<char> have((size_t)n, 0);
    for (int r = 0; r < world; ++r)
        for (int k = 0; k < 4; ++k)
            for (int e = 0; e < tabs[r].n[k]; ++e) {
                const i64 v = node_of(r, tabs[r].root[k][e]);
                if (!have[v]) { have[v] = 1; node_rank[v] = r; node_root[v] = tabs[r].root[k][e]; node_val[v] = tabs[r].val[k][e]; }
            }
    for (int r = 0; r + 1 < world; ++r)
        for (int pair = 0; pair < 2; ++pair) {
            const int mine = 2 + pair, theirs = pair;
            if (tabs[r].n[mine] != tabs[r + 1].n[theirs])
                return nl_fail(err, errlen, NL_EINVAL, "slab tables of ranks %d and %d disagree (%d vs %d entries): the ghost bit planes are stale",
                               r, r + 1, tabs[r].n[mine], tabs[r + 1].n[theirs]);
            for (int e = 0; e < tabs[r].n[mine]; ++e) {
                i64 a = find(node_of(r, tabs[r].root[mine][e])), b = find(node_of(r + 1, tabs[r + 1].root[theirs][e]));
                if (a != b) { if (a < b) par[b] = a; else par[a] = b; }          // the smaller node stays the root
            }
        }
    i64 nc = 0;
    std::vector<i64> id((size_t)n, -1);
    for (i64 i = 0; i < n; ++i) { const i64 rt = find(i); if (id[rt] < 0) id[rt] = nc++; node_comp[i] = id[rt]; }   // rt <= i: numbered by smallest node
    *n_comp = nc;
    return NL_OK;
}

extern "C" int nl_label_store(nl_ctx *c, int32_t *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->i_labels < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_label_store before nl_label_run");
    return store_planes(c, c->f[c->i_labels], host, 4, z0, z1, err, errlen);
}

// ------------------------------------------------------------------------------ frame streaming ---
// 3-D+T stacks (BASELINE config 5): frame t+1 travels host -> HBM on a copy stream while frame t computes, and the
// outputs of frame t-1 travel back on a second copy stream.  Host buffers must be pinned (nl_pinned_alloc) for the
// copies to be asynchronous.
// ------------------------------------------------------------------ Network (pixel class, branch labels) -------
// networking.py:672-683: skeleton voxels classified by their 3x3x3 (2-D: 3x3) occupancy.
extern "C" int nl_skel_pixel_class(nl_ctx *c, const int32_t *skel_host, uint8_t *pixel_class_host, int64_t *n_skel,
                                   char *err, size_t errlen) {
    NL_ENTER(c);
    NL_JOIN_SIDE(c);
    if (!skel_host) return nl_fail(err, errlen, NL_EINVAL, "nl_skel_pixel_class: skel is NULL");
    if (c->own_lo != 0 || c->own_hi != c->nzl || c->gnz != c->nzl) return nl_fail(err, errlen, NL_EINVAL, "the Network kernels run on a whole volume");
    const int wpr = (int)((c->nx + 63) / 64);
    const i64 nrows = c->nzl * c->ny, nw = nrows * wpr;
    ProfScope ps(c, "network");
    NL_HIP(hipMemcpyAsync(c->f[3], skel_host, (size_t)c->n * 4, hipMemcpyHostToDevice, c->stream));
    unsigned long long *skel = (unsigned long long *)c->m[1], *branch = (unsigned long long *)c->m[2];
    mk_pack_labels_kernel<<<grid1d(nw * 64, 256, (i64)1 << 20), 256, 0, c->stream>>>((const int *)c->f[3], skel, (int)c->nx, nrows, wpr);
    NL_CHECK_LAUNCH();
    uint8_t *pc = (uint8_t *)c->f[2];
    NL_HIP(hipMemsetAsync(pc, 0, (size_t)c->n, c->stream));
    unsigned long long *d_cnt = (unsigned long long *)c->d_small;
    NL_HIP(zero_small(d_cnt, 8, c->stream));
    nw_pixel_class_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, c->stream>>>(skel, pc, branch, geom(c), wpr, d_cnt);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(c->h_small, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    if (pixel_class_host) NL_HIP(hipMemcpyAsync(pixel_class_host, pc, (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    if (n_skel) *n_skel = (int64_t)(*(unsigned long long *)c->h_small);
    c->i_labels = -1; c->frangi_ready = 0; c->gauss_ext = nullptr; c->fsq_cache_valid = 0; c->mk_state = 0;
    c->nw_state = 1;
    return NL_OK;
}

// networking.py:758-800: label(pixel_class > 0 & pixel_class != 4, structure = ones(3,3,3)) -> int32 ids in raster order.
// pixel_class_host = NULL uses the classes nl_skel_pixel_class left on the device.
extern "C" int nl_skel_branch_labels(nl_ctx *c, const uint8_t *pixel_class_host, int32_t *labels_host, int64_t *n_labels,
                                     char *err, size_t errlen) {
    NL_ENTER(c);
    NL_JOIN_SIDE(c);
    if (c->own_lo != 0 || c->own_hi != c->nzl || c->gnz != c->nzl) return nl_fail(err, errlen, NL_EINVAL, "the Network kernels run on a whole volume");
    if (c->nx > 65535) return nl_fail(err, errlen, NL_EINVAL, "rows longer than 65535 voxels are not supported here");
    LabelGeo g;
    g.nz = c->nzl; g.ny = c->ny; g.nx = c->nx;
    g.nrows = c->nzl * c->ny; g.wpr = (int)((c->nx + 63) / 64); g.nwords = g.nrows * g.wpr;
    g.rows = c->d_rows;
    g.bitsA = (unsigned long long *)c->m[2]; g.bitsB = (unsigned long long *)c->m[1];
    g.paint_row0 = 0; g.paint_row1 = g.nrows; g.paint_out = (int *)c->f[3];
    label_geo_whole(g);
    ProfScope ps(c, "network");
    if (pixel_class_host) {
        uint8_t *pc = (uint8_t *)c->f[2];
        NL_HIP(hipMemcpyAsync(pc, pixel_class_host, (size_t)c->n, hipMemcpyHostToDevice, c->stream));
        nw_pack_branch_kernel<<<grid1d(g.nwords * 64, 256, (i64)1 << 20), 256, 0, c->stream>>>(pc, g.bitsA, (int)c->nx, g.nrows, g.wpr);
        NL_CHECK_LAUNCH();
        c->i_labels = -1; c->frangi_ready = 0; c->gauss_ext = nullptr; c->fsq_cache_valid = 0; c->mk_state = 0;
    } else if (c->nw_state < 1) {
        return nl_fail(err, errlen, NL_ESTATE, "nl_skel_branch_labels(pixel_class = NULL) before nl_skel_pixel_class");
    }
    const i64 cap = c->n / 2;
    RunSet rs;
    rs.runs = (RunRec *)c->f[0];
    rs.parent = (int *)c->f[1];
    int *aux = rs.parent + cap;
    bool overflow = false;
    int rc = build_components<26>(c, g, g.bitsA, 0, rs, cap, &overflow, err, errlen);
    if (rc) return rc;
    if (overflow) return nl_fail(err, errlen, NL_ENOMEM, "more branch runs than scratch [out of memory]");
    if ((rc = number_and_paint(c, g, rs, aux, n_labels, err, errlen))) return rc;
    if (labels_host) {
        NL_HIP(hipMemcpyAsync(labels_host, g.paint_out, (size_t)c->n * 4, hipMemcpyDeviceToHost, c->stream));
        NL_HIP(hipStreamSynchronize(c->stream));
    }
    c->nw_state = 0;
    return NL_OK;
}

extern "C" int nl_pinned_alloc(void **ptr, int64_t bytes, char *err, size_t errlen) {
    if (!ptr || bytes < 1) return nl_fail(err, errlen, NL_EINVAL, "bad pinned allocation request");
    hipError_t e = hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) return nl_fail(err, errlen, NL_ENOMEM, "hipHostMalloc(%lld): %s [out of memory]", (i64)bytes, hipGetErrorString(e));
    return NL_OK;
}
extern "C" int nl_pinned_free(void *ptr) { if (ptr) hipHostFree(ptr); return NL_OK; }
// page-lock memory the caller already owns (a numpy array): copies from / into it become asynchronous too
extern "C" int nl_host_register(void *ptr, int64_t bytes, char *err, size_t errlen) {
    if (!ptr || bytes < 1) return nl_fail(err, errlen, NL_EINVAL, "bad host registration request");
    hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return nl_fail(err, errlen, NL_EHIP, "hipHostRegister(%lld bytes): %s", (i64)bytes, hipGetErrorString(e)); }
    return NL_OK;
}
extern "C" int nl_host_unregister(void *ptr) { if (ptr) (void)hipHostUnregister(ptr); return NL_OK; }

static int stream_init(nl_ctx *c, char *err, size_t errlen) {
    if (c->copy_in) return NL_OK;
    NL_HIP(hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
    NL_HIP(hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) NL_HIP(hipEventCreateWithFlags(&c->ev_in[k], hipEventDisableTiming));
    NL_HIP(hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
    NL_HIP(hipEventCreateWithFlags(&c->ev_fetched, hipEventDisableTiming));
    NL_HIP(hipEventRecord(c->ev_fetched, c->copy_out));       // "nothing pending"
    return NL_OK;
}

// H2D of a whole local frame into input slot 0/1 on the copy stream (returns at once)
extern "C" int nl_input_load_async(nl_ctx *c, int slot, const void *host_pinned, int dtype, char *err, size_t errlen) {
    NL_ENTER_IO(c);
    const size_t es = dtype_size(dtype);
    if (!es || !host_pinned || slot < 0 || slot > 1) return nl_fail(err, errlen, NL_EINVAL, "bad async load arguments");
    int rc = stream_init(c, err, errlen);
    if (rc) return rc;
    if (c->d_in_slot[slot] && c->in_bytes[slot] < (size_t)c->n * es) { hipFree(c->d_in_slot[slot]); c->d_in_slot[slot] = nullptr; }
    if (!c->d_in_slot[slot]) { NL_HIP(hipMalloc(&c->d_in_slot[slot], (size_t)c->n * es)); c->in_bytes[slot] = (size_t)c->n * es; }
    c->in_dtype[slot] = dtype;
    NL_HIP(hipMemcpyAsync(c->d_in_slot[slot], host_pinned, (size_t)c->n * es, hipMemcpyHostToDevice, c->copy_in));
    NL_HIP(hipEventRecord(c->ev_in[slot], c->copy_in));
    return NL_OK;
}

// make the compute stream wait for that slot and use it as the resident input of the next nl_filter_begin
extern "C" int nl_input_select(nl_ctx *c, int slot, char *err, size_t errlen) {
    NL_ENTER(c);
    if (slot < 0 || slot > 1 || !c->d_in_slot[slot]) return nl_fail(err, errlen, NL_ESTATE, "input slot %d was never loaded", slot);
    NL_HIP(hipStreamWaitEvent(c->stream, c->ev_in[slot], 0));
    if (c->d_input && !c->input_borrowed) hipFree(c->d_input);
    c->d_input = c->d_in_slot[slot];
    c->input_borrowed = 1;
    c->input_dtype = c->in_dtype[slot];
    return NL_OK;
}

// D2D of the frame's outputs into staging volumes (compute stream), so the next frame may overwrite the originals
extern "C" int nl_outputs_stage(nl_ctx *c, int with_labels, char *err, size_t errlen) {
    NL_ENTER(c);
    int rc = stream_init(c, err, errlen);
    if (rc) return rc;
    if (!c->d_stage_fr) NL_HIP(hipMalloc((void **)&c->d_stage_fr, (size_t)c->n * 4));
    if (with_labels && !c->d_stage_lab) NL_HIP(hipMalloc((void **)&c->d_stage_lab, (size_t)c->n * 4));
    if (with_labels && c->i_labels < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_outputs_stage(with_labels) before nl_label_run");
    NL_HIP(hipStreamWaitEvent(c->stream, c->ev_fetched, 0));   // the previous frame's fetch has left the staging volumes
    NL_HIP(hipMemcpyAsync(c->d_stage_fr, c->f[c->i_vmax], (size_t)c->n * 4, hipMemcpyDeviceToDevice, c->stream));
    if (with_labels) NL_HIP(hipMemcpyAsync(c->d_stage_lab, c->f[c->i_labels], (size_t)c->n * 4, hipMemcpyDeviceToDevice, c->stream));
    NL_HIP(hipEventRecord(c->ev_staged, c->stream));
    return NL_OK;
}

// D2H of the staged outputs on the second copy stream (returns at once); nl_outputs_wait blocks until they landed
extern "C" int nl_outputs_fetch_async(nl_ctx *c, float *frangi_pinned, int32_t *labels_pinned, char *err, size_t errlen) {
    NL_ENTER_IO(c);
    if (!c->copy_out || !c->d_stage_fr) return nl_fail(err, errlen, NL_ESTATE, "nl_outputs_fetch_async before nl_outputs_stage");
    NL_HIP(hipStreamWaitEvent(c->copy_out, c->ev_staged, 0));
    if (frangi_pinned) NL_HIP(hipMemcpyAsync(frangi_pinned, c->d_stage_fr, (size_t)c->n * 4, hipMemcpyDeviceToHost, c->copy_out));
    if (labels_pinned) {
        if (!c->d_stage_lab) return nl_fail(err, errlen, NL_ESTATE, "labels were not staged");
        NL_HIP(hipMemcpyAsync(labels_pinned, c->d_stage_lab, (size_t)c->n * 4, hipMemcpyDeviceToHost, c->copy_out));
    }
    NL_HIP(hipEventRecord(c->ev_fetched, c->copy_out));
    return NL_OK;
}
extern "C" int nl_outputs_wait(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER_IO(c);
    if (c->ev_fetched) NL_HIP(hipEventSynchronize(c->ev_fetched));
    return NL_OK;
}

// ---- packed outputs (pack_out.inc) ----------------------------------------------------------------------------------
#define NL_PK_MAGIC 0x4b43415031304c4ell          // "NL01PACK"
struct PkHeader {                                 // 16 x int64, at the start of the blob; offsets in bytes from the blob's start
    long long magic, nz, ny, nx, wpr, n_values, n_runs, with_labels;
    long long off_fb, off_lb, off_fo, off_lo, off_fv, off_lr, total, reserved;
};
static inline size_t pk_pad(size_t b) { return (b + 15) & ~(size_t)15; }

// Frangi frame (+ labels) of the current frame -> packed blob in a staging buffer of the context (compute stream; the next
// frame may then overwrite the volumes).  *nbytes = size of the blob, 0 when the frame does not pack (more than a quarter
// of the voxels non-zero, or X-neighbours with different labels): use nl_outputs_stage / nl_outputs_fetch_async then.
extern "C" int nl_outputs_pack(nl_ctx *c, int with_labels, int64_t *nbytes, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_JOIN_SIDE(c);
    if (!nbytes) return nl_fail(err, errlen, NL_EINVAL, "nbytes is NULL");
    if (with_labels && c->i_labels < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_outputs_pack(with_labels) before nl_label_run");
    if (c->own_lo != 0 || c->own_hi != c->nzl) return nl_fail(err, errlen, NL_EINVAL, "packed outputs are for whole local volumes (no ghost planes)");
    int rc = stream_init(c, err, errlen);
    if (rc) return rc;
    const i64 rows = c->nzl * c->ny;
    const int wpr = (int)((c->nx + 63) / 64), nx = (int)c->nx;
    PkHeader h{};
    h.magic = NL_PK_MAGIC; h.nz = c->nzl; h.ny = c->ny; h.nx = c->nx; h.wpr = wpr; h.with_labels = with_labels ? 1 : 0;
    const size_t bits_b = pk_pad((size_t)rows * wpr * 8), off_b = pk_pad((size_t)(rows + 1) * 4);
    h.off_fb = pk_pad(sizeof(PkHeader)); h.off_lb = h.off_fb + bits_b; h.off_fo = h.off_lb + (with_labels ? bits_b : 0);
    h.off_lo = h.off_fo + off_b; h.off_fv = h.off_lo + (with_labels ? off_b : 0);
    const size_t cap = (size_t)h.off_fv + pk_pad((size_t)c->n) + 64;              // room for n / 4 items in total
    if (cap > c->pack_cap) {
        if (c->d_pack) hipFree(c->d_pack);
        c->d_pack = nullptr; c->pack_cap = 0;
        NL_HIP(hipMalloc(&c->d_pack, cap));
        c->pack_cap = cap;
    }
    char *blob = (char *)c->d_pack;
    NL_HIP(hipStreamWaitEvent(c->stream, c->ev_fetched, 0));     // the previous frame's blob has left the staging buffer
    unsigned int *cnt = c->d_rows;                                // per-row counts (Label's row tables are free between frames)
    unsigned int *flag = (unsigned int *)c->d_small + 40;
    unsigned long long *d_total = (unsigned long long *)c->d_small + 16;
    unsigned long long *h_tot = (unsigned long long *)c->h_small;
    NL_HIP(zero_small(flag, 4, c->stream));
    const unsigned grid = grid1d(rows * 64, 256, (i64)1 << 22);
    ProfScope ps(c, "pack");
    // counts, bit planes, row offsets
    pk_count_kernel<0><<<grid, 256, 0, c->stream>>>((const unsigned int *)c->f[c->i_vmax], (unsigned long long *)(blob + h.off_fb), cnt, rows, nx, wpr, flag);
    NL_CHECK_LAUNCH();
    if ((rc = scan_excl_u32(c, cnt, (unsigned int *)(blob + h.off_fo), rows, err, errlen))) return rc;
    NL_HIP(hipMemcpyAsync(&h_tot[0], d_total, 8, hipMemcpyDeviceToHost, c->stream));
    if (with_labels) {
        pk_count_kernel<1><<<grid, 256, 0, c->stream>>>((const unsigned int *)c->f[c->i_labels], (unsigned long long *)(blob + h.off_lb), cnt, rows, nx, wpr, flag);
        NL_CHECK_LAUNCH();
        if ((rc = scan_excl_u32(c, cnt, (unsigned int *)(blob + h.off_lo), rows, err, errlen))) return rc;
        NL_HIP(hipMemcpyAsync(&h_tot[1], d_total, 8, hipMemcpyDeviceToHost, c->stream));
    }
    NL_HIP(hipMemcpyAsync(&h_tot[2], flag, 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    h.n_values = (long long)h_tot[0]; h.n_runs = with_labels ? (long long)h_tot[1] : 0;
    const bool bad = (*(unsigned int *)&h_tot[2]) != 0u;
    h.off_lr = h.off_fv + pk_pad((size_t)h.n_values * 4);
    h.total = h.off_lr + pk_pad((size_t)h.n_runs * 4);
    if (bad || (size_t)h.total > c->pack_cap || h.n_values > 0xffffffffll || h.n_runs > 0xffffffffll) { *nbytes = 0; return NL_OK; }
    // header, the closing entries of the offset tables, then the items
    unsigned int *h_u = (unsigned int *)(h_tot + 4);
    h_u[0] = (unsigned int)h.n_values; h_u[1] = (unsigned int)h.n_runs;
    PkHeader *h_hdr = (PkHeader *)(h_tot + 8);
    *h_hdr = h;
    NL_HIP(hipMemcpyAsync(blob, h_hdr, sizeof(PkHeader), hipMemcpyHostToDevice, c->stream));
    NL_HIP(hipMemcpyAsync(blob + h.off_fo + (size_t)rows * 4, &h_u[0], 4, hipMemcpyHostToDevice, c->stream));
    if (with_labels) NL_HIP(hipMemcpyAsync(blob + h.off_lo + (size_t)rows * 4, &h_u[1], 4, hipMemcpyHostToDevice, c->stream));
    if (h.n_values) pk_emit_kernel<0><<<grid, 256, 0, c->stream>>>((const unsigned int *)c->f[c->i_vmax], (const unsigned long long *)(blob + h.off_fb),
                                                                 (const unsigned int *)(blob + h.off_fo), (unsigned int *)(blob + h.off_fv), rows, nx, wpr);
    if (h.n_runs) pk_emit_kernel<1><<<grid, 256, 0, c->stream>>>((const unsigned int *)c->f[c->i_labels], (const unsigned long long *)(blob + h.off_lb),
                                                               (const unsigned int *)(blob + h.off_lo), (unsigned int *)(blob + h.off_lr), rows, nx, wpr);
    NL_CHECK_LAUNCH();
    NL_HIP(hipEventRecord(c->ev_staged, c->stream));
    // the pinned scratch the header travelled through is reused by the next entry point: let the copies finish
    NL_HIP(hipStreamSynchronize(c->stream));
    *nbytes = h.total;
    return NL_OK;
}

// D2H of the packed blob on the second copy stream (returns at once); nl_outputs_wait blocks until it landed
extern "C" int nl_outputs_fetch_packed_async(nl_ctx *c, void *host_pinned, int64_t nbytes, char *err, size_t errlen) {
    NL_ENTER_IO(c);
    if (!c->copy_out || !c->d_pack) return nl_fail(err, errlen, NL_ESTATE, "nl_outputs_fetch_packed_async before nl_outputs_pack");
    if (!host_pinned || nbytes < (int64_t)sizeof(PkHeader) || (size_t)nbytes > c->pack_cap) return nl_fail(err, errlen, NL_EINVAL, "bad packed fetch arguments");
    NL_HIP(hipStreamWaitEvent(c->copy_out, c->ev_staged, 0));
    NL_HIP(hipMemcpyAsync(host_pinned, c->d_pack, (size_t)nbytes, hipMemcpyDeviceToHost, c->copy_out));
    NL_HIP(hipEventRecord(c->ev_fetched, c->copy_out));
    return NL_OK;
}

// Host only: expand a packed blob into the caller's dense arrays (labels may be NULL).  zero_fill = 0 when the arrays are
// known to hold zeros already (a freshly created file or calloc'ed array): only rows with content are touched then.
extern "C" int nl_outputs_unpack(const void *blob_, int64_t nbytes, float *frangi, int32_t *labels, int64_t dst_elems, int zero_fill,
                                 int threads, char *err, size_t errlen) {
    const char *blob = (const char *)blob_;
    if (!blob || nbytes < (int64_t)sizeof(PkHeader) || !frangi) return nl_fail(err, errlen, NL_EINVAL, "bad unpack arguments");
    PkHeader h;
    memcpy(&h, blob, sizeof(h));
    if (h.magic != NL_PK_MAGIC || h.total > nbytes || h.total < (long long)sizeof(PkHeader) || h.nz < 0 || h.ny < 0 || h.nx < 0 ||
        h.wpr != (h.nx + 63) / 64)
        return nl_fail(err, errlen, NL_EINVAL, "not a packed-output blob");
    if (labels && !h.with_labels) return nl_fail(err, errlen, NL_EINVAL, "the blob holds no labels");
    const i64 rows = h.nz * h.ny, nx = h.nx;
    const int wpr = (int)h.wpr;
    // the destination arrays are the caller's: the header of a blob from another context (or a damaged one) must not decide
    // how far they are written
    if (dst_elems != rows * nx)
        return nl_fail(err, errlen, NL_EINVAL, "the blob describes a %lld x %lld x %lld volume, the destination holds %lld elements",
                       (i64)h.nz, (i64)h.ny, (i64)h.nx, (i64)dst_elems);
    // every section inside the blob
    {
        const long long bits_b = rows * wpr * 8, off_b = (rows + 1) * 4;
        auto inside = [&](long long off, long long len) { return off >= (long long)sizeof(PkHeader) && len >= 0 && off <= h.total && len <= h.total - off; };
        bool ok = h.n_values >= 0 && h.n_runs >= 0 && inside(h.off_fb, bits_b) && inside(h.off_fo, off_b) && inside(h.off_fv, h.n_values * 4);
        if (h.with_labels) ok = ok && inside(h.off_lb, bits_b) && inside(h.off_lo, off_b) && inside(h.off_lr, h.n_runs * 4);
        if (!ok) return nl_fail(err, errlen, NL_EINVAL, "packed-output blob: a section lies outside its %lld bytes", (i64)h.total);
        // the per-row item offsets are what indexes the item arrays: monotone and inside the counts
        const unsigned int *fo_ = (const unsigned int *)(blob + h.off_fo);
        if (fo_[0] != 0 || (long long)fo_[rows] > h.n_values) ok = false;
        if (h.with_labels) { const unsigned int *lo_ = (const unsigned int *)(blob + h.off_lo); if (lo_[0] != 0 || (long long)lo_[rows] > h.n_runs) ok = false; }
        if (!ok) return nl_fail(err, errlen, NL_EINVAL, "packed-output blob: row offsets disagree with the item counts");
    }
    const unsigned long long *fb = (const unsigned long long *)(blob + h.off_fb), *lb = (const unsigned long long *)(blob + h.off_lb);
    const unsigned int *fo = (const unsigned int *)(blob + h.off_fo), *lo = (const unsigned int *)(blob + h.off_lo);
    const float *fv = (const float *)(blob + h.off_fv);
    const int32_t *lr = (const int32_t *)(blob + h.off_lr);
    auto work = [&](i64 r0, i64 r1) {
        for (i64 row = r0; row < r1; ++row) {
            float *dst = frangi + row * nx;
            if (zero_fill) memset(dst, 0, (size_t)nx * 4);
            unsigned int k = fo[row];
            if (fo[row + 1] != k) {
                const unsigned long long *bw = fb + row * wpr;
                for (int w = 0; w < wpr; ++w) {
                    unsigned long long b = bw[w];
                    float *d64 = dst + (i64)w * 64;
                    while (b) { d64[__builtin_ctzll(b)] = fv[k++]; b &= b - 1; }
                }
            }
            if (!labels) continue;
            int32_t *ld = labels + row * nx;
            if (zero_fill) memset(ld, 0, (size_t)nx * 4);
            unsigned int q = lo[row];
            if (lo[row + 1] == q) continue;
            const unsigned long long *bw = lb + row * wpr;
            int32_t cur = 0;
            bool open = false;                            // the previous word ended inside a run
            for (int w = 0; w < wpr; ++w) {
                unsigned long long b = bw[w];
                int32_t *d64 = ld + (i64)w * 64;
                if (!b) { open = false; continue; }
                bool first = true;
                while (b) {
                    const int s = __builtin_ctzll(b);
                    const unsigned long long rest = ~(b >> s);
                    const int len = rest ? __builtin_ctzll(rest) : 64 - s;
                    if (!(first && s == 0 && open)) cur = lr[q++];
                    for (int t = 0; t < len; ++t) d64[s + t] = cur;
                    b = (s + len >= 64) ? 0ull : (b & ~(((1ull << len) - 1ull) << s));
                    first = false;
                }
                open = (bw[w] >> 63) != 0;
            }
        }
    };
    int nt = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    if (rows < 4096) nt = 1;
    if (nt == 1) { work(0, rows); return NL_OK; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, rows * t / nt, rows * (t + 1) / nt);
    for (auto &th : pool) th.join();
    return NL_OK;
}

// ---------------------------------------------------------------------------------- debug -------
extern "C" int nl_ctx_info(nl_ctx *c, const char *key, double *value) {
    if (!c || !key || !value) return NL_EINVAL;
    if (!strcmp(key, "fast_div")) *value = c->fast_div2 ? 2 : c->fast_div;      // 2: two-instruction division proven, 1: three, 0: float64
    else if (!strcmp(key, "hessian_tile_rows")) *value = hv_rs(c) ? 2 * hv_rs(c) : hm_ty();
    else if (!strcmp(key, "vesselness_one_pass")) *value = c->spec_ok;
    else if (!strcmp(key, "chain_available")) *value = (!c->two_d && c->spec_ok && hv_rs(c)) ? 1 : 0;   // nl_chain_begin's own precondition
    else if (!strcmp(key, "last_fsq_min")) *value = c->last_fsq_min;
    else if (!strcmp(key, "last_spec_overflow")) *value = c->last_spec_overflow;
    else if (!strcmp(key, "last_label_sparse")) *value = c->last_label_sparse;
    else if (!strcmp(key, "device_bytes")) *value = (double)nl_ctx_bytes(c->nzl, c->ny, c->nx);
    else return NL_EINVAL;
    return NL_OK;
}

extern "C" int nl_debug_eig_frangi(nl_ctx *c, const float *h6, int64_t n, int impl, float alpha_sq, float beta_sq,
                                   float gamma_sq, float *out4, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!h6 || !out4 || n < 1 || n * 6 > c->n) return nl_fail(err, errlen, NL_EINVAL, "bad debug batch (n=%lld)", (i64)n);
    float *d_in = c->f[(c->i_gauss + 1) % 3], *d_out = c->f[(c->i_gauss + 2) % 3];
    NL_HIP(hipMemcpyAsync(d_in, h6, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    debug_eig_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(d_in, n, impl, alpha_sq, beta_sq, gamma_sq, d_out);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(out4, d_out, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

// --------------------------------------------------------------------------------- timing -------
extern "C" int nl_timer_begin(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_HIP(hipEventRecord(c->t0, c->stream));
    return NL_OK;
}
extern "C" int nl_timer_end_ms(nl_ctx *c, float *ms, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_HIP(hipEventRecord(c->t1, c->stream));
    NL_HIP(hipEventSynchronize(c->t1));
    float t = 0;
    NL_HIP(hipEventElapsedTime(&t, c->t0, c->t1));
    if (ms) *ms = t;
    return NL_OK;
}
extern "C" int nl_prof_enable(nl_ctx *c, int on) {
    if (!c) return NL_OK;
    if (on && !c->prof_on) {                 // stock the pool outside the timed region
        hipSetDevice(c->device);
        size_t used = 0;
        for (auto &kv : c->prof) used += kv.second.size();
        while (c->prof_pool.size() + used < 1024) {
            ProfRec r;
            if (hipEventCreate(&r.a) != hipSuccess) break;
            if (hipEventCreate(&r.b) != hipSuccess) { hipEventDestroy(r.a); break; }
            c->prof_pool.push_back(r);
        }
    }
    c->prof_on = on;
    return NL_OK;
}
extern "C" int nl_prof_reset(nl_ctx *c) {
    if (!c) return NL_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->side);
    for (auto &kv : c->prof) for (auto &r : kv.second) c->prof_pool.push_back(r);       // kept for the next scopes
    c->prof.clear();
    c->prof_sum.clear();
    return NL_OK;
}
extern "C" int nl_prof_get(nl_ctx *c, const char *name, double *ms, int64_t *launches) {
    if (!c || !name) return NL_EINVAL;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->side);
    if (c->xstream) hipStreamSynchronize(c->xstream);
    prof_harvest(c, true);
    double tot = 0; int64_t k = 0;
    auto it = c->prof_sum.find(name);
    if (it != c->prof_sum.end()) { tot = it->second.first; k = it->second.second; }
    if (ms) *ms = tot;
    if (launches) *launches = k;
    return NL_OK;
}
