// libnellie_hip.so -- hand-written HIP for gfx950 (MI355X): Nellie's Filter -> Label hot path.
// C-ABI in include/nellie_amd.h.  Compile with -ffp-contract=off: every float operation
// below is meant to round exactly where numpy/scipy round.
#include "nl_host.h"

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
        // The installed ROCm's copy BY PATH first: a bare "librccl.so.1" is answered with whatever object of that SONAME the process
        // already holds -- e.g. the RCCL a PyTorch wheel bundles (built against another HIP runtime: ncclCommInitRank then fails with
        // "unhandled cuda error"; found when a test imported torch into the pytest process, round 5).
        std::string rp;
        if (const char *e = getenv("ROCM_PATH")) rp = std::string(e) + "/lib/librccl.so.1";
        const char *names[] = {rp.c_str(), "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
        for (const char *n : names) { if (!*n) continue; api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.handle) break; }
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
static ncclResult_t comm_missing() { return (ncclResult_t)lb::kMissing; }
ncclResult_t CommApi::GetUniqueId(ncclUniqueId *id) { return rccl_real().ok ? rccl_real().GetUniqueId(id) : comm_missing(); }
ncclResult_t CommApi::CommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
    if (lb::is_loopback_id(id.internal)) return lb::comm_init(comm, world, id.internal, rank);
    if (!rccl_real().ok) return comm_missing();
    const ncclResult_t r = rccl_real().CommInitRank(comm, world, id, rank);
    if (r == ncclSuccess) ++n_real;
    return r;
}
ncclResult_t CommApi::CommDestroy(ncclComm_t comm) {
    if (lb::is_ours(comm)) return lb::comm_destroy(comm);
    if (!rccl_real().ok) return comm_missing();
    --n_real;
    return rccl_real().CommDestroy(comm);
}
const char *CommApi::GetErrorString(ncclResult_t r) {
    if ((int)r == lb::kMissing) return "librccl.so could not be loaded";
    if (rccl_real().handle && rccl_real().ok) return rccl_real().GetErrorString(r);
    switch (r) {
        case ncclInvalidArgument: return "invalid argument (loopback transport)";
        case ncclSystemError: return "rendezvous timed out or a peer failed (loopback transport)";
        case ncclUnhandledCudaError: return "HIP error (loopback transport)";
        default: return "error (loopback transport)";
    }
}
ncclResult_t CommApi::GroupStart() {
    lb::group_start();
    return n_real.load() > 0 ? rccl_real().GroupStart() : ncclSuccess;
}
ncclResult_t CommApi::GroupEnd() {
    const ncclResult_t r = lb::group_end();
    const ncclResult_t q = n_real.load() > 0 ? rccl_real().GroupEnd() : ncclSuccess;
    return r != ncclSuccess ? r : q;
}
ncclResult_t CommApi::Send(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    if (lb::is_ours(comm)) return lb::submit(lb::Op{0, buf, nullptr, count, dt, ncclSum, peer, (lb::Comm *)comm, st});
    return rccl_real().Send(buf, count, dt, peer, comm, st);
}
ncclResult_t CommApi::Recv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    if (lb::is_ours(comm)) return lb::submit(lb::Op{1, nullptr, buf, count, dt, ncclSum, peer, (lb::Comm *)comm, st});
    return rccl_real().Recv(buf, count, dt, peer, comm, st);
}
ncclResult_t CommApi::AllReduce(const void *src, void *dst, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    if (lb::is_ours(comm)) return lb::submit(lb::Op{2, src, dst, count, dt, op, -1, (lb::Comm *)comm, st});
    return rccl_real().AllReduce(src, dst, count, dt, op, comm, st);
}
ncclResult_t CommApi::AllGather(const void *src, void *dst, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st) {
    if (lb::is_ours(comm)) return lb::submit(lb::Op{3, src, dst, count, dt, ncclSum, -1, (lb::Comm *)comm, st});
    return rccl_real().AllGather(src, dst, count, dt, comm, st);
}
ncclResult_t CommApi::Broadcast(const void *src, void *dst, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t st) {
    if (lb::is_ours(comm)) return lb::submit(lb::Op{4, src, dst, count, dt, ncclSum, root, (lb::Comm *)comm, st});
    return rccl_real().Broadcast(src, dst, count, dt, root, comm, st);
}
CommApi &rccl() { static CommApi api; return api; }

#include "sampling.inc"
#include "hessian.inc"
#include "hessian_pair.inc"
#include "hv_launch.h"
#include "filter2d.inc"
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

// fused Y+X Gaussian: tiled / register-blocked X pass (default) or the row-at-a-time kernel (NELLIE_GYX_TILE=0)
bool gyx_tiled() {
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
    if (v < 0) { const char *e = getenv("NELLIE_HV_RS"); v = e ? atoi(e) : 8; if (v != 0 && v != 8 && !(NL_HV_VARIANTS && v == 16)) v = 8; }
    return v;
}
// the pair kernel addresses the planes of a Z chunk through one buffer resource (32-bit byte offsets)
static int hv_rs(const nl_ctx *c) { return ((i64)(HM_ZCHUNK + 4) * c->ny * c->nx * 4 < ((i64)1 << 32)) ? hv_rs_env() : 0; }
static HessP hessp(const nl_ctx *c);
static HessDv<1> hessdv_fast(const nl_ctx *c) { return hessdv_fast(hessp(c)); }      // (dv_* and the HessP forms: hv_launch.h)
static HessDv<0> hessdv_exact(const nl_ctx *c) { return hessdv_exact(hessp(c)); }
// the pair walk (its own translation unit, hv_launch.h): division variant as proven for this context's divisors
static int hv_fastv(const nl_ctx *c) { return c->fast_div2 ? 2 : (c->fast_div ? 1 : 0); }
// pair-rows per lane of the pair walk (NELLIE_HV_NP=2: four voxels per lane at 2 waves / SIMD -- the round-5 experiment, hessian_pair.inc;
// only instantiated for the 16-row tile with the two-instruction division)
static int hv_np(const nl_ctx *c) {
#if NL_HV_VARIANTS
    const char *e = getenv("NELLIE_HV_NP");            // read per call: tests switch it inside one process
    return (e && atoi(e) == 2 && hv_rs(c) == 8 && hv_fastv(c) == 2) ? 2 : 1;
#else
    (void)c;
    return 1;
#endif
}
// The wave-autonomous walk (round 6, hessian_dpp.inc): strips of 4 rows x 60 columns per wave, no LDS.  Statistics and one-pass modes; the
// statistics and known-threshold modes (MODE 0 / 1: the two-pass fallback of a scale) stay with the pair kernel.  NELLIE_HV_DPP=1 selects it
// for the one-pass walk (default: the pair kernel -- measured equal at 1024^3, profiles/r06_walk_dpp_production.txt).
#define HD_SR 4
#define HD_COLS 60
static bool hv_dpp(const nl_ctx *c) {
    const char *e = getenv("NELLIE_HV_DPP");            // read per call: tests switch it inside one process
    return hv_rs(c) != 0 && e && atoi(e) == 1;
}
// Planes per wave: 128 where that leaves several rounds of waves (a chunk's prologue -- four planes loaded, the first derivatives of one --
// is paid half as often), 32 on small frames (a 128 x 512 x 512 frame has 1152 strips: 64-plane chunks would fill three quarters of the
// 3072 wave slots once), else 64.
static int hd_zchunk(const nl_ctx *c, i64 nz, i64 strips) {
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("NELLIE_HV_ZCHUNK"); forced = e ? atoi(e) : 0; }
    auto fits = [&](int z) { return (i64)(z + 4) * c->ny * c->nx * 4 < ((i64)1 << 32); };
    if (forced > 0 && (forced & 3) == 0 && fits(forced)) return forced;
    if (fits(128) && strips * ((nz + 127) / 128) >= 8192) return 128;
    if (fits(64) && strips * ((nz + 63) / 64) >= 4096) return 64;
    return 32;
}

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
int64_t vq_alloc_entries(int64_t nzl, int64_t ny, int64_t nx) {
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
    if (c->mk_act) hipFree(c->mk_act);
    if (c->d_vq_count) hipFree(c->d_vq_count);
    if (c->d_rows) hipFree(c->d_rows);
    if (c->d_ag) hipFree(c->d_ag);
    if (c->d_sl) hipFree(c->d_sl);
    if (c->d_seg_done) hipFree(c->d_seg_done);
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


int upload_convert(nl_ctx *c, const void *host, int dtype, float *dst, i64 count, char *err, size_t errlen) {
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
    // a frame abandoned between nl_gauss_step_ahead and nl_gauss_commit (an exception in the host's scale loop) leaves a cascade step on the
    // side stream that still writes the ping-pong volumes: this frame's first kernels come after it
    if (c->ahead_pending) NL_HIP(hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
    c->ahead_pending = 0;
    c->fsq_cache_valid = 0;
    NL_HIP(zero_small((char *)c->d_small + (52 << 10), 4, c->stream));
    c->vmax_zero_lo = c->vmax_zero_hi = 0;
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
    // a frame abandoned between nl_gauss_step_ahead and nl_gauss_commit (an exception in the host's scale loop) leaves a cascade step on the
    // side stream that still writes the ping-pong volumes: this frame's first kernels come after it
    if (c->ahead_pending) NL_HIP(hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
    c->ahead_pending = 0;
    c->fsq_cache_valid = 0;
    NL_HIP(zero_small((char *)c->d_small + (52 << 10), 4, c->stream));
    c->vmax_zero_lo = c->vmax_zero_hi = 0;
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


// Y and X passes in one kernel (gauss_yx_tile_kernel): equal radii up to GM_MAX_R that fit the Y axis
static bool gauss_can_yx(const nl_ctx *c, const double *wy, int ry, const double *wx, int rx) {
    return wy && wx && ry == rx && ry >= 1 && ry <= GM_MAX_R && ry <= c->ny && !getenv("NELLIE_NO_FUSED_YX");
}
// Ping-pong volumes a cascade step writes one after the other: 1 (fused Z+Y+X), 2 (Z, then Y+X) or 3 (one per axis: radii beyond
// GM_MAX_R, unequal in-plane radii).  The third destination of a three-pass step is the step's own SOURCE volume.
static int gauss_step_volumes(const nl_ctx *c, const double *wz, const double *wy, int ry, const double *wx, int rx) {
    return (wz ? 1 : 0) + (gauss_can_yx(c, wy, ry, wx, rx) ? 1 : (wy ? 1 : 0) + (wx ? 1 : 0));
}

extern "C" int nl_gauss_step(nl_ctx *c, const double *wz, int rz, const double *wy, int ry, const double *wx, int rx,
                             int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    if (c->ahead_pending) return nl_fail(err, errlen, NL_ESTATE, "nl_gauss_step while a step enqueued ahead is uncommitted");
    if (z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    if (c->halo_pending) {                       // ghost planes of the source volume still travelling (nl_halo_exchange_at, async)
        // "halo_wait": the event pair brackets nothing but the wait, so its time is what the exchange of this cascade step EXPOSED on the
        // main stream (0 when the planes arrived while the scale's walk ran) -- the figure bench.py's N > 1 line reports per rank
        ProfScope ps(c, "halo_wait");
        NL_HIP(hipStreamWaitEvent(c->stream, c->ev_x_done, 0));
        c->halo_pending = 0;
    }
    const VolGeom v = geom(c);
    // three ping-pong volumes f[0..2]: the source of a pass is dead once the pass has run,
    // so "the next one" is always a legal destination
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)(z1 - z0));
    int src = c->i_gauss;
    const float *srcp = gauss_cur(c);       // the source of the next pass (the borrowed input before the first one)
    GaussW gw;
    int rc;
    const bool can_yx = gauss_can_yx(c, wy, ry, wx, rx);
    bool fused_yx = false;
    // The whole step in one kernel (gauss_zyx.inc) when the radii have an instantiation and Y and X share their weights, which is
    // what Filter asks for (sigma_vec = (s / z_ratio, s, s)) -- on volumes that do not fit the caches: the fused kernel trades HBM
    // traffic for redundant float64 work, and a 128 x 512 x 512 frame (134 MB a volume) streams from the 256 MB Infinity Cache anyway
    // (gauss ms per frame, two kernels / fused: 128 x 512 x 512 0.62 / 0.77; 256 x 512 x 512 1.32 / 1.24; 512^3 2.24 / 2.16; 256 x 1024^2
    // 4.44 / 4.00; 1024^3 17.7 / 15.3).
    // NELLIE_GAUSS_FUSED=0 / 1: never / whenever the radii allow (tests run the fused kernel on small volumes that way).
    const char *e_fz = getenv("NELLIE_GAUSS_FUSED");
    const bool fused_zyx = e_fz ? atoi(e_fz) != 0 : c->n >= ((i64)1 << 26);
    if (fused_zyx && wz && can_yx && gyx_tiled() && gl_zyx_ok(c, rz, ry, c->f[(src + 1) % 3]) &&
        memcmp(wy, wx, (size_t)(2 * ry + 1) * sizeof(double)) == 0) {
        GaussW gy;
        if ((rc = fill_gw(gw, wz, rz, err, errlen)) || (rc = fill_gw(gy, wy, ry, err, errlen))) return rc;
        if ((z0 - rz < 0 && c->gz0 > 0) || (z1 - 1 + rz >= c->nzl && c->gz0 + c->nzl < c->gnz))
            return nl_fail(err, errlen, NL_EINVAL, "Z pass of radius %d on planes [%lld,%lld) reaches outside the local slab", rz, (i64)z0, (i64)z1);
        const int dst = (src + 1) % 3;
        char scope[32];
        snprintf(scope, sizeof scope, "gauss_zyx<%d,%d>", rz, ry);        // one timer per kernel instantiation, as a profiler lists them
        ProfScope ps(c, scope);
        // the frame's first step also zeroes the running scale maximum on its planes (see gauss_zyx.inc: zero_out)
        static int zero_in_passing = -1;
        if (zero_in_passing < 0) { const char *e = getenv("NELLIE_ZERO_IN_GAUSS"); zero_in_passing = (e && !atoi(e)) ? 0 : 1; }
        const bool zero = zero_in_passing && c->mask_slots_used == 0 && c->vmax_zero_hi == 0 && c->stream != c->side;
        (void)gl_zyx(c, rz, ry, srcp, c->f[dst], v, z0, z1, gauss_ws_of(gw), gauss_ws_of(gy), zero ? c->f[c->i_vmax] : nullptr);
        if (zero) { c->vmax_zero_lo = z0; c->vmax_zero_hi = z1; }
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
        fused_yx = true;
        wz = nullptr;
    }
    if (wz) {
        if ((rc = fill_gw(gw, wz, rz, err, errlen))) return rc;
        // every tap must land inside the local slab unless it reflects at a true face
        if ((z0 - rz < 0 && c->gz0 > 0) || (z1 - 1 + rz >= c->nzl && c->gz0 + c->nzl < c->gnz))
            return nl_fail(err, errlen, NL_EINVAL, "Z pass of radius %d on planes [%lld,%lld) reaches outside the local slab", rz, (i64)z0, (i64)z1);
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_z");
        if (!gl_fast(0, c, srcp, c->f[dst], v, z0, z1, gw)) gl_axis(0, false, c, grid, srcp, c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
    }
    if (!fused_yx && can_yx) {
        GaussW gy, gx;
        if ((rc = fill_gw(gy, wy, ry, err, errlen))) return rc;
        if ((rc = fill_gw(gx, wx, rx, err, errlen))) return rc;
        const GaussWS wsy = gauss_ws_of(gy), wsx = gauss_ws_of(gx);
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_yx");
        const dim3 g2((unsigned)((c->nx + GYX_COLS - 1) / GYX_COLS), (unsigned)((c->ny + v.chunk - 1) / v.chunk), (unsigned)(z1 - z0));
        (void)gl_yx(c, gyx_tiled(), false, ry, srcp, c->f[dst], v, z0, z1, wsy, wsx, g2);
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
        fused_yx = true;
    }
    if (wy && !fused_yx) {
        if ((rc = fill_gw(gw, wy, ry, err, errlen))) return rc;
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_y");
        if (!gl_fast(1, c, srcp, c->f[dst], v, z0, z1, gw)) gl_axis(1, false, c, grid, srcp, c->f[dst], v, z0, z1, gw);
        NL_CHECK_LAUNCH();
        src = dst; srcp = c->f[dst];
    }
    if (wx && !fused_yx) {
        if ((rc = fill_gw(gw, wx, rx, err, errlen))) return rc;
        const int dst = (src + 1) % 3;
        ProfScope ps(c, "gauss_x");
        if (!gl_fast(2, c, srcp, c->f[dst], v, z0, z1, gw)) gl_axis(2, false, c, grid, srcp, c->f[dst], v, z0, z1, gw);
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
    // A step of three passes would write its third pass into the volume it started from -- the Gaussian the current scale is
    // still reading (found by the fuzzer's explicit sigma lists, round 5: radii beyond GM_MAX_R).  Callers ask
    // nl_ctx_info("gauss_yx_max_r") and run such steps in order.
    if (gauss_step_volumes(c, wz, wy, ry, wx, rx) > 2)
        return nl_fail(err, errlen, NL_ESTATE, "a cascade step of three passes (radii %d / %d / %d) cannot run ahead: it needs the current Gaussian's volume", rz, ry, rx);
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
int fetch_counted(nl_ctx *c, const float *stage, const unsigned int *d_n, i64 max_count, float *out, i64 cap, int64_t *n, char *err, size_t errlen) {
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
        if (false) {}
        else if (hv_rs(c)) {
            const int rsv = hv_rs(c), ntyv = (int)((c->ny + 2 * rsv - 1) / (2 * rsv));
            NL_HIP(nl_hv_launch(HvLaunch{0, rsv, hv_np(c), hv_fastv(c), (unsigned)(ntx * ntyv * nzc), c->stream, gauss_cur(c), nullptr, nullptr, 0, geom(c),
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

#define NL_NAN_FLAG_OFF (52 << 10)      // byte offset in d_small of the "NaN Hessian solved" word (VessP::nan_flag), zeroed per frame
// the planes [z0, z1) of the vesselness volume are all zero already: the frame's first cascade step did it in passing (gauss_zyx.inc)
static inline bool vmax_is_zero(const nl_ctx *c, i64 z0, i64 z1) { return c->vmax_zero_hi > 0 && c->vmax_zero_lo <= z0 && z1 <= c->vmax_zero_hi; }
static VessP make_vessp(nl_ctx *c, float gamma_sq, float alpha_sq, float beta_sq, int use_thr, float thr) {
    VessP vp{};
    vp.gamma_sq = gamma_sq; vp.alpha_sq = alpha_sq; vp.beta_sq = beta_sq; vp.use_thr = use_thr; vp.thr = thr;
    vp.max_abs = c->frob_max_abs; vp.max_finite = c->frob_max_finite;
    vp.cnt_lo = (int)c->own_lo; vp.cnt_hi = (int)c->own_hi;
    vp.first = c->mask_slots_used == 0 ? 1 : 0;
    vp.fsq_min = mask_threshold_on_fsq(c->frob_max_abs, use_thr, thr);
    vp.m_inf = use_thr ? (c->frob_max_finite > thr) : (c->frob_max_finite > 0.0f);
    vp.all_ones = (use_thr && thr == -INFINITY) ? 1 : 0;       // the host's way of saying mask=False (pipeline.py: thr_cmp = -inf)
    vp.nan_flag = (unsigned int *)((char *)c->d_small + NL_NAN_FLAG_OFF);
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
        if (rs) vp.qcap *= 2 * hv_np(c);                 // a wave owns two (four) row segments
        hipStream_t hs = c->stream;
        if (rs && hv_dpp(c)) {
            // wave-autonomous walk: one region per strip and chunk; the queue's bytes are shared out evenly (0.47 entries per voxel of a strip
            // where the pair kernel has 0.5: its strips overhang the volume by up to 59 columns)
            const int ntxd = (int)((c->nx + HD_COLS - 1) / HD_COLS), ntyd = (int)((c->ny + HD_SR - 1) / HD_SR);
            const int zd = hd_zchunk(c, z1 - z0, (i64)ntxd * ntyd);
            const i64 nzd = (z1 - z0 + zd - 1) / zd;
            const i64 nreg = (i64)ntxd * ntyd * nzd;
            i64 cap = vq_alloc_entries(c->nzl, c->ny, c->nx) / nreg;
            if (cap > (i64)HD_SR * 64 * zd) cap = (i64)HD_SR * 64 * zd;
            if (cap > (1 << 24) - 1) cap = (1 << 24) - 1;
            if (cap >= 64 && 2 * nreg <= vq_alloc_regions(c->nzl, c->ny, c->nx)) {     // (region counts + the strips' h_mask counts)
                vp.zchunk = zd; vp.qcap = (int)cap;
                // the strips OR their bits into the words (two strips share a word): the slot's planes start from zero
                NL_HIP(hipMemsetAsync(cm + z0 * c->ny * wpr, 0, (size_t)(z1 - z0) * c->ny * wpr * 8, hs));
                NL_HIP(nl_hv_launch(HvLaunch{2, HD_SR, 0, hv_fastv(c), (unsigned)nreg, hs, gauss_cur(c), cm, pm, wpr, geom(c), hessp(c), vp, vq, (int)z0, (int)z1,
                                             ntxd, ntyd, res, d_cnt, dev_lohi}));
                NL_CHECK_LAUNCH();
                c->spec_nregions = (unsigned)nreg;
                c->spec_qcap = vp.qcap;
                goto launched;
            }
        }
#define NL_DEV_LOHI dev_lohi
#define NL_LAUNCH_SPEC(TYV, FASTV, HR)                                                                                    \
        hessian_g_kernel<2, TYV, FASTV><<<nblocks, HGCfg<TYV>::NT, HGCfg<TYV>::lds_bytes(), c->stream>>>(                 \
            gauss_cur(c), cm, pm, wpr, geom(c), HR, vp, vq, (int)z0, (int)z1, ntx, nty, res, d_cnt)
        if (rs) NL_HIP(nl_hv_launch(HvLaunch{2, rs, hv_np(c), hv_fastv(c), nblocks, hs, gauss_cur(c), cm, pm, wpr, geom(c), hessp(c), vp, vq, (int)z0, (int)z1,
                                             ntx, nty, res, d_cnt, NL_DEV_LOHI}));
        else if (ty == 8) { if (c->fast_div) NL_LAUNCH_SPEC(8, true, hessdv_fast(c)); else NL_LAUNCH_SPEC(8, false, hessdv_exact(c)); }
        else { if (c->fast_div) NL_LAUNCH_SPEC(16, true, hessdv_fast(c)); else NL_LAUNCH_SPEC(16, false, hessdv_exact(c)); }
#undef NL_LAUNCH_SPEC
#undef NL_DEV_LOHI
        NL_CHECK_LAUNCH();
        c->spec_nregions = nblocks * (unsigned)(rs ? rs / hv_np(c) : ty);
        c->spec_qcap = vp.qcap;
launched: ;
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
static int resolve_enqueue(nl_ctx *c, VessP vp, unsigned long long *d_cnt, const float *dev_params, char *err, size_t errlen, bool force_side = false) {
    const i64 plane = c->ny * c->nx, z0 = c->spec_z0, z1 = c->spec_z1;
    const bool side = force_side || resolve_on_side();
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
        if (vp.first && !vmax_is_zero(c, z0, z1))
            NL_HIP(hipMemsetAsync(c->f[c->i_vmax] + z0 * plane, 0, (size_t)(z1 - z0) * plane * 4, st));
        c->vmax_zero_hi = 0;
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

// Device chain (round 5, second session): the resolve kernel of scale s needs nothing the cascade step of scale s+1 touches, and the ten
// threshold kernels of scale s+1 (one wave or a 10^6-point lattice each: ~0.15 ms of a nearly idle GPU) need nothing the resolve kernel
// writes.  So nl_chain_scale(s) only NOTES the resolve launch; nl_chain_scale(s+1) -- called after the host has enqueued cascade step
// s+1 -- starts it on the side stream behind that step, and the threshold kernels run beside it on the main stream; the walk of scale
// s+1 joins (spec_enqueue: NL_JOIN_SIDE).  Beside the cascade step itself the resolve kernel only costs (both are bound by the float64
// pipe: NELLIE_RESOLVE_OVERLAP, profiles/r05_walk_variants_1024cube.txt).  Single context, volumes of 2^26 voxels and more (below, the
// side stream carries the cascade step that runs ahead); NELLIE_RESOLVE_DEFER=0: off.
static bool resolve_defer_ok(const nl_ctx *c) {
    const char *e = getenv("NELLIE_RESOLVE_DEFER");               // read per call: tests switch it (2: whatever the size, for the small volumes of the suite)
    const int on = e ? atoi(e) : 1;
    if (on == 2) return !c->comm && !resolve_on_side();
    return on && !c->comm && !c->ahead_pending && !resolve_on_side() && c->n >= ((i64)1 << 26);
}
static int resolve_deferred_launch(nl_ctx *c, bool on_side, char *err, size_t errlen) {
    if (!c->def_resolve) return NL_OK;
    c->def_resolve = 0;
    NL_JOIN_SIDE(c);          // the exact round of that scale ran on the side stream: whatever the main stream does next comes after it
    VessP vp;
    static_assert(sizeof(VessP) <= sizeof(((nl_ctx *)nullptr)->def_vp), "nl_ctx::def_vp holds a VessP");
    memcpy(&vp, c->def_vp, sizeof(VessP));
    return resolve_enqueue(c, vp, c->def_cnt, c->def_params, err, errlen, on_side);
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
    if (!c->two_d && (!c->spec_ok || !hv_rs(c))) return nl_fail(err, errlen, NL_ESTATE, "the device-resident chain needs the 3-D one-pass walk");
    if (!c->d_chain) {
        NL_HIP(hipMalloc(&c->d_chain, sizeof(ChainScale) * NL_CHAIN_MAX_SCALES));
        NL_HIP(hipHostMalloc(&c->h_chain, sizeof(ChainScale) * NL_CHAIN_MAX_SCALES, hipHostMallocDefault));
        NL_HIP(hipEventCreateWithFlags(&c->ev_chain, hipEventDisableTiming));
    }
    chain_init_kernel<<<16, 256, 0, c->stream>>>((ChainScale *)c->d_chain, n_scales);
    chain_init2_kernel<<<1, 64, 0, c->stream>>>((ChainScale *)c->d_chain, n_scales);
    NL_CHECK_LAUNCH();
    c->chain_n = n_scales; c->chain_k = 0; c->chain_copy_pending = 0;
    c->def_resolve = 0;
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
    // the resolve kernel of the previous scale, held back until now: beside this scale's threshold kernels, behind its cascade step
    if ((rc = resolve_deferred_launch(c, true, err, errlen))) return rc;
    ChainScale *cs = (ChainScale *)c->d_chain + c->chain_k;
    c->chain_par[c->chain_k][0] = division; c->chain_par[c->chain_k][1] = margin; c->chain_par[c->chain_k][2] = test_scale;
    ++c->chain_k;
    c->frob_max_abs = 1.0f; c->frob_max_finite = 0.0f;                    // the bracket round: max_abs := 1 (pipeline.py _fsq_bracket)
    if (c->two_d) {
        // Round 6: images (im_info.no_z; filtering.py:675-690, 732-741) through the same records -- two passes, no walk and no queue: the
        // raw round only measures the sample range the exact round's edges are derived from (its "bracket" decides nothing: the caller passes
        // a margin of 1), the statistics kernel stands where the walk stands, and the vesselness kernel reads gamma_sq / fsq_min / m_inf
        // from the record as the resolve kernel does.  ~20 host round trips of a 2048^2 frame gone.
        if (!chain_unfused_sampling()) {
            if ((rc = range_hist_pair_enqueue(c, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_gauss, (char *)&cs->h_raw, nullptr, nullptr, err, errlen))) return rc;
        } else {
            if ((rc = range_hist_enqueue_at(c, NL_FIELD_GAUSS, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_gauss, nullptr, err, errlen))) return rc;
            if ((rc = range_hist_enqueue_at(c, NL_FIELD_FROB, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_raw, nullptr, err, errlen))) return rc;
        }
        chain_thr1_kernel<<<2, 64, 0, c->stream>>>(cs, division, margin, test_scale);
        {
            ProfScope ps(c, "hessian_stats");
            hessian2d_stats_kernel<<<grid2d_rows(c->nx, c->ny), 256, 0, c->stream>>>(gauss_cur(c), geom(c), hessp(c), cs->stats);
        }
        chain_post_kernel<<<1, 64, 0, c->stream>>>(cs);
        NL_CHECK_LAUNCH();
        {
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
        }
        chain_thr2_kernel<<<1, 64, 0, c->stream>>>(cs, division);
        NL_CHECK_LAUNCH();
        VessP vp{};
        vp.alpha_sq = (float)alpha_sq; vp.beta_sq = (float)beta_sq; vp.use_thr = 1;
        vp.cnt_lo = (int)c->own_lo; vp.cnt_hi = (int)c->own_hi;
        vp.first = c->mask_slots_used == 0 ? 1 : 0;
        vp.nan_flag = (unsigned int *)((char *)c->d_small + NL_NAN_FLAG_OFF);
        ProfScope ps(c, "vesselness");
        const int wpr = (int)((c->nx + 63) / 64);
        const i64 slot_words = c->nzl * c->ny * wpr;
        const int k_scale = c->mask_slots_used++;
        vp.have_prev = k_scale > 0;
        unsigned long long *cm = (unsigned long long *)c->m[0] + (i64)(k_scale & 1) * slot_words;
        const unsigned long long *pm = (unsigned long long *)c->m[0] + (i64)((k_scale + 1) & 1) * slot_words;
        if (vp.first && !vmax_is_zero(c, 0, c->nzl)) NL_HIP(hipMemsetAsync(c->f[c->i_vmax], 0, (size_t)c->n * 4, c->stream));
        c->vmax_zero_hi = 0;
        c->spec_valid = 0;
        vesselness2d_kernel<<<grid2d_rows((i64)wpr * 64, c->ny), 256, 0, c->stream>>>(gauss_cur(c), c->f[c->i_vmax], cm, pm, wpr, geom(c), hessp(c), vp,
                                                                                        &cs->cnt_resolve, (const float *)cs);
        NL_CHECK_LAUNCH();
        return NL_OK;
    }
    if (!chain_unfused_sampling()) {
        if ((rc = range_hist_pair_enqueue(c, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_gauss, (char *)&cs->h_raw, nullptr, nullptr, err, errlen))) return rc;
    } else {
        if ((rc = range_hist_enqueue_at(c, NL_FIELD_GAUSS, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_gauss, nullptr, err, errlen))) return rc;
        if ((rc = range_hist_enqueue_at(c, NL_FIELD_FROB, sz, sy, sx, NL_CHAIN_BINS, (char *)&cs->h_raw, nullptr, err, errlen))) return rc;
    }
    chain_thr1_kernel<<<2, 64, 0, c->stream>>>(cs, division, margin, test_scale);
    NL_CHECK_LAUNCH();
    if ((rc = spec_enqueue(c, spacing, 0.0f, 0.0f, z0, z1, cs->stats, &cs->cnt_walk, &cs->fsq_lo, err, errlen))) return rc;
    // (With the resolve kernel held back, the exact round of the scale on the side stream as well was measured: no gain, docs/HISTORY.md.)
    const bool defer = resolve_defer_ok(c) && c->fsq_cache_valid;     // (the cache is this scale's: use_fsq_cache below launches nothing)
    hipStream_t st = c->stream;
    chain_post_kernel<<<1, 64, 0, st>>>(cs);
    NL_CHECK_LAUNCH();
    {   // the exact round: edges from the normalised range, histogram of the cached frob_sq under the device's normalisation
        Lattice L; FieldSrc fs;
        if ((rc = make_lattice(c, sz, sy, sx, L, err, errlen))) return rc;
        if ((rc = make_field(c, NL_FIELD_FROB, fs, err, errlen))) return rc;
        if ((rc = use_fsq_cache(c, fs, L, err, errlen))) return rc;
        fs.norm_dev = cs->norm;
        const i64 total = L.cz * L.cy * L.cx;
        ProfScope ps(c, "sample", st);
        sample_edges_kernel<<<1, 64, 0, st>>>(cs->h_exact.res, NL_CHAIN_BINS, cs->h_exact.edges, cs->h_exact.res + 4);
        const size_t sh = (size_t)(NL_CHAIN_BINS + 2) * 4 + (size_t)NL_CHAIN_BINS * 4;
        if (total > 0) sample_hist_kernel<<<grid1d(total, 256, sample_grid_cap()), 256, sh, st>>>(fs, geom(c), L, cs->h_exact.edges, NL_CHAIN_BINS, cs->h_exact.counts, cs->h_exact.res + 4);
        NL_CHECK_LAUNCH();
        if (fused(c) && (rc = reduce_u64_sum(c, cs->h_exact.counts, NL_CHAIN_BINS, err, errlen))) return rc;
    }
    chain_thr2_kernel<<<1, 64, 0, st>>>(cs, division);
    NL_CHECK_LAUNCH();
    VessP vp{};
    vp.alpha_sq = (float)alpha_sq; vp.beta_sq = (float)beta_sq; vp.use_thr = 1;
    vp.cnt_lo = (int)c->own_lo; vp.cnt_hi = (int)c->own_hi;
    vp.first = c->mask_slots_used == 0 ? 1 : 0;
    if (defer) {
        memcpy(c->def_vp, &vp, sizeof(VessP));
        c->def_cnt = &cs->cnt_resolve; c->def_params = (const float *)cs; c->def_resolve = 1;
        return NL_OK;
    }
    return resolve_enqueue(c, vp, &cs->cnt_resolve, (const float *)cs, err, errlen);
}

// Optional, before nl_chain_finish: start the download of the records now, so that work enqueued after this call (the samples
// of the frame's percentile threshold) runs on the device while nl_chain_finish waits for the records only and repeats the
// decisions on the host.
extern "C" int nl_chain_flush(nl_ctx *c, char *err, size_t errlen) {
    NL_ENTER(c);
    if (!c->d_chain || c->chain_k < 1) return nl_fail(err, errlen, NL_ESTATE, "nl_chain_flush without scales");
    { int rcd = resolve_deferred_launch(c, false, err, errlen); if (rcd) return rcd; }       // the last scale's: nothing left to run beside
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
    if (!c->chain_copy_pending) { int rcd = resolve_deferred_launch(c, false, err, errlen); if (rcd) return rcd; }
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
        if (vp.first && !vmax_is_zero(c, 0, c->nzl)) NL_HIP(hipMemsetAsync(c->f[c->i_vmax], 0, (size_t)c->n * 4, c->stream));
        c->vmax_zero_hi = 0;
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
        if (vp.first && !vmax_is_zero(c, z0, z1))      // vesselness = zeros (filtering.py:807); only voxels alive in every mask are ever read again
            NL_HIP(hipMemsetAsync(c->f[c->i_vmax] + z0 * plane, 0, (size_t)(z1 - z0) * plane * 4, c->stream));
        c->vmax_zero_hi = 0;
        const int rs = hv_rs(c);
        const int ty = rs ? 2 * rs : hm_ty();
        const int nty = (int)((c->ny + ty - 1) / ty);
        if (rs) vp.qcap = 2 * hv_np(c) * HM_REGION;
#define NL_LAUNCH_VESS(TYV, FASTV, HR)                                                                                    \
        hessian_g_kernel<1, TYV, FASTV><<<nblocks, HGCfg<TYV>::NT, HGCfg<TYV>::lds_bytes(), c->stream>>>(                 \
            gauss_cur(c), cm, pm, wpr, geom(c), HR, vp, vq, (int)za, (int)zb, ntx, nty, nullptr, d_cnt)
        for (i64 za = z0; za < z1; za += planes_per_launch) {
            const i64 zb = za + planes_per_launch < z1 ? za + planes_per_launch : z1;
            const int nzc = (int)((zb - za + HM_ZCHUNK - 1) / HM_ZCHUNK);
            const unsigned nblocks = (unsigned)(ntx * nty * nzc);
            if (rs) NL_HIP(nl_hv_launch(HvLaunch{1, rs, hv_np(c), hv_fastv(c), nblocks, c->stream, gauss_cur(c), cm, pm, wpr, geom(c), hessp(c), vp, vq, (int)za, (int)zb,
                                                 ntx, nty, nullptr, d_cnt, nullptr}));
            else if (ty == 8) { if (c->fast_div) NL_LAUNCH_VESS(8, true, hessdv_fast(c)); else NL_LAUNCH_VESS(8, false, hessdv_exact(c)); }
            else { if (c->fast_div) NL_LAUNCH_VESS(16, true, hessdv_fast(c)); else NL_LAUNCH_VESS(16, false, hessdv_exact(c)); }
            NL_CHECK_LAUNCH();
            const unsigned nregions = nblocks * (unsigned)(rs ? rs / hv_np(c) : ty);
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
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, 1);
    ProfScope ps(c, "log2d");
    const float *src = gauss_cur(c);
    float *t = c->d_2d[0], *A = c->d_2d[1], *B = c->d_2d[2], *lap = c->d_2d[3];
    // gaussian_laplace: second derivative along Y (then plain Gaussian along X), plus the one along X.  Both terms in one walk over
    // the image (the pair kernel of Markers' LoG: A = XY(gy2, gx0) + XY(gy0, gx2), the float32 sum the combine kernel forms) when
    // the radius allows; four one-axis passes otherwise.
    bool summed = false;
    if (gyx_tiled() && r >= 1 && r <= GM_MAX_R && r <= c->ny && r <= c->nx) {
        const GaussWS wya = gauss_ws_of(gy2), wxa = gauss_ws_of(gx0), wyb = gauss_ws_of(gy0), wxb = gauss_ws_of(gx2);
        const dim3 g2((unsigned)((c->nx + GYX_COLS - 1) / GYX_COLS), (unsigned)((c->ny + v.chunk - 1) / v.chunk), 1);
        (void)gl_yx_dual(c, false, r, src, A, v, 0, 1, wya, wxa, wyb, wxb, g2);
        summed = true;
    } else {
        gl_axis(1, false, c, grid, src, t, v, 0, 1, gy2);
        gl_axis(2, false, c, grid, t, A, v, 0, 1, gx0);
        gl_axis(1, false, c, grid, src, t, v, 0, 1, gy0);
        gl_axis(2, false, c, grid, t, B, v, 0, 1, gx2);
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
    // (1024 workgroups: both kernels end in one atomic per workgroup on one word -- 8192 of them were 80 of the apply kernel's 99 us at 2048^2)
    log2d_clip_max_kernel<<<grid1d(c->n, 256, 1024), 256, 0, c->stream>>>(c->d_2d[3], c->n, res);
    NL_CHECK_LAUNCH();
    NL_HIP(hipMemcpyAsync(c->h_small, res, 4, hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    float mx;
    memcpy(&mx, c->h_small, 4);
    const float denom = mx + 1e-12f;                      // float32 + weak python float
    log2d_apply_kernel<<<grid1d(c->n, 256, 1024), 256, 0, c->stream>>>(c->d_2d[3], denom, c->f[c->i_vmax], c->n, d_pos);
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
    if (c->mask_slots_used == 0 && !vmax_is_zero(c, z0, z1))       // every scale was skipped: vesselness was never written
        NL_HIP(hipMemsetAsync(c->f[c->i_vmax] + z0 * plane, 0, (size_t)(z1 - z0) * plane * 4, c->stream));
    c->vmax_zero_hi = 0;
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
    nl_launch_threshold_pack(grid1d((m1 - m0) * c->ny * 64, 256, (i64)1 << 22), c->stream,
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

int store_planes(nl_ctx *c, const void *dev_base, void *host, size_t elem, int64_t z0, int64_t z1, char *err, size_t errlen) {
    if (!host || z0 < 0 || z1 > c->nzl || z0 >= z1) return nl_fail(err, errlen, NL_EINVAL, "bad plane range [%lld,%lld)", (i64)z0, (i64)z1);
    const i64 plane = c->ny * c->nx;
    NL_HIP(hipMemcpyAsync(host, (const char *)dev_base + (size_t)z0 * plane * elem, (size_t)(z1 - z0) * plane * elem,
                          hipMemcpyDeviceToHost, c->stream));
    NL_HIP(hipStreamSynchronize(c->stream));
    return NL_OK;
}

extern "C" int nl_filter_store(nl_ctx *c, float *host, int64_t z0, int64_t z1, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_LABBITS(c);
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
        (void)hipGetLastError();        // (see comm_acquire)
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
    (void)hipGetLastError();        // RCCL checks the thread's last HIP error during init: it must not inherit one that was handled long ago
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

// ---- Label's threshold sampling (flat strided samples of the Frangi volume, labelling.py:426-433) ----
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


// ---------------------------------------------------------------------------------- debug -------
extern "C" int nl_ctx_info(nl_ctx *c, const char *key, double *value) {
    if (!c || !key || !value) return NL_EINVAL;
    if (!strcmp(key, "fast_div")) *value = c->fast_div2 ? 2 : c->fast_div;      // 2: two-instruction division proven, 1: three, 0: float64
    else if (!strcmp(key, "hessian_tile_rows")) *value = hv_rs(c) ? 2 * hv_rs(c) : hm_ty();
    else if (!strcmp(key, "hv_variants")) *value = NL_HV_VARIANTS;               // 1: a build with the rejected forms of the walk (nellie_hv.hip)
    else if (!strcmp(key, "vesselness_one_pass")) *value = c->spec_ok;
    else if (!strcmp(key, "chain_available")) *value = (c->two_d || (c->spec_ok && hv_rs(c))) ? 1 : 0;   // nl_chain_begin's own precondition
    else if (!strcmp(key, "last_fsq_min")) *value = c->last_fsq_min;
    else if (!strcmp(key, "last_spec_overflow")) *value = c->last_spec_overflow;
    else if (!strcmp(key, "last_label_sparse")) *value = c->last_label_sparse;
    else if (!strcmp(key, "device_bytes")) *value = (double)nl_ctx_bytes(c->nzl, c->ny, c->nx);
    else if (!strcmp(key, "gauss_yx_max_r")) *value = getenv("NELLIE_NO_FUSED_YX") ? 0 : GM_MAX_R;   // largest in-plane radius whose Y and X passes share a kernel
    else if (!strcmp(key, "queue_entries")) {
        // entries the last one-pass walk (nl_vesselness_spec / a chain scale) left in the eigen queue: the sum of the per-wave region counts
        // (waits for the stream; bench.py's fused-design byte model prices the walk's stores and the resolve kernel's reads with it)
        if (hipSetDevice(c->device) != hipSuccess) return NL_EHIP;
        if (c->side_pending && hipStreamSynchronize(c->side) != hipSuccess) return NL_EHIP;
        const unsigned nreg = c->spec_nregions;
        std::vector<unsigned int> h(nreg);
        if (nreg && (hipMemcpyAsync(h.data(), c->d_vq_count, (size_t)nreg * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                     hipStreamSynchronize(c->stream) != hipSuccess)) return NL_EHIP;
        double t = 0.0;
        for (unsigned k = 0; k < nreg; ++k) t += (double)h[k];
        *value = t;
    }
    else if (!strcmp(key, "nan_hessian")) {
        // Filter.run(mask=False) only: 1 if a Hessian with a NaN entry reached the eigen-solver since the frame began (waits for the stream)
        if (hipSetDevice(c->device) != hipSuccess) return NL_EHIP;
        unsigned int flag = 0;
        if (c->side_pending && hipStreamSynchronize(c->side) != hipSuccess) return NL_EHIP;
        if (hipMemcpyAsync(&flag, (char *)c->d_small + NL_NAN_FLAG_OFF, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) return NL_EHIP;
        *value = flag ? 1.0 : 0.0;
    }
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
