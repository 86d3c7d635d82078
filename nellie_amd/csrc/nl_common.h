// Internal declarations shared by the translation units of libnellie_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/nellie_amd.h"

typedef long long i64;

struct ProfRec { hipEvent_t a, b; };

struct nl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    i64 nzl = 0, ny = 0, nx = 0;      // local slab shape
    i64 gz0 = 0, gnz = 0;             // global placement
    i64 own_lo = 0, own_hi = 0;       // owned planes (local coordinates)
    i64 n = 0;                        // nzl*ny*nx

    float *f[4] = {nullptr, nullptr, nullptr, nullptr};   // four float32 volumes
    uint8_t *m[3] = {nullptr, nullptr, nullptr};          // three byte volumes
    int i_gauss = 0;        // f[] index holding the current Gaussian volume
    int i_vmax = 3;         // f[] index holding vesselness / the Filter output
    int i_labels = -1;      // f[] index holding the int32 labels after nl_label_run
    // m[0] = cumulative mask during Filter

    void *d_input = nullptr;   // optional resident raw frame (nl_input_load)
    int input_dtype = -1;
    int input_borrowed = 0;           // d_input points at a streaming slot (not owned)
    // frame streaming (nl_input_load_async & co)
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    void *d_in_slot[2] = {nullptr, nullptr};
    size_t in_bytes[2] = {0, 0};
    int in_dtype[2] = {-1, -1};
    hipEvent_t ev_in[2] = {nullptr, nullptr};
    hipEvent_t ev_staged = nullptr, ev_fetched = nullptr;
    float *d_stage_fr = nullptr;
    void *d_pack = nullptr; size_t pack_cap = 0;       // packed outputs (pack_out.inc): header | bit planes | row offsets | items
    int pack_with_label = 0;                           // nl_outputs_pack_with_label: nl_label_run enqueues the frame's pack under its own wait
    int pack_pending = 0;                              // ... and has done so for the labels now on the device (nl_outputs_pack returns at once)
    int *d_stage_lab = nullptr;
    void *d_small = nullptr;   // scratch for reductions / histograms / weights (64 KiB)
    void *h_small = nullptr;   // pinned mirror
    void *h_prefix = nullptr;  // pinned: the count and the first NL_PREFIX samples of a positive gather travel in one transfer
    float *d_vq = nullptr;     // global queue of voxels to eigen-solve (32-byte entries, one region per wave)
    unsigned int *d_vq_count = nullptr;   // entries written per region
    int nw_state = 0;             // Network: 1 after nl_skel_pixel_class (classes + branch bits resident)
    const float *mk_int = nullptr;   // Markers: the float32 intensities when they are read in place (a resident float32 input), else they live in d_vq
    float *mk_use = nullptr;      // Markers: LoG source when use_im = 'frangi' (inside d_vq), else the distance image
    float *mk_scratch = nullptr;  // Markers: volume between the Y and the X pass of the any-radius LoG path, allocated on first use
    std::atomic<unsigned long long> epoch{0};      // C-ABI calls made on this context (see NL_KEEP_SUPPORT)
    unsigned long long support_epoch = ~0ull - 8;   // epoch at which d_support described the Frangi frame
    const unsigned long long *d_support = nullptr;
    unsigned long long labbits_epoch = ~0ull - 8;   // epoch at which m[1] held the bits `labels > 0` of the label volume (nl_label_run's last
                                                    // mask: what it painted from); Markers then skips re-deriving them from 4 B/voxel of labels
    int last_label_sparse = 0;
    unsigned char *mk_act = nullptr; size_t mk_act_cap = 0;   // Markers, sparse LoG (markers.inc): [tile list (int32) | tile bytes | Z-march map]
    int mk_act_valid = 0;         // the tile list describes the current mask ...
    int mk_ntiles = 0;            // ... and holds this many active 16 x 64 tiles
    int mk_state = 0;             // Markers: 0 idle, 1 begun, 2 distance done, 3 finished
    int mk_first_scale = 1;
    int two_d = 0;                // the frame is a (Y, X) image (im_info.no_z): 2-D Hessian, eigenvalues, Frangi, opening
    float *d_2d[4] = {nullptr, nullptr, nullptr, nullptr};   // 2-D only: LoG scratch (intermediate, two terms, running maximum)
    float *d_fsq_cache = nullptr;     // frob_sq at the lattice points of the current scale (NL_FIELD_FROB is sampled up to 4x)
    i64 fsq_cache_cap = 0, fsq_cache_key[3] = {0, 0, 0};
    int fsq_cache_valid = 0;
    float *gauss_ext = nullptr;   // current Gaussian volume when it is NOT one of f[0..2]: the resident float32 input itself,
                                  // until the first cascade step has written a volume of its own (saves the 8 B/voxel copy)
    int vq_chunks = 1;         // Z chunks (HM_ZCHUNK planes) one vesselness launch may cover
    int spec_ok = 0;           // the queue can hold a one-pass (MODE 2) vesselness of the whole slab
    int spec_valid = 0;        // a MODE 2 pass is waiting for nl_vesselness_resolve
    float spec_lo = 0, spec_hi = 0;
    i64 spec_z0 = 0, spec_z1 = 0;
    unsigned int spec_nregions = 0;
    int spec_qcap = 0;                   // entries per queue region of that pass
    unsigned long long spec_count = 0;   // owned voxels the pass already counted as h_mask
    hipStream_t side = nullptr;          // the resolve kernel of scale s runs here, beside the Gaussian of scale s+1
    hipEvent_t ev_side = nullptr, ev_main = nullptr;
    hipEvent_t ev_ahead = nullptr;       // a cascade step enqueued ahead on `side` (nl_gauss_step_ahead)
    int ahead_pending = 0, ahead_gauss = 0;
    int side_pending = 0;                // work on `side` the main stream has not been ordered after yet
    // device chain, round 5: the resolve kernel of scale s is held back until the cascade step of scale s+1 has been enqueued and then runs
    // on `side` beside that scale's threshold kernels (nl_chain_scale); def_* = what resolve_enqueue needs then
    int def_resolve = 0;
    alignas(8) unsigned char def_vp[160] = {};       // a VessP (hessian.inc), copied in and out
    unsigned long long *def_cnt = nullptr;
    const float *def_params = nullptr;
    int last_spec_overflow = 0;          // the last one-pass walk overflowed a queue region (diagnostics)
    float last_fsq_min = 0;    // the exact mask threshold of the last scale (diagnostics)
    void *d_blk = nullptr;     // per-block partials for scans
    unsigned int *d_rows = nullptr;   // per-row run counts and offsets (Label on runs)
    unsigned long long *gbits[2] = {nullptr, nullptr};   // Z-slab Label: GLOBAL bit masks (lazily allocated)
    unsigned int *grows = nullptr;                       // ... and the global per-row arrays
    int comm_poisoned = 0;     // an RCCL call of this context failed: its communicators are destroyed, not pooled
    i64 blk_cap = 0;
    // Z-slab Label without replication (nl_slab_*): planes [sl_e0, sl_e1) = owned planes + one ghost plane per interior side
    i64 sl_e0 = 0, sl_e1 = 0;
    i64 sl_nruns = 0;
    int sl_phase = -1;
    int *d_sl = nullptr; size_t d_sl_ints = 0;     // tables of the slab protocol: header, entry indices, this rank's blob, the gathered blobs
    int *h_sl = nullptr; size_t h_sl_ints = 0;     // their page-locked landing area (also stages the host's patch lists)
    int sl_capE = 0;                               // entries per plane the tables hold
    int sl_numbered = 0;
    uint8_t *d_seg_done = nullptr; size_t seg_done_cap = 0;   // Label's two-level union-find: one "joined in LDS" byte per (plane, row band)

    float hz = 1, hy = 1, hx = 1;            // float32(h)
    float hz2 = 2, hy2 = 2, hx2 = 2;         // float32(2.0*h)
    int have_spacing = 0;
    int fast_div = 0;                        // 3-instruction division proven exact for the spacings in use
    int fast_div2 = 0;                       // 2-instruction division (hessian.inc: Dv<2>) proven exact for them (pair kernel)
    double chk_spacing[3] = {0, 0, 0};
    float frob_max_abs = 1.0f, frob_max_finite = 0.0f;
    int frangi_ready = 0;
    i64 vmax_zero_lo = 0, vmax_zero_hi = 0;   // planes of the vesselness volume a cascade step of this frame has zeroed in passing and nothing
                               // has written since (gauss_zyx.inc: zero_out); the first evaluated scale then skips its own fill
    int mask_slots_used = 0;   // per-scale h_mask bit planes written since the frame began
    hipEvent_t ev_chain = nullptr; int chain_copy_pending = 0;
    i64 gp_total = -1; float *gp_stage = nullptr;   // nl_sample_gather_positive_begin .. _end
    // nl_tail_enqueue .. nl_tail_finish: percentile threshold + _mask_volume decided on the device (percentile.inc)
    void *d_pct = nullptr;      // PctRec, sample counter, voxel counter, two histograms
    void *h_pct = nullptr;      // pinned landing area of the record and the count
    int tail_pending = 0, tail_dst = -1;
    void *d_chain = nullptr, *h_chain = nullptr;     // device-resident threshold chain (chain.inc): records of a frame's scales, pinned mirror
    int chain_n = 0, chain_k = 0;
    double chain_par[16][3] = {};                    // (division, margin, test scale) each scale was enqueued with

    void *comm = nullptr;             // ncclComm_t (RCCL), set by nl_comm_init
    int fuse_reduce = 0;              // nl_comm_fuse: the sampling / statistics entry points reduce across the ranks on the device
    void *d_ag = nullptr; size_t ag_cap = 0;      // staging of nl_allgather_bytes
    void *h_ag = nullptr; size_t h_ag_cap = 0;    // page-locked landing area of nl_allgather_var
    void *comm2 = nullptr;            // second communicator: asynchronous ghost-plane exchanges (nl_comm_init2)
    hipStream_t xstream = nullptr;    // ... and their stream
    hipEvent_t ev_x_main = nullptr, ev_x_done = nullptr;
    int halo_pending = 0;             // an asynchronous exchange the next nl_gauss_step has to wait for
    int world = 1, rank = 0;

    hipEvent_t t0 = nullptr, t1 = nullptr;
    int prof_on = 0;
    std::map<std::string, std::vector<ProfRec>> prof;
    std::map<std::string, std::pair<double, int64_t>> prof_sum;      // completed scopes: total ms, count
    std::vector<ProfRec> prof_pool;      // event pairs ready for use: creating events inside a timed region stalls now and then
                                         // (one hipEventCreate was measured at 57 ms when the runtime grew its pool)
    std::vector<hipEvent_t> ev_pool;
};

static inline int nl_fail(char *err, size_t errlen, int code, const char *fmt, ...) {
    if (err && errlen) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, errlen, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define NL_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipGetLastError();   /* reported through our own channel: a sticky "last error" would fail the next library */ \
                                       /* that checks it -- RCCL's communicator init did, 280 tests later (round 5) */            \
            int code_ = (e_ == hipErrorOutOfMemory) ? NL_ENOMEM                               \
                        : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? NL_ENODEV \
                                                                                  : NL_EHIP;  \
            return nl_fail(err, errlen, code_, "%s: %s (%s:%d)%s", #expr, hipGetErrorString(e_), \
                           __FILE__, __LINE__, code_ == NL_ENOMEM ? " [out of memory]" : ""); \
        }                                                                                     \
    } while (0)

#define NL_CHECK_LAUNCH() NL_HIP(hipGetLastError())

// Completed event pairs are turned into per-group sums and go back to the pool (all = false: only from the front of each
// group, as far as the events have completed -- no waiting).
static inline void prof_harvest(nl_ctx *c, bool all) {
    for (auto &kv : c->prof) {
        auto &v = kv.second;
        size_t k = 0;
        for (; k < v.size(); ++k) {
            if (!all && hipEventQuery(v[k].b) != hipSuccess) { (void)hipGetLastError(); break; }    // "not ready" is an answer, not an error to leave behind
            float t = 0;
            if (hipEventElapsedTime(&t, v[k].a, v[k].b) == hipSuccess) { auto &s = c->prof_sum[kv.first]; s.first += t; s.second += 1; }
            c->prof_pool.push_back(v[k]);
        }
        v.erase(v.begin(), v.begin() + k);
    }
}

// RAII profiling scope: records a HIP-event pair on the context stream around a kernel group.
struct ProfScope {
    nl_ctx *c; ProfRec r; bool on;
    hipStream_t st;
    ProfScope(nl_ctx *ctx, const char *name, hipStream_t stream = nullptr) : c(ctx), on(ctx->prof_on != 0), st(stream ? stream : ctx->stream) {
        if (!on) return;
        if (c->prof_pool.empty()) prof_harvest(c, false);
        if (!c->prof_pool.empty()) { r = c->prof_pool.back(); c->prof_pool.pop_back(); }
        else { hipEventCreate(&r.a); hipEventCreate(&r.b); }
        hipEventRecord(r.a, st);
        key = name;
    }
    // the record is filed only now, complete: a harvest in between never sees a pair whose second event is unrecorded
    ~ProfScope() { if (on) { hipEventRecord(r.b, st); c->prof[key].push_back(r); } }
    std::string key;
};

