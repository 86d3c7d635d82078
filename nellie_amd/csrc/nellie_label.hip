// Translation unit of libnellie_hip.so (gfx950): Label (runs + union-find, Z-slab protocol), Network's dense steps, frame streaming and the
// packed outputs.  C-ABI in include/nellie_amd.h; shared host pieces in nl_host.h.
#include "nl_host.h"
#include "label_voxels.inc"
#include "label_runs.inc"
#include "pack_out.inc"
#include "network.inc"

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
    bool rank_independent_bands = false;   // Z slab: the row bands of the in-LDS union level are a function of ny alone (build_components)
    bool whole_frame_paint = false;        // nl_label_run on a whole local volume: paint_out holds the frame's labels (the pack may follow, number_and_paint)
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
    // segments per plane: at least ~1024 workgroups for the in-LDS level (a 136-plane slab would otherwise use half of the CUs).
    // On a Z slab the banding must NOT depend on the rank: the tables the ranks exchange hold one entry per segment component of a
    // shared plane, so both ranks that see a plane have to cut it into the same row bands -- and a rank's plane count differs from its
    // neighbour's (uneven splits, one or two ghost planes).  There the rule reads ny alone: at most 8 bands of >= 32 rows.
    static int seg_target = -1;                                       // NELLIE_UF_SEG_WGS=1: one workgroup per plane (round 3)
    if (seg_target < 0) { const char *e = getenv("NELLIE_UF_SEG_WGS"); seg_target = (e && atoi(e) > 0) ? atoi(e) : 1024; }
    int seg_shift = 5;
    if (g.rank_independent_bands) {
        const i64 want = seg_target == 1 ? 1 : 8;
        while (((g.ny + ((i64)1 << seg_shift) - 1) >> seg_shift) > want) ++seg_shift;
    } else {
        const i64 want = (seg_target + g.nz - 1) / g.nz;
        while (((g.ny + ((i64)1 << seg_shift) - 1) >> seg_shift) > want) ++seg_shift;
    }
    const int nseg = (int)((g.ny + ((i64)1 << seg_shift) - 1) >> seg_shift);
    // one "done in LDS" byte per (plane, band)
    const size_t seg_bytes = (size_t)g.nz * nseg;
    if (two_level && rs.proot && rs.link && seg_bytes > c->seg_done_cap) {
        if (c->d_seg_done) { NL_HIP(hipStreamSynchronize(c->stream)); NL_HIP(hipFree(c->d_seg_done)); c->d_seg_done = nullptr; c->seg_done_cap = 0; }
        const size_t want_bytes = seg_bytes < 8192 ? 8192 : seg_bytes;
        NL_HIP(hipMalloc((void **)&c->d_seg_done, want_bytes));
        c->seg_done_cap = want_bytes;
    }
    const bool lvl2 = two_level && rs.proot && rs.link;
    int *link = lvl2 ? rs.link : nullptr;                            // rl_emit_kernel fills the pair filter's slots with -1
    if (g.wpr <= 30)
        rl_emit_kernel<true><<<(unsigned)((g.nrows + 255) / 256), 256, (size_t)256 * (g.wpr + 1) * 8, c->stream>>>(bits, invert, row_off, rs.runs, rs.parent, g.nrows, g.wpr, (int)g.nx, link, cap, d_ovf);
    else
        rl_emit_kernel<false><<<(unsigned)((g.nrows + 255) / 256), 256, 0, c->stream>>>(bits, invert, row_off, rs.runs, rs.parent, g.nrows, g.wpr, (int)g.nx, link, cap, d_ovf);
    NL_CHECK_LAUNCH();
    const unsigned gr = run_blocks(rs, g.nrows);
    const RunN rn = rs.rn();
    if (lvl2) {
        uint8_t *seg_done = c->d_seg_done;
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

static int pack_enqueue_fwd(nl_ctx *c, int with_labels, const float *frangi, const int *labels, char *err, size_t errlen);
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
    // a streamed stack packs every frame right behind its labels: enqueue that now, under this wait (nl_outputs_pack_with_label)
    bool packed = false;
    if (c->pack_with_label && g.whole_frame_paint && c->own_lo == 0 && c->own_hi == c->nzl && c->frangi_ready) {
        const int rcp = pack_enqueue_fwd(c, 1, c->f[c->i_vmax], g.paint_out, err, errlen);
        if (rcp) return rcp;
        packed = true;
    }
    NL_HIP(hipStreamSynchronize(c->stream));
    if (any) total = *(unsigned long long *)c->h_small;
    if (overflow) *overflow = any && !rs.host_count && *(unsigned int *)((char *)c->h_small + 8) != 0;
    c->pack_pending = (packed && !(overflow && *overflow)) ? 1 : 0;
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
    g.whole_frame_paint = true;
    c->pack_pending = 0;
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
    c->labbits_epoch = c->epoch.load();           // m[1] = the mask the labels were painted from (labels > 0, bit for bit)
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
    g.rank_independent_bands = true;
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
    if (n > cap) { *n_comp = 0; return nl_fail(err, errlen, NL_EINVAL, "the output arrays hold %lld nodes, the tables name %lld (n_nodes says how many to allocate)", (long long)cap, (long long)n); }
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
    NL_KEEP_LABBITS(c);
    if (c->i_labels < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_label_store before nl_label_run");
    return store_planes(c, c->f[c->i_labels], host, 4, z0, z1, err, errlen);
}

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
    nl_launch_pack_labels(grid1d(nw * 64, 256, (i64)1 << 20), c->stream, (const int *)c->f[3], skel, (int)c->nx, nrows, wpr);
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

// ------------------------------------------------------------------------------ frame streaming ---
// 3-D+T stacks (BASELINE config 5): frame t+1 travels host -> HBM on a copy stream while frame t computes, and the
// outputs of frame t-1 travel back on a second copy stream.  Host buffers must be pinned (nl_pinned_alloc) for the
// copies to be asynchronous.
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
    // (a kernel that PULLS the frame from page-locked host memory instead of the SDMA copy was tried in round 6: 8 workgroups already slow the
    // other contexts' kernels more than the copy engine does -- profiles/r06_h2d_pull.txt)
    NL_HIP(hipMemcpyAsync(c->d_in_slot[slot], host_pinned, (size_t)c->n * es, hipMemcpyHostToDevice, c->copy_in));
    NL_HIP(hipEventRecord(c->ev_in[slot], c->copy_in));
    return NL_OK;
}

// host wait for that slot's upload (copy-thread call: touches the slot's event only)
extern "C" int nl_input_wait(nl_ctx *c, int slot, char *err, size_t errlen) {
    NL_ENTER_IO(c);
    if (slot < 0 || slot > 1 || !c->copy_in || !c->d_in_slot[slot]) return nl_fail(err, errlen, NL_ESTATE, "input slot %d was never loaded", slot);
    NL_HIP(hipEventSynchronize(c->ev_in[slot]));
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
    NL_KEEP_LABBITS(c);
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
// header, closing entries of the offset tables, and what the emit kernels need -- from the device's own counts (one thread)
__global__ void pk_header_kernel(PkHeader h, char *blob, const unsigned int *flag, PkDev *dev, unsigned long long cap, long long rows) {
    const unsigned long long nv = dev->n_values, nr = h.with_labels ? dev->n_runs : 0ull;
    const unsigned long long off_lr = (unsigned long long)h.off_fv + ((nv * 4ull + 15ull) & ~15ull);
    const unsigned long long total = off_lr + ((nr * 4ull + 15ull) & ~15ull);
    const bool ok = *flag == 0u && total <= cap && nv <= 0xffffffffull && nr <= 0xffffffffull;
    dev->n_runs = nr; dev->off_lr = off_lr; dev->total = ok ? total : 0ull;
    if (!ok) return;
    h.n_values = (long long)nv; h.n_runs = (long long)nr; h.off_lr = (long long)off_lr; h.total = (long long)total;
    *reinterpret_cast<PkHeader *>(blob) = h;
    reinterpret_cast<unsigned int *>(blob + h.off_fo)[rows] = (unsigned int)nv;
    if (h.with_labels) reinterpret_cast<unsigned int *>(blob + h.off_lo)[rows] = (unsigned int)nr;
}

#define NL_PK_DEV_OFF (56 << 10)           // byte offset of the PkDev record in d_small / of its landing place in h_small
// Everything of nl_outputs_pack up to (not including) the wait: counts, bit planes, row offsets, header, items, and the D2H of the
// PkDev record.  Round 5: no host round trip between counting and emitting, so that nl_label_run can enqueue the pack of the frame
// under its OWN wait (nl_outputs_pack_with_label) -- a streamed config-5 frame loses two of its six host waits.
static int pack_enqueue(nl_ctx *c, int with_labels, const float *frangi, const int *labels, char *err, size_t errlen) {
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
        if (c->d_pack) { NL_HIP(hipStreamSynchronize(c->stream)); hipFree(c->d_pack); }
        c->d_pack = nullptr; c->pack_cap = 0;
        NL_HIP(hipMalloc(&c->d_pack, cap));
        c->pack_cap = cap;
    }
    char *blob = (char *)c->d_pack;
    NL_HIP(hipStreamWaitEvent(c->stream, c->ev_fetched, 0));     // the previous frame's blob has left the staging buffer
    unsigned int *cnt = c->d_rows;                                // per-row counts (Label's row tables are free once the labels are painted)
    unsigned int *flag = (unsigned int *)c->d_small + 40;
    unsigned long long *d_total = (unsigned long long *)c->d_small + 16;
    PkDev *dev = (PkDev *)((char *)c->d_small + NL_PK_DEV_OFF);
    NL_HIP(zero_small(flag, 4, c->stream));
    const unsigned grid = grid1d(rows * 64, 256, (i64)1 << 22);
    ProfScope ps(c, "pack");
    // counts, bit planes, row offsets
    pk_count_kernel<0><<<grid, 256, 0, c->stream>>>((const unsigned int *)frangi, (unsigned long long *)(blob + h.off_fb), cnt, rows, nx, wpr, flag);
    NL_CHECK_LAUNCH();
    if ((rc = scan_excl_u32(c, cnt, (unsigned int *)(blob + h.off_fo), rows, err, errlen))) return rc;
    NL_HIP(hipMemcpyAsync(&dev->n_values, d_total, 8, hipMemcpyDeviceToDevice, c->stream));
    if (with_labels) {
        pk_count_kernel<1><<<grid, 256, 0, c->stream>>>((const unsigned int *)labels, (unsigned long long *)(blob + h.off_lb), cnt, rows, nx, wpr, flag);
        NL_CHECK_LAUNCH();
        if ((rc = scan_excl_u32(c, cnt, (unsigned int *)(blob + h.off_lo), rows, err, errlen))) return rc;
        NL_HIP(hipMemcpyAsync(&dev->n_runs, d_total, 8, hipMemcpyDeviceToDevice, c->stream));
    }
    pk_header_kernel<<<1, 1, 0, c->stream>>>(h, blob, flag, dev, (unsigned long long)c->pack_cap, (long long)rows);
    NL_CHECK_LAUNCH();
    // the items (the kernels return at once when the frame does not pack)
    pk_emit_kernel<0><<<grid, 256, 0, c->stream>>>((const unsigned int *)frangi, (const unsigned long long *)(blob + h.off_fb),
                                                   (const unsigned int *)(blob + h.off_fo), blob, (long long)h.off_fv, dev, rows, nx, wpr);
    if (with_labels) pk_emit_kernel<1><<<grid, 256, 0, c->stream>>>((const unsigned int *)labels, (const unsigned long long *)(blob + h.off_lb),
                                                                    (const unsigned int *)(blob + h.off_lo), blob, (long long)h.off_fv, dev, rows, nx, wpr);
    NL_CHECK_LAUNCH();
    NL_HIP(hipEventRecord(c->ev_staged, c->stream));
    NL_HIP(hipMemcpyAsync((char *)c->h_small + NL_PK_DEV_OFF, dev, sizeof(PkDev), hipMemcpyDeviceToHost, c->stream));
    return NL_OK;
}

static int pack_enqueue_fwd(nl_ctx *c, int with_labels, const float *frangi, const int *labels, char *err, size_t errlen) {
    return pack_enqueue(c, with_labels, frangi, labels, err, errlen);
}

extern "C" int nl_outputs_pack(nl_ctx *c, int with_labels, int64_t *nbytes, char *err, size_t errlen) {
    NL_ENTER(c);
    NL_KEEP_LABBITS(c);
    NL_JOIN_SIDE(c);
    if (!nbytes) return nl_fail(err, errlen, NL_EINVAL, "nbytes is NULL");
    if (with_labels && c->i_labels < 0) return nl_fail(err, errlen, NL_ESTATE, "nl_outputs_pack(with_labels) before nl_label_run");
    if (c->own_lo != 0 || c->own_hi != c->nzl) return nl_fail(err, errlen, NL_EINVAL, "packed outputs are for whole local volumes (no ghost planes)");
    // nl_label_run of THIS frame already enqueued the pack and waited for it (nl_outputs_pack_with_label)
    const bool done = c->pack_pending && with_labels;
    c->pack_pending = 0;
    if (!done) {
        int rc = pack_enqueue(c, with_labels, c->f[c->i_vmax], with_labels ? (const int *)c->f[c->i_labels] : nullptr, err, errlen);
        if (rc) return rc;
        NL_HIP(hipStreamSynchronize(c->stream));
    }
    const PkDev *h = (const PkDev *)((const char *)c->h_small + NL_PK_DEV_OFF);
    *nbytes = (int64_t)h->total;
    return NL_OK;
}

// on != 0: nl_label_run ends by enqueueing nl_outputs_pack(with_labels = 1) of the frame under its own wait; the next
// nl_outputs_pack(ctx, 1, ...) then returns at once.  For streamed stacks (nellie_amd/streaming.py).
extern "C" int nl_outputs_pack_with_label(nl_ctx *c, int on, char *err, size_t errlen) {
    NL_ENTER(c);
    c->pack_with_label = on ? 1 : 0;
    c->pack_pending = 0;
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


extern "C" int nl_host_zero(void *dst, int64_t bytes, int threads, char *err, size_t errlen) {
    if (!dst || bytes < 0) return nl_fail(err, errlen, NL_EINVAL, "bad zero-fill request");
    const int nt = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    const size_t chunk = (((size_t)bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) {
        const size_t a = (size_t)t * chunk;
        if (a >= (size_t)bytes) break;
        const size_t b = a + chunk < (size_t)bytes ? a + chunk : (size_t)bytes;
        pool.emplace_back([=]() { memset((char *)dst + a, 0, b - a); });
    }
    for (auto &th : pool) th.join();
    return NL_OK;
}

// used by nl_mask_volume* in nellie_hip.hip (the kernel lives in this unit: label_runs.inc)
void nl_launch_threshold_pack(unsigned int grid, hipStream_t st, const float *f, const unsigned long long *support, unsigned long long *bits,
                              int has_thr, float thr, int nx, i64 nrows, int wpr, const float *thr_dev) {
    rl_threshold_pack_kernel<<<grid, 256, 0, st>>>(f, support, bits, has_thr, thr, nx, nrows, wpr, thr_dev);
}
