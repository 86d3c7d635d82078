// Translation unit of libnellie_hip.so (gfx950): Markers (distance transform, multiscale LoG, peaks).  C-ABI in include/nellie_amd.h;
// shared host pieces in nl_host.h.
#include "nl_host.h"
#include "markers.inc"

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
    // device-resident labels straight from nl_label_run: m[1] still holds the mask they were painted from -- the bits wanted here
    const bool have_bits = !labels_host && c->labbits_epoch + 1 == c->epoch.load() && !getenv("NELLIE_MK_REPACK");
    if (!have_bits) {
        mk_pack_labels_kernel<<<grid1d(nrows * wpr * 64, 256, (i64)1 << 20), 256, 0, c->stream>>>(lab, (unsigned long long *)c->m[1], (int)c->nx, nrows, wpr);
        NL_CHECK_LAUNCH();
    }
    // intensities as float32 (score_img[...] = intensity_im[...], mocap_marking.py:595-596)
    c->mk_int = nullptr;
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
            case NL_F32: c->mk_int = (const float *)c->d_input; break;        // read where it lies (nothing writes the intensities)
            case NL_F64: convert_kernel<double><<<g, 256, 0, c->stream>>>((const double *)c->d_input, mk_intensity(c), c->n); break;
            case NL_U64: convert_kernel<uint64_t><<<g, 256, 0, c->stream>>>((const uint64_t *)c->d_input, mk_intensity(c), c->n); break;
            case NL_I64: convert_kernel<int64_t><<<g, 256, 0, c->stream>>>((const int64_t *)c->d_input, mk_intensity(c), c->n); break;
        }
        NL_CHECK_LAUNCH();
    }
    c->i_labels = -1; c->frangi_ready = 0; c->gauss_ext = nullptr; c->fsq_cache_valid = 0;
    c->mk_state = 1; c->mk_first_scale = 1; c->mk_use = nullptr; c->mk_act_valid = 0;
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
    // best response = 0, no peaks yet (mocap_marking.py:483-484): the first scale's peak kernel defines `best` where it is read
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
    const dim3 grid((unsigned)((c->nx + 255) / 256), (unsigned)c->ny, (unsigned)c->nzl);
    const dim3 g2((unsigned)((c->nx + GYX_COLS - 1) / GYX_COLS), (unsigned)((c->ny + v.chunk - 1) / v.chunk), (unsigned)c->nzl);
    float *dist = c->f[0], *tz = c->f[1], *lap = c->f[2];
    const float *use = c->mk_use ? c->mk_use : dist;            // the image the LoG runs on
    // Sparse LoG (round 5; markers.inc): the in-plane passes run on the 16 x 64 tiles that hold a voxel within one voxel of the mask
    // (one workgroup per listed tile, log_yx_sparse_kernel), the Z march on the 64 x 4 x 64 tiles those stage.  NELLIE_MK_SPARSE=0:
    // dense.  NELLIE_MK_POISON=1 (tests): both scratch volumes are filled with NaNs first -- a value read from a skipped tile would show.
    const char *e_sp = getenv("NELLIE_MK_SPARSE");
    const bool sparse = !(e_sp && !atoi(e_sp)) && !generic && !flat && gyx_tiled() && rz <= GM_MAX_R && rz <= c->gnz && ryx <= c->nx;
    VolGeom vz = v;
    const unsigned long long *need_z = nullptr;
    const int *tile_list = nullptr;
    if (sparse) {
        vz.chunk = 64;
        const int ntx = (int)((c->nx + LS_TX - 1) / LS_TX), nty = (int)((c->ny + LS_TY - 1) / LS_TY);
        const int nx64 = (int)((c->nx + 63) / 64), ny4 = (int)((c->ny + 3) / 4), nzc = (int)((c->nzl + vz.chunk - 1) / vz.chunk);
        const size_t n_t = (size_t)c->nzl * nty * ntx, n_z = (size_t)nzc * ny4 * nx64;
        const size_t list_bytes = ((n_t * 4 + 7) & ~(size_t)7);
        const size_t need = 64 + list_bytes + n_z * 8 + n_t;
        if (need > c->mk_act_cap) {
            if (c->mk_act) { NL_HIP(hipStreamSynchronize(c->stream)); NL_HIP(hipFree(c->mk_act)); c->mk_act = nullptr; }
            NL_HIP(hipMalloc((void **)&c->mk_act, need));
            c->mk_act_cap = need; c->mk_act_valid = 0;
        }
        unsigned int *d_count = (unsigned int *)c->mk_act;             // [count | list | tile bytes | Z-march map]
        int *list = (int *)(c->mk_act + 64);
        unsigned long long *zbits = (unsigned long long *)(c->mk_act + 64 + list_bytes);    // [count | list | Z-march plane bits | tile bytes]
        unsigned char *tile_act = (unsigned char *)(zbits + n_z);
        if (!c->mk_scratch) NL_HIP(hipMalloc((void **)&c->mk_scratch, (size_t)c->n * 4));       // the second Z-filtered volume (both Z terms in one walk)
        const int wpr = (int)((c->nx + 63) / 64);
        if (!c->mk_act_valid) {                         // the mask is the same for every sigma of a frame: one list, one wait
            NL_HIP(hipMemsetAsync(tile_act, 0, n_t, c->stream));
            NL_HIP(zero_small(d_count, 4, c->stream));
            mk_tile_mark_kernel<<<grid1d(c->nzl * c->ny * wpr, 256, 1 << 14), 256, 0, c->stream>>>((const unsigned long long *)c->m[1], tile_act, v, wpr, ntx, nty);
            mk_tile_list_kernel<<<grid1d((i64)n_t, 256, 1 << 12), 256, 0, c->stream>>>(tile_act, (i64)n_t, list, d_count);
            NL_CHECK_LAUNCH();
            NL_HIP(hipMemcpyAsync(c->h_small, d_count, 4, hipMemcpyDeviceToHost, c->stream));
            NL_HIP(hipStreamSynchronize(c->stream));
            c->mk_ntiles = (int)*(unsigned int *)c->h_small;
            c->mk_act_valid = 1;
        }
        NL_HIP(hipMemsetAsync(zbits, 0, n_z * 8, c->stream));
        if (c->mk_ntiles > 0)
            mk_z_active_kernel<<<grid1d(c->mk_ntiles, 256, 1 << 12), 256, 0, c->stream>>>(list, d_count, zbits, v, ntx, nty, ryx, vz.chunk, nx64, ny4);
        NL_CHECK_LAUNCH();
        need_z = zbits; tile_list = list;
        if (getenv("NELLIE_MK_DEBUG")) {              // occupancy of the maps (diagnostics: waits for the stream)
            std::vector<unsigned long long> h(n_z);
            NL_HIP(hipMemcpyAsync(h.data(), zbits, n_z * 8, hipMemcpyDeviceToHost, c->stream));
            NL_HIP(hipStreamSynchronize(c->stream));
            size_t a = 0, b = 0;
            for (size_t k = 0; k < n_z; ++k) { a += h[k] != 0; b += (size_t)__builtin_popcountll(h[k]); }
            fprintf(stderr, "[markers] r=%d: %d of %zu in-plane tiles listed (%.1f %%), Z-march workgroups %.1f %%, (plane, column tile) pairs %.1f %%\n", ryx,
                    c->mk_ntiles, n_t, 100.0 * c->mk_ntiles / n_t, 100.0 * a / n_z, 100.0 * b / ((double)c->nzl * ny4 * nx64));
        }
        const char *e_po = getenv("NELLIE_MK_POISON");
        if (e_po && atoi(e_po)) {
            NL_HIP(hipMemsetAsync(tz, 0xff, (size_t)c->n * 4, c->stream));
            NL_HIP(hipMemsetAsync(lap, 0xff, (size_t)c->n * 4, c->stream));
            NL_HIP(hipMemsetAsync(c->mk_scratch, 0xff, (size_t)c->n * 4, c->stream));
        }
    }
    auto zpass = [&](const GaussW &gz) {
        if (!gl_fast(0, c, use, tz, v, z0, z1, gz)) gl_axis(0, false, c, grid, use, tz, v, z0, z1, gz);
    };
    // large radii: the fused Y+X kernel turns compute-bound (one output per thread reads 2R+1 LDS values); a marching Y
    // pass plus the stand-alone X kernel (four outputs per thread) through one more scratch volume is faster there
    static int split_from = -1;
    if (split_from < 0) { const char *e = getenv("NELLIE_MK_SPLIT_R"); split_from = e ? atoi(e) : (gyx_tiled() ? 99 : 8); }
    float *tmp2 = mk_intensity(c) + c->n;
    const bool can_split = !c->mk_use && vq_alloc_entries(c->nzl, c->ny, c->nx) * 32 >= c->n * 8;     // mk_use lives in tmp2's place
    const float *yx_src = flat ? use : tz;                     // (captured by reference: the sparse path switches it between its two walks)
    auto yx = [&](const GaussW &gy, const GaussW &gx, bool acc) {
        const GaussWS wy = gauss_ws_of(gy), wx = gauss_ws_of(gx);
        if (can_split && ryx >= split_from) { (void)gl_y_then_x(c, acc, ryx, yx_src, tmp2, lap, v, z0, z1, wy, wx); return; }
        if (sparse) { (void)gl_log_yx_sparse(c, false, acc, ryx, yx_src, lap, v, tile_list, c->mk_ntiles, wy, wx, wy, wx); return; }
        (void)gl_yx(c, gyx_tiled(), acc, ryx, yx_src, lap, v, z0, z1, wy, wx, g2);
    };
    // the two in-plane terms in one walk over their common input
    static int dual = -1;
    if (dual < 0) { const char *e = getenv("NELLIE_MK_DUAL"); dual = (e && !atoi(e)) ? 0 : 1; }
    auto yx_dual = [&](bool acc) {
        if (sparse) {
            (void)gl_log_yx_sparse(c, true, acc, ryx, yx_src, lap, v, tile_list, c->mk_ntiles, gauss_ws_of(gy2), gauss_ws_of(gx0), gauss_ws_of(gy0), gauss_ws_of(gx2));
            return;
        }
        (void)gl_yx_dual(c, acc, ryx, yx_src, lap, v, z0, z1, gauss_ws_of(gy2), gauss_ws_of(gx0), gauss_ws_of(gy0), gauss_ws_of(gx2), g2);
    };
    const bool use_dual = dual && gyx_tiled();
    auto yx_generic = [&](const GaussW &gy, const GaussW &gx, bool acc) {
        gl_axis(1, false, c, grid, yx_src, c->mk_scratch, v, z0, z1, gy);
        gl_axis(2, acc, c, grid, c->mk_scratch, lap, v, z0, z1, gx);
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
        } else if (sparse && use_dual) {
            // both Z terms in one walk over the image (tz = d2/dz2 term's input, mk_scratch = the plain-Gaussian one), then the listed tiles
            (void)gl_z_dual(c, rz, use, tz, c->mk_scratch, vz, z0, z1, gauss_ws_of(gz2), gauss_ws_of(gz0), need_z);
            yx(gy0, gx0, false);
            yx_src = c->mk_scratch;
            yx_dual(true);
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
        // groups of four rows share their neighbour rows (NELLIE_MK_PEAK4=0: one word at a time, the round-3 kernel)
        const char *e4 = getenv("NELLIE_MK_PEAK4");
        if (e4 && !atoi(e4))
            mk_peak_kernel<<<grid1d(nw, 256, (i64)1 << 20), 256, 0, c->stream>>>(lap, s2, (const unsigned long long *)c->m[1], dist, c->f[3],
                                                                                     (unsigned long long *)c->m[0], v, wpr, c->mk_first_scale);
        else
            mk_peak4_kernel<<<grid1d(c->nzl * ((c->ny + 3) / 4) * wpr, 256, (i64)1 << 20), 256, 0, c->stream>>>(
                lap, s2, (const unsigned long long *)c->m[1], dist, c->f[3], (unsigned long long *)c->m[0], v, wpr, c->mk_first_scale);
        NL_CHECK_LAUNCH();
        c->mk_first_scale = 0;
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
        mk_nms_kernel<<<grid1d(nw, 256, 256 * 32), 256, 0, c->stream>>>((const unsigned long long *)c->m[0], c->mk_int ? c->mk_int : mk_intensity(c), peak_min_distance,
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


// used by nl_skel_* in nellie_label.hip (the kernel lives in this unit: markers.inc)
void nl_launch_pack_labels(unsigned int grid, hipStream_t st, const int *lab, unsigned long long *bits, int nx, i64 nrows, int wpr) {
    mk_pack_labels_kernel<<<grid, 256, 0, st>>>(lab, bits, nx, nrows, wpr);
}
