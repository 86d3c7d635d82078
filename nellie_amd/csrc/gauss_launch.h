// Internal: the Gaussian passes of libnellie_hip.so live in their own translation unit (nellie_gauss.hip: ~70 unrolled kernel
// instantiations, a minute of compile time); the cascade (nellie_hip.hip), the 2-D blob filter and the LoG of Markers (nellie_markers.hip)
// reach them through these launchers.  Every launcher enqueues on c->stream and returns false when the radius has no specialised kernel.
#pragma once
#define NL_MAX_RADIUS 63
struct GaussW { double w[NL_MAX_RADIUS + 1]; int r; };   // w[k] = weight at distance k from the centre
#define GM_MAX_R 12       // Filter's cascade needs <= 5 at 0.1 um; the LoG kernels of Markers (truncate 4.0) reach 11
struct GaussWS { double w[GM_MAX_R + 1]; };
#define GYX_COLS 256      // fused Y+X pass: output columns per workgroup
#define GYX_THREADS 320
#define GYX_HALO 32       // staged columns left of the first output: (GYX_THREADS - GYX_COLS) / 2, >= GM_MAX_R
#define GX_SEG 1024       // stand-alone X pass: row segment per workgroup

static int fill_gw(GaussW &gw, const double *w, int r, char *err, size_t errlen) {
    if (r < 0 || r > NL_MAX_RADIUS) return nl_fail(err, errlen, NL_EINVAL, "Gaussian radius %d outside [0,%d]", r, NL_MAX_RADIUS);
    gw.r = r;
    for (int k = 0; k <= r; ++k) gw.w[k] = w[r + k];   // w[] has 2r+1 entries centred at r (symmetric)
    return NL_OK;
}

static inline GaussWS gauss_ws_of(const GaussW &g) { GaussWS w; for (int k = 0; k <= GM_MAX_R; ++k) w.w[k] = k <= g.r ? g.w[k] : 0.0; return w; }

// marching Z / Y pass or the LDS X pass of radius gw.r <= GM_MAX_R (axis 0, 1, 2)
bool gl_fast(int axis, nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw);
// one thread per voxel, any radius, scipy's multiple reflection; acc: dst += result
void gl_axis(int axis, bool acc, nl_ctx *c, dim3 grid, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw);
// fused Y+X pass (tiled: the register-blocked X pass; else one row at a time), radius r on both axes; g2 = (x tiles, y chunks, planes)
bool gl_yx(nl_ctx *c, bool tiled, bool acc, int r, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wy,
           const GaussWS &wx, dim3 g2);
// dst (+)= XY(wya, wxa)(src) + XY(wyb, wxb)(src) in one walk (the two in-plane terms of generic_laplace)
bool gl_yx_dual(nl_ctx *c, bool acc, int r, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wya,
                const GaussWS &wxa, const GaussWS &wyb, const GaussWS &wxb, dim3 g2);
// marching Y pass into tmp, then the stand-alone X pass into dst (acc: dst += result): large radii of Markers' LoG
bool gl_y_then_x(nl_ctx *c, bool acc, int r, const float *src, float *tmp, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wy,
                 const GaussWS &wx);
// sparse in-plane passes of the LoG (log_yx_sparse_kernel): the listed 16 x 64 tiles only; dual: both terms (wya, wxa) + (wyb, wxb)
#define LS_TY 16
#define LS_TX 64
bool gl_log_yx_sparse(nl_ctx *c, bool dual, bool acc, int r, const float *src, float *dst, const VolGeom &v, const int *list, int ntiles,
                      const GaussWS &wya, const GaussWS &wxa, const GaussWS &wyb, const GaussWS &wxb);
// Z march for both Z terms of a LoG scale in one walk (gauss_march_dual_kernel): dst_a = Z(wa)(src), dst_b = Z(wb)(src) on the planes the map
// names -- need_bits: one 64-bit word per workgroup of the 64-column x 4-row x v.chunk-plane grid (v.chunk <= 64), bit k = plane k of the
// chunk; NULL = everything
bool gl_z_dual(nl_ctx *c, int r, const float *src, float *dst_a, float *dst_b, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wa, const GaussWS &wb,
               const unsigned long long *need_bits);
// the whole cascade step in one kernel (gauss_zyx.inc, nellie_gzyx.hip): Z radius rz with weights gz, Y and X radius r with the SAME
// weights gyx (sigma_vec = (s / z_ratio, s, s), filtering.py:816-825); false: no instantiation for these radii / this shape
bool gl_zyx_ok(const nl_ctx *c, int rz, int r, const float *dst);
// zero_out (may be NULL): a volume filled with zeros on planes [z0, z1) in passing (the running scale maximum of a frame's first step)
bool gl_zyx(nl_ctx *c, int rz, int r, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &gz, const GaussWS &gyx,
            float *zero_out);
