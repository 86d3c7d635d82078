// Translation unit of libnellie_hip.so (gfx950): the separable Gaussian passes (scipy correlate1d arithmetic, gauss.inc) and their launchers
// (gauss_launch.h).
#include "nl_host.h"
#include "gauss.inc"

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

bool gl_fast(int axis, nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
    if (axis == 0) return launch_gauss_fast<0>(c, src, dst, v, z0, z1, gw);
    if (axis == 1) return launch_gauss_fast<1>(c, src, dst, v, z0, z1, gw);
    return launch_gauss_fast<2>(c, src, dst, v, z0, z1, gw);
}

void gl_axis(int axis, bool acc, nl_ctx *c, dim3 grid, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussW &gw) {
    const dim3 blk(256, 1, 1);
    if (axis == 0) { if (acc) gauss_axis_kernel<0, true><<<grid, blk, 0, c->stream>>>(src, dst, v, z0, z1, gw); else gauss_axis_kernel<0><<<grid, blk, 0, c->stream>>>(src, dst, v, z0, z1, gw); }
    else if (axis == 1) { if (acc) gauss_axis_kernel<1, true><<<grid, blk, 0, c->stream>>>(src, dst, v, z0, z1, gw); else gauss_axis_kernel<1><<<grid, blk, 0, c->stream>>>(src, dst, v, z0, z1, gw); }
    else { if (acc) gauss_axis_kernel<2, true><<<grid, blk, 0, c->stream>>>(src, dst, v, z0, z1, gw); else gauss_axis_kernel<2><<<grid, blk, 0, c->stream>>>(src, dst, v, z0, z1, gw); }
}

#define NL_R12(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12)

bool gl_yx(nl_ctx *c, bool tiled, bool acc, int r, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wy,
           const GaussWS &wx, dim3 g2) {
    const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
    if (tiled) {
        switch (r) {
#define NL_YT(RR) case RR: if (acc) gauss_yx_tile_kernel<RR, true><<<g2.x * g2.y * g2.z, GYX_THREADS, 0, c->stream>>>(src, dst, v, z0, z1, wy, wx, vec4, (int)g2.x, (int)g2.y); \
                           else gauss_yx_tile_kernel<RR, false><<<g2.x * g2.y * g2.z, GYX_THREADS, 0, c->stream>>>(src, dst, v, z0, z1, wy, wx, vec4, (int)g2.x, (int)g2.y); return true;
            NL_R12(NL_YT)
#undef NL_YT
        }
        return false;
    }
    switch (r) {
#define NL_YX(RR) case RR: if (acc) gauss_yx_kernel<RR, true><<<g2, GYX_THREADS, 0, c->stream>>>(src, dst, v, z0, z1, wy, wx); \
                           else gauss_yx_kernel<RR, false><<<g2, GYX_THREADS, 0, c->stream>>>(src, dst, v, z0, z1, wy, wx); return true;
        NL_R12(NL_YX)
#undef NL_YX
    }
    return false;
}

bool gl_yx_dual(nl_ctx *c, bool acc, int r, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wya,
                const GaussWS &wxa, const GaussWS &wyb, const GaussWS &wxb, dim3 g2) {
    const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
    const unsigned nb = g2.x * g2.y * g2.z;
    switch (r) {
#define NL_D(RR) case RR: if (acc) gauss_yx_dual_kernel<RR, true><<<nb, GYX_THREADS, 0, c->stream>>>(src, dst, v, z0, z1, wya, wxa, wyb, wxb, vec4, (int)g2.x, (int)g2.y); \
                          else gauss_yx_dual_kernel<RR, false><<<nb, GYX_THREADS, 0, c->stream>>>(src, dst, v, z0, z1, wya, wxa, wyb, wxb, vec4, (int)g2.x, (int)g2.y); return true;
        NL_R12(NL_D)
#undef NL_D
    }
    return false;
}

bool gl_y_then_x(nl_ctx *c, bool acc, int r, const float *src, float *tmp, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wy,
                 const GaussWS &wx) {
    const dim3 gym((unsigned)((c->nx + 63) / 64), (unsigned)((z1 - z0 + 3) / 4), (unsigned)((c->ny + v.chunk - 1) / v.chunk));
    const dim3 gxk((unsigned)((c->nx + GX_SEG - 1) / GX_SEG), (unsigned)c->ny, (unsigned)(z1 - z0));
    const int vec4 = (c->nx % 4 == 0) ? 1 : 0;
    switch (r) {
#define NL_S(RR) case RR: gauss_march_kernel<1, RR><<<gym, 256, 0, c->stream>>>(src, tmp, v, z0, z1, wy);                      \
                          gauss_x_kernel<RR><<<gxk, 256, 0, c->stream>>>(tmp, dst, v, z0, z1, wx, vec4, acc ? 1 : 0); return true;
        NL_R12(NL_S)
#undef NL_S
    }
    return false;
}

bool gl_log_yx_sparse(nl_ctx *c, bool dual, bool acc, int r, const float *src, float *dst, const VolGeom &v, const int *list, int ntiles,
                      const GaussWS &wya, const GaussWS &wxa, const GaussWS &wyb, const GaussWS &wxb) {
    if (ntiles <= 0) return r >= 1 && r <= GM_MAX_R;
    const int ntx = (int)((c->nx + LS_TX - 1) / LS_TX), nty = (int)((c->ny + LS_TY - 1) / LS_TY);
    switch (r) {
#define NL_LS(RR) case RR:                                                                                                          \
        if (dual) { if (acc) log_yx_sparse_kernel<RR, true, true><<<ntiles, LS_NT, 0, c->stream>>>(src, dst, v, list, ntx, nty, wya, wxa, wyb, wxb); \
                    else log_yx_sparse_kernel<RR, true, false><<<ntiles, LS_NT, 0, c->stream>>>(src, dst, v, list, ntx, nty, wya, wxa, wyb, wxb); }  \
        else { if (acc) log_yx_sparse_kernel<RR, false, true><<<ntiles, LS_NT, 0, c->stream>>>(src, dst, v, list, ntx, nty, wya, wxa, wyb, wxb);    \
               else log_yx_sparse_kernel<RR, false, false><<<ntiles, LS_NT, 0, c->stream>>>(src, dst, v, list, ntx, nty, wya, wxa, wyb, wxb); }      \
        return true;
        NL_R12(NL_LS)
#undef NL_LS
    }
    return false;
}

bool gl_z_dual(nl_ctx *c, int r, const float *src, float *dst_a, float *dst_b, const VolGeom &v, i64 z0, i64 z1, const GaussWS &wa, const GaussWS &wb,
               const unsigned long long *need_bits) {
    if (r < 1 || r > GM_MAX_R || r > c->gnz || (need_bits && v.chunk > 64)) return false;
    const dim3 grid((unsigned)((c->nx + 63) / 64), (unsigned)((c->ny + 3) / 4), (unsigned)((z1 - z0 + v.chunk - 1) / v.chunk));
    switch (r) {
#define NL_ZD(RR) case RR: gauss_march_dual_kernel<RR><<<grid, 256, 0, c->stream>>>(src, dst_a, dst_b, v, z0, z1, wa, wb, need_bits); return true;
        NL_R12(NL_ZD)
#undef NL_ZD
    }
    return false;
}
