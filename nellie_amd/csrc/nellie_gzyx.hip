// Translation unit of libnellie_hip.so (gfx950): the fused Z+Y+X cascade step (gauss_zyx.inc) and its launcher (gauss_launch.h).
#include "nl_host.h"
#include "gauss.inc"
#include "gauss_zyx.inc"

template <int RZ, int R>
static void launch_zyx(nl_ctx *c, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, int zchunk, const GaussWS &gz, const GaussWS &gyx,
                       float *zero_out) {
    constexpr int TY = GzyxCfg<R>::TY;
    const int ntx = (int)((c->nx + 63) / 64), nty = (int)((c->ny + TY - 1) / TY), nzc = (int)((z1 - z0 + zchunk - 1) / zchunk);
    gauss_zyx_kernel<RZ, R><<<(unsigned)ntx * nty * nzc, GZ_NT, 0, c->stream>>>(src, dst, v, (int)z0, (int)z1, zchunk, gz, gyx, ntx, nty, zero_out);
}

// planes a workgroup marches: 128 (measured best at 1024^3: 64 -> 2.93 ms, 128 -> 2.78, 256 -> 2.90 for R = 4); fewer on small volumes, whose
// tiles alone do not fill the chip (a 128 x 512 x 512 frame of BASELINE config 5 is 8 x 11 tiles: 88 workgroups for 256 CUs with one
// chunk -- the streamed stack went from 3.2 to 3.7 ms per frame before this rule), and where a chunk's planes + 2 rz would not fit the
// 4 GiB a buffer resource addresses; 0: not even 8 planes do
static int zyx_chunk(const nl_ctx *c, int rz, int r, i64 nplanes) {
    const i64 plane_bytes = c->ny * c->nx * 4;
    const int ty = r <= 2 ? GzyxCfg<2>::TY : (r <= 4 ? GzyxCfg<4>::TY : GzyxCfg<5>::TY);      // rows of a tile
    const i64 tiles = ((c->nx + 63) / 64) * ((c->ny + ty - 1) / ty);
    static int min_wgs = 0;                      // NELLIE_ZYX_MIN_WGS: A/B of the rule (workgroups a launch should have)
    if (!min_wgs) { const char *e = getenv("NELLIE_ZYX_MIN_WGS"); min_wgs = (e && atoi(e) > 0) ? atoi(e) : 512; }
    // planes a workgroup marches: 128 (R = 4 at 1024^3: 64 / 128 / 256 planes 2.91 / 2.73 / 2.92 ms), 256 for the 32-row tiles of R = 5, whose 2R-plane warm-up
    // weighs more (3.82 -> 3.71 ms; profiles/r05_ubench_gauss_zyx_lds64.txt)
    int zchunk = r >= 5 ? 256 : 128;
    while (zchunk > 8 && tiles * ((nplanes + zchunk - 1) / zchunk) < min_wgs) zchunk >>= 1;
    while (zchunk > 8 && (i64)(zchunk + 2 * rz) * plane_bytes >= ((i64)1 << 32)) zchunk >>= 1;
    return (i64)(zchunk + 2 * rz) * plane_bytes < ((i64)1 << 32) ? zchunk : 0;
}

bool gl_zyx_ok(const nl_ctx *c, int rz, int r, const float *dst) {
    // radii with an instantiation, one reflection at most (the kernel's index rule), 16-byte aligned rows to store into
    return rz >= 1 && rz <= 5 && r >= 3 && r <= 5 && rz <= c->gnz && r <= c->ny && r <= c->nx && !c->two_d && ((size_t)dst & 15) == 0 &&
           zyx_chunk(c, rz, r, c->nzl) > 0;
}

bool gl_zyx(nl_ctx *c, int rz, int r, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &gz, const GaussWS &gyx,
            float *zero_out) {
    if (!gl_zyx_ok(c, rz, r, dst)) return false;
    const int zchunk = zyx_chunk(c, rz, r, z1 - z0);
#define NL_ZYX(RZ_, R_) case RZ_ * 8 + R_: launch_zyx<RZ_, R_>(c, src, dst, v, z0, z1, zchunk, gz, gyx, zero_out); return true;
    switch (rz * 8 + r) {
        NL_ZYX(1, 3) NL_ZYX(1, 4) NL_ZYX(1, 5) NL_ZYX(2, 3) NL_ZYX(2, 4) NL_ZYX(2, 5) NL_ZYX(3, 3) NL_ZYX(3, 4) NL_ZYX(3, 5)
        NL_ZYX(4, 3) NL_ZYX(4, 4) NL_ZYX(4, 5) NL_ZYX(5, 3) NL_ZYX(5, 4) NL_ZYX(5, 5)
    }
#undef NL_ZYX
    return false;
}
