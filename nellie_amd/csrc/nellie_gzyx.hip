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

// planes a workgroup marches: 128 (measured best at 1024^3: 64 -> 2.93 ms, 128 -> 2.78, 256 -> 2.90 for R = 4), fewer where the
// chunk's planes + 2 rz would not fit the 4 GiB a buffer resource addresses; 0: not even 8 planes do
static int zyx_chunk(const nl_ctx *c, int rz) {
    const i64 plane_bytes = c->ny * c->nx * 4;
    int zchunk = 128;
    while (zchunk > 8 && (i64)(zchunk + 2 * rz) * plane_bytes >= ((i64)1 << 32)) zchunk >>= 1;
    return (i64)(zchunk + 2 * rz) * plane_bytes < ((i64)1 << 32) ? zchunk : 0;
}

bool gl_zyx_ok(const nl_ctx *c, int rz, int r, const float *dst) {
    // radii with an instantiation, one reflection at most (the kernel's index rule), 16-byte aligned rows to store into
    return rz >= 1 && rz <= 5 && r >= 3 && r <= 5 && rz <= c->gnz && r <= c->ny && r <= c->nx && !c->two_d && ((size_t)dst & 15) == 0 &&
           zyx_chunk(c, rz) > 0;
}

bool gl_zyx(nl_ctx *c, int rz, int r, const float *src, float *dst, const VolGeom &v, i64 z0, i64 z1, const GaussWS &gz, const GaussWS &gyx,
            float *zero_out) {
    if (!gl_zyx_ok(c, rz, r, dst)) return false;
    const int zchunk = zyx_chunk(c, rz);
#define NL_ZYX(RZ_, R_) case RZ_ * 8 + R_: launch_zyx<RZ_, R_>(c, src, dst, v, z0, z1, zchunk, gz, gyx, zero_out); return true;
    switch (rz * 8 + r) {
        NL_ZYX(1, 3) NL_ZYX(1, 4) NL_ZYX(1, 5) NL_ZYX(2, 3) NL_ZYX(2, 4) NL_ZYX(2, 5) NL_ZYX(3, 3) NL_ZYX(3, 4) NL_ZYX(3, 5)
        NL_ZYX(4, 3) NL_ZYX(4, 4) NL_ZYX(4, 5) NL_ZYX(5, 3) NL_ZYX(5, 4) NL_ZYX(5, 5)
    }
#undef NL_ZYX
    return false;
}
