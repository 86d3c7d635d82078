"""
Test doubles for the Z-slab host logic (tests only).

`OracleCtx` implements the subset of `nellie_amd.hipnative.Context` that `FramePipeline` /
`ShardedFramePipeline` drive, with the ORACLE doing the arithmetic on a numpy slab.  It lets the slab
geometry, the ghost-depth bookkeeping and the collective threshold logic run on CPU under gloo
(world_size 2) -- the HIP engine itself is covered by the GPU tests.  Boundary rules at the ends of the local
array are only right where those ends are true global faces; everywhere else the planes they touch are
outside the valid range by construction, which is exactly what these tests verify.
"""
import numpy as np

from oracle import nellie_oracle as orc

FIELD_GAUSS, FIELD_FROB, FIELD_FRANGI = 0, 1, 2


class OracleCtx:
    def __init__(self, shape, device=0, gz0=0, gnz=None, own=None):
        self.shape = tuple(int(s) for s in shape)
        self.gz0 = int(gz0)
        self.gnz = int(gnz) if gnz is not None else self.shape[0]
        self.own = (0, self.shape[0]) if own is None else (int(own[0]), int(own[1]))
        self.gauss = np.zeros(self.shape, np.float32)
        self.vmax = np.zeros(self.shape, np.float32)
        self.cmask = np.ones(self.shape, bool)
        self.frangi = None
        self.norm = (1.0, 0.0)
        self.spacing = None

    def close(self):
        pass

    def sync(self):
        pass

    # ------------------------------------------------------------------ Filter
    def filter_load(self, frame, z0=0, z1=None):
        z1 = self.shape[0] if z1 is None else z1
        self.gauss = np.zeros(self.shape, np.float32)
        self.gauss[z0:z1] = np.asarray(frame, dtype=np.float32)
        self.vmax = np.zeros(self.shape, np.float32)
        self.cmask = np.ones(self.shape, bool)

    def gauss_step(self, wz, wy, wx, z0=0, z1=None):
        z1 = self.shape[0] if z1 is None else z1
        out = self.gauss
        for axis, w in enumerate((wz, wy, wx)):
            if w is not None:
                out = orc.correlate1d_reflect_f32(out, np.asarray(w)[::-1], axis)
        # only planes [z0, z1) are defined afterwards: poison the rest so a wrong halo depth cannot hide
        res = np.full(self.shape, np.float32(np.nan))
        res[z0:z1] = out[z0:z1]
        self.gauss = res

    def _lattice(self, strides):
        sz, sy, sx = strides
        lo, hi = self.own
        zs = [z for z in range(lo, hi) if (self.gz0 + z) % sz == 0]
        return zs, slice(None, None, sy), slice(None, None, sx)

    def _field(self, field):
        if field == FIELD_GAUSS:
            return self.gauss
        if field == FIELD_FRANGI:
            return self.frangi
        frob = self._frob()
        return frob

    def _frob(self):
        with np.errstate(all="ignore"):
            h6 = orc.hessian_components(np.nan_to_num(self.gauss, nan=0.0), self.spacing)
            fsq = h6[0] ** 2 + h6[3] ** 2 + h6[5] ** 2 + np.float32(2.0) * (h6[1] ** 2 + h6[2] ** 2 + h6[4] ** 2)
            frob = np.sqrt(fsq) / np.float32(self.norm[0])
        frob = np.where(np.isinf(frob), np.float32(self.norm[1]), frob)
        return frob, h6, fsq

    def sample_gather(self, field, strides):
        zs, ys, xs = self._lattice(strides)
        f = self._field(field)
        f = f[0] if isinstance(f, tuple) else f
        if not zs:
            return np.zeros(0, np.float32)
        return np.ascontiguousarray(f[zs][:, ys, xs]).reshape(-1).astype(np.float32)

    def sample_minmax(self, field, strides):
        s = self.sample_gather(field, strides)
        p = s[s > 0]
        if p.size == 0:
            return np.float32(0), np.float32(0), 0
        return p.min(), p.max(), int(p.size)

    def sample_hist(self, field, strides, edges):
        s = self.sample_gather(field, strides)
        p = s[s > 0]
        nb = len(edges) - 1
        if p.size == 0:
            return np.zeros(nb, np.int64)
        first, last = edges[0], edges[-1]
        fi = ((p - first) / np.float32(last - first)) * np.float32(nb)
        idx = fi.astype(np.intp)
        idx[idx == nb] -= 1
        idx[p < edges[idx]] -= 1
        inc = (p >= edges[idx + 1]) & (idx != nb - 1)
        idx[inc] += 1
        return np.bincount(idx, minlength=nb).astype(np.int64)

    def hessian_stats(self, spacing):
        self.spacing = tuple(float(s) for s in spacing)
        lo, hi = self.own
        self.norm = (1.0, 0.0)
        _, h6, fsq = self._frob()
        mabs = max(float(np.max(np.abs(c[lo:hi]))) for c in h6)
        f = fsq[lo:hi]
        fin = f[np.isfinite(f)]
        return np.float32(mabs), np.float32(fin.max() if fin.size else 0.0), bool(np.isinf(f).any())

    def set_frob_norm(self, max_abs, max_finite):
        self.norm = (float(max_abs), float(max_finite))

    def vesselness_step(self, gamma_sq, alpha_sq, beta_sq, thr, want_count=True, z0=-1, z1=-1):
        lo, hi = self.own
        if z0 < 0:
            z0, z1 = lo, hi
        frob, h6, _ = self._frob()
        with np.errstate(invalid="ignore"):
            m = (frob > np.float32(thr)) if thr is not None else (frob > 0)
        sel = np.zeros(self.shape, bool)
        sel[z0:z1] = True
        coords = np.where(m & sel)
        ev = orc.sort_by_abs(orc.eigvalsh3_f32(*[c[coords] for c in h6]))
        v = orc.frangi_response(ev, alpha_sq, beta_sq, gamma_sq)
        scale = np.zeros(self.shape, np.float32)
        scale[coords] = v
        self.vmax[z0:z1] = np.maximum(self.vmax[z0:z1], scale[z0:z1])
        self.cmask[z0:z1] &= m[z0:z1]
        return int(m[lo:hi].sum())

    def filter_finish(self, z0=-1, z1=-1):
        lo, hi = self.own
        if z0 < 0:
            z0, z1 = lo, hi
        self.frangi = np.zeros(self.shape, np.float32)
        self.frangi[z0:z1] = (self.vmax * self.cmask)[z0:z1]
        return int((self.frangi[lo:hi] > 0).sum())

    def mask_volume(self, thr):
        lo, hi = self.own
        m = self.frangi > np.float32(thr)
        # the opening needs the zero border only at true global faces; interior ghost planes are real data
        opened = orc.binary_dilation6(orc.binary_erosion6(m))
        res = self.frangi * opened
        out = np.zeros(self.shape, np.float32)
        out[lo:hi] = res[lo:hi]
        self.frangi = out

    def filter_store(self, z0=0, z1=None, out=None):
        z1 = self.shape[0] if z1 is None else z1
        return np.ascontiguousarray(self.frangi[z0:z1])

    def flat_sample_gather(self, field, offset, step):
        lo, hi = self.own
        plane = self.shape[1] * self.shape[2]
        g0, g1 = (self.gz0 + lo) * plane, (self.gz0 + hi) * plane
        k0 = 0 if g0 <= offset else -(-(g0 - offset) // step)
        k1 = -(-(g1 - offset) // step) if g1 > offset else 0
        idx = offset + np.arange(k0, max(k0, k1)) * step - self.gz0 * plane
        return self._field(field).reshape(-1)[idx].astype(np.float32)

    # ------------------------------------------------------------------ slabs
    def planes_get(self, field, z0, z1):
        return np.ascontiguousarray(self._field(field)[z0:z1])

    def planes_put(self, field, z0, z1, planes):
        self._field(field)[z0:z1] = planes
