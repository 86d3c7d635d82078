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

    # ------------------------------------------------------------------ Label on slabs (numpy stand-in for nl_slab_*)
    # Same contract as the device entry points (include/nellie_amd.h): trees over the owned planes + one ghost plane per
    # interior side, run tables of the four planes the neighbours also see, patches, per-phase application.  The trees
    # here are scipy's full components (the device keeps face-touching background runs apart): the protocol only needs
    # a valid forest.
    def label_load_frangi(self, frangi, z0=0, z1=None):
        z1 = self.shape[0] if z1 is None else z1
        self.frangi = np.zeros(self.shape, np.float32)
        self.frangi[z0:z1] = frangi

    def label_intensity_mask(self, original, thresh, z0=None, z1=None):
        z0 = 0 if z0 is None else z0
        z1 = self.shape[0] if z1 is None else z1
        keep = np.asarray(original).astype(np.float64) > float(thresh)
        self.frangi[z0:z1] = np.where(keep, self.frangi[z0:z1], np.float32(0))

    def slab_label_pack(self, thr):
        lo, hi = self.own
        self.bits = [np.zeros(self.shape, bool), np.zeros(self.shape, bool)]
        if thr is not None:
            self.bits[0][lo:hi] = self.frangi[lo:hi] > np.float32(thr)

    def slab_bits_get(self, which, plane):
        return self.bits[which][plane].copy()

    def slab_bits_put(self, which, plane, words):
        self.bits[which][plane] = words

    def _ext(self):
        lo, hi = self.own
        e0 = lo - (1 if self.gz0 + lo > 0 else 0)
        e1 = hi + (1 if self.gz0 + hi < self.gnz else 0)
        return e0, e1

    def _slab_components(self, phase):
        from scipy import ndimage as ndi
        lo, hi = self.own
        e0, e1 = self._ext()
        m = self.bits[0][e0:e1]
        if phase == 0:
            m = ~m
        nz, ny, nx = m.shape
        # runs in raster order
        flat = np.concatenate([np.zeros((nz * ny, 1), bool), m.reshape(nz * ny, nx), np.zeros((nz * ny, 1), bool)], axis=1)
        d = np.diff(flat.astype(np.int8), axis=1)
        rows, xs = np.nonzero(d == 1)
        _, xe = np.nonzero(d == -1)
        structure = ndi.generate_binary_structure(3, 1) if phase == 0 else np.ones((3, 3, 3), bool)
        comp, _ = ndi.label(m, structure=structure)
        rc = comp.reshape(nz * ny, nx)[rows, xs]
        nruns = rows.size
        first = {}
        parent = np.empty(nruns, np.int64)
        for i in range(nruns):
            parent[i] = first.setdefault(int(rc[i]), i)
        self._sl = dict(phase=phase, rows=rows, xs=xs, xe=xe, parent=parent, e0=e0, ny=ny, nx=nx, nz=nz)
        own_rows = (rows >= (lo - e0) * ny) & (rows < (hi - e0) * ny)
        if phase == 0:
            gz = self.gz0 + e0 + rows // ny
            y = rows % ny
            face = (gz == 0) | (gz == self.gnz - 1) | (y == 0) | (y == ny - 1) | (xs == 0) | (xe == nx)
            aux = np.zeros(nruns, np.int64)
            np.maximum.at(aux, parent, face.astype(np.int64))
        elif phase == 1:
            aux = np.zeros(nruns, np.int64)
            np.add.at(aux, parent[own_rows], (xe - xs)[own_rows])
        else:
            aux = np.full(nruns, 2 ** 31 - 1, np.int64)
            np.minimum.at(aux, parent[own_rows], np.flatnonzero(own_rows))
        self._sl["aux"] = aux
        planes = [0 if e0 < lo else None, lo - e0, hi - 1 - e0, (hi - e0) if e1 > hi else None]
        self._sl["plane_runs"] = [np.flatnonzero(rows // ny == p) if p is not None else np.zeros(0, np.int64) for p in planes]
        return nruns, [int(r.size) for r in self._sl["plane_runs"]]

    def slab_phase(self, phase, gather_world=0):
        """The rank's tables of one phase as the library's blob (one entry per RUN here: any unit both ranks cut a shared plane
        into alike will do -- the library uses segment components)."""
        from nellie_amd.sharded import pack_slab_tables
        assert not gather_world, "the CPU double gathers through its communicator"
        self._slab_components(phase)
        sl = self._sl
        roots = [sl["parent"][r].astype(np.int32) for r in sl["plane_runs"]]
        vals = [sl["aux"][sl["parent"][r]].astype(np.int32) for r in sl["plane_runs"]]
        return [pack_slab_tables(roots, vals, nruns=sl["parent"].size)]

    def slab_patch(self, roots, values):
        self._sl["aux"][np.asarray(roots, np.int64)] = np.asarray(values, np.int64)

    def _own_run_mask(self):
        sl = self._sl
        lo, hi = self.own
        return (sl["rows"] >= (lo - sl["e0"]) * sl["ny"]) & (sl["rows"] < (hi - sl["e0"]) * sl["ny"])

    def _paint_runs(self, target, sel, values=None):
        sl = self._sl
        view = target[sl["e0"]:sl["e0"] + sl["nz"]].reshape(sl["nz"] * sl["ny"], sl["nx"])
        for i in np.flatnonzero(sel):
            view[sl["rows"][i], sl["xs"][i]:sl["xe"][i]] = True if values is None else values[i]

    def slab_apply(self, min_area=0):
        sl = self._sl
        own = self._own_run_mask()
        if sl["phase"] == 0:
            self._paint_runs(self.bits[0], own & (sl["aux"][sl["parent"]] == 0))
        else:
            lo, hi = self.own
            self.bits[1][lo:hi] = False
            self._paint_runs(self.bits[1], own & (sl["aux"][sl["parent"]] >= min_area))

    def slab_majority(self):
        lo, hi = self.own
        e0, e1 = self._ext()
        p = np.pad(self.bits[1][e0:e1].astype(np.int32), 1, mode="edge")      # edge = clamp: right at true faces, unused at ghosts
        s = sum(p[a:a + e1 - e0, b:b + self.shape[1], c:c + self.shape[2]] for a in range(3) for b in range(3) for c in range(3))
        self.bits[0][lo:hi] = (s >= 14)[lo - e0:hi - e0]

    def slab_number(self, clear, select):
        sl = self._sl
        n = sl["parent"].size
        sel = (sl["parent"] == np.arange(n)) & self._own_run_mask()
        sel[np.asarray(clear, np.int64)] = False
        sel[np.asarray(select, np.int64)] = True
        self._sl["sel"] = sel
        self._sl["rank_of"] = np.cumsum(sel) - sel          # exclusive
        return int(sel.sum()), (self._sl["rank_of"][np.asarray(select, np.int64)] + 1).astype(np.int32)

    def slab_paint(self, base, roots, labels):
        sl = self._sl
        newid = np.where(sl["sel"], base + sl["rank_of"] + 1, 0).astype(np.int64)
        newid[np.asarray(roots, np.int64)] = np.asarray(labels, np.int64)
        self.labels = np.zeros(self.shape, np.int32)
        self._paint_runs(self.labels, self._own_run_mask(), newid[sl["parent"]])

    def label_store(self, z0=0, z1=None, out=None):
        z1 = self.shape[0] if z1 is None else z1
        return np.ascontiguousarray(self.labels[z0:z1])
