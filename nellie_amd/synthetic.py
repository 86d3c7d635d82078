"""
Deterministic synthetic volumes for parity tests and `bench.py`.

The reference ships no benchmark inputs (its sample file is absent from the
checkout), so the build defines its own generator: N(100, 5) float32 noise plus
axis-aligned tube segments with a Gaussian cross-section (amplitude 200, radius
U(1.5, 4.0) voxels), one third of them along each of X, Y and Z, each covering a
random 20-60 % stretch of its axis (so a volume holds many separate objects, not
one connected net).  A tube's profile is evaluated inside a +-5r window (it is
< 1e-3 of a noise sigma beyond).  `numpy.random.default_rng` (PCG64) streams are stable across numpy
versions, so a (shape, seed) pair names the same bytes everywhere.
"""
from __future__ import annotations

import numpy as np

SEEDS = {"C2": 1234, "C3": 2345, "C4": 3456, "C5": 4567}


def n_tubes(shape) -> int:
    n = int(np.prod(shape))
    return max(6, int(n / 2 ** 20 * 0.75))


def make_volume(shape, seed: int, tubes: int | None = None, dtype=np.float32,
                z_offset: int = 0, global_nz: int | None = None) -> np.ndarray:
    """
    (Z, Y, X) volume.  With `z_offset`/`global_nz` a rank generates only its own
    Z-slab `[z_offset, z_offset + shape[0])` of the global (global_nz, Y, X)
    volume: tube geometry is drawn for the global shape, noise per global plane.
    """
    nz, ny, nx = (int(s) for s in shape)
    gz = int(global_nz) if global_nz is not None else nz
    gshape = (gz, ny, nx)
    k = n_tubes(gshape) if tubes is None else int(tubes)
    vol = np.empty((nz, ny, nx), dtype=np.float32)
    # noise: one independent stream per global plane so slabs agree with the full volume
    for z in range(nz):
        prng = np.random.default_rng([int(seed), 1, z + z_offset])
        vol[z] = prng.standard_normal((ny, nx), dtype=np.float32) * np.float32(5.0) + np.float32(100.0)
    rng = np.random.default_rng([int(seed), 0])
    axes = rng.integers(0, 3, size=k) if k % 3 else np.repeat(np.arange(3), k // 3)
    radii = rng.uniform(1.5, 4.0, size=k)
    cu = rng.uniform(0.0, 1.0, size=k)
    cv = rng.uniform(0.0, 1.0, size=k)
    seg_len = rng.uniform(0.2, 0.6, size=k)
    seg_pos = rng.uniform(0.0, 1.0, size=k)
    dims = (gz, ny, nx)
    for ax, r, u, v, sl, sp in zip(axes, radii, cu, cv, seg_len, seg_pos):
        # the stretch [a0, a1) of the tube's own axis it occupies
        length = max(4, int(round(sl * dims[ax])))
        a0 = int(round(sp * (dims[ax] - length)))
        a1 = min(dims[ax], a0 + length)
        other = [d for d in range(3) if d != ax]
        c0 = u * (dims[other[0]] - 1)
        c1 = v * (dims[other[1]] - 1)
        w = int(np.ceil(5.0 * r))
        lo0, hi0 = max(0, int(c0) - w), min(dims[other[0]], int(c0) + w + 1)
        lo1, hi1 = max(0, int(c1) - w), min(dims[other[1]], int(c1) + w + 1)
        g0 = np.arange(lo0, hi0, dtype=np.float64)
        g1 = np.arange(lo1, hi1, dtype=np.float64)
        prof = (200.0 * np.exp(-((g0[:, None] - c0) ** 2 + (g1[None, :] - c1) ** 2)
                               / (2.0 * r * r))).astype(np.float32)
        if ax == 0:      # along Z: profile over (Y, X), planes a0..a1 clipped to this slab
            zlo, zhi = max(a0, z_offset), min(a1, z_offset + nz)
            if zlo < zhi:
                vol[zlo - z_offset:zhi - z_offset, lo0:hi0, lo1:hi1] += prof[None, :, :]
        else:
            # profile over (Z, other): clip the Z window to this slab
            zlo, zhi = max(lo0, z_offset), min(hi0, z_offset + nz)
            if zlo >= zhi:
                continue
            p = prof[zlo - lo0:zhi - lo0]
            if ax == 1:  # along Y: profile over (Z, X)
                vol[zlo - z_offset:zhi - z_offset, a0:a1, lo1:hi1] += p[:, None, :]
            else:        # along X: profile over (Z, Y)
                vol[zlo - z_offset:zhi - z_offset, lo1:hi1, a0:a1] += p[:, :, None]
    if np.dtype(dtype) != np.float32:
        info = np.iinfo(dtype)
        vol = np.clip(np.rint(vol), info.min, info.max).astype(dtype)
    return vol


ISO_01 = {"X": 0.1, "Y": 0.1, "Z": 0.1, "T": 1.0}
ANISO_03 = {"X": 0.1, "Y": 0.1, "Z": 0.3, "T": 1.0}


def make_image_2d(shape, seed, dtype=np.float32):
    """Deterministic (Y, X) test image for the 2-D (no_z) path: N(100, 5) noise + a horizontal ridge, a diagonal
    ridge and two blobs."""
    rng = np.random.default_rng(seed)
    ny, nx = shape
    img = rng.normal(100.0, 5.0, shape)
    yy, xx = np.mgrid[:ny, :nx]
    img += 200.0 * np.exp(-((yy - 0.45 * ny) ** 2) / (2 * 2.5 ** 2))
    img += 160.0 * np.exp(-((yy - 0.9 * xx - 0.1 * ny) ** 2) / (2 * 3.0 ** 2))
    for cy, cx, r in ((0.2 * ny, 0.7 * nx, 3.0), (0.75 * ny, 0.3 * nx, 4.0)):
        img += 180.0 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r ** 2))
    if np.issubdtype(np.dtype(dtype), np.integer):
        return np.clip(np.rint(img), 0, np.iinfo(dtype).max).astype(dtype)
    return img.astype(dtype)


def make_skeleton(shape, seed, n_walks=None, branch_prob=0.04):
    """A synthetic skeleton-like int32 label image: one-voxel-wide random walks (26-/8-connected steps with momentum)
    that fork now and then, so that tips, edges, junction clusters and isolated voxels all occur.  Deterministic."""
    rng = np.random.default_rng(seed)
    shape = tuple(int(s) for s in shape)
    nd = len(shape)
    out = np.zeros(shape, np.int32)
    n = int(np.prod(shape))
    if n_walks is None:
        n_walks = max(3, n // 2500)
    hi = np.array(shape) - 1
    for lab in range(1, n_walks + 1):
        stack = [(np.array([rng.integers(0, s) for s in shape]), rng.integers(-1, 2, nd))]
        budget = int(rng.integers(20, 250))
        while stack and budget > 0:
            pos, step = stack.pop()
            length = int(rng.integers(3, 25))
            for _ in range(length):
                if not step.any():
                    step = rng.integers(-1, 2, nd)
                    continue
                out[tuple(pos)] = lab
                budget -= 1
                if rng.random() < 0.3:
                    step = np.clip(step + rng.integers(-1, 2, nd), -1, 1)
                if rng.random() < branch_prob:
                    stack.append((pos.copy(), rng.integers(-1, 2, nd)))
                pos = np.clip(pos + step, 0, hi)
    # a few isolated voxels
    for _ in range(max(1, n_walks // 3)):
        out[tuple(rng.integers(0, s) for s in shape)] = n_walks + 1
    return out
