import sys, warnings, numpy as np
sys.path.insert(0, '.')
warnings.simplefilter("ignore")
from nellie_amd import pipeline as pl
from nellie_amd.synthetic import ISO_01, make_volume
from oracle import nellie_oracle as orc
shape = (20, 40, 40)
vol = make_volume(shape, 9)
vol[10, 20, 20] = np.nan
tr = []
vess, masks = orc.compute_vesselness(vol, ISO_01, mask=False, trace=tr)
pipe = pl.FramePipeline(shape)
pipe.ctx.filter_load(vol)
sig = pl.default_sigmas(ISO_01)
for s, delta in enumerate(pl.cascade_deltas(sig, pl.z_ratio_of(ISO_01))):
    pipe.ctx.gauss_step(*[pl.gaussian_weights(d) for d in delta])
    g = pipe.ctx.gauss_store()
    og = tr[s]["gauss"]
    print("scale", s, "device nan", int(np.isnan(g).sum()), "oracle nan", int(np.isnan(og).sum()), "nan sets equal", np.array_equal(np.isnan(g), np.isnan(og)),
          "finite equal", np.array_equal(g[~np.isnan(og)], og[~np.isnan(og)]))
    h6 = orc.hessian_components(og, orc.spacing3(ISO_01))
    anynan = np.zeros(shape, bool)
    for c in h6:
        anynan |= ~np.isfinite(c)
    vs = tr[s].get("vessel_scale")
    print("   oracle: voxels with a non-finite Hessian entry", int(anynan.sum()), "of which response > 0:", int((vs[anynan] > 0).sum()) if vs is not None else None)
pipe.close()
# device per-scale response: run the scales one at a time through compute_vesselness with explicit sigma lists of growing length
prev = np.zeros(shape, np.float32)
for k in range(1, len(sig) + 1):
    pipe = pl.FramePipeline(shape)
    pipe.compute_vesselness(vol.copy(), pl.FilterParams(dim_res=ISO_01, sigmas=list(sig[:k])) if hasattr(pl.FilterParams, "sigmas") else pl.FilterParams(dim_res=ISO_01), mask=False)
    out = pipe.download_frangi()
    pipe.close()
    ref = np.zeros(shape, np.float32)
    for t in tr[:k]:
        if "vessel_scale" in t:
            ref = np.maximum(ref, t["vessel_scale"])
    tol = 1e-4 * np.abs(ref) + 1e-6 * np.abs(ref).max()
    bad = np.abs(out - ref) > tol
    print("first", k, "scales: bad", int(bad.sum()), [tuple(int(v) for v in b) for b in np.argwhere(bad)[:6]])
