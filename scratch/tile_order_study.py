"""Would 2-D batches lower the iterations a batch runs (= the maximum over its 64 cells)?  Oracle trip counts on the
synthetic 1/4-degree surface; wet cells listed tile by tile (tiles of w x h cells, row-major inside a tile, tiles in
row-major order), cut into batches of 64.  (scratch study, CPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import oracle as orc
from coflux import interface_computations as ic, synthetic as syn

nx, ny, h = 1440, 560, 7
params = ic.flux_params(ic.SimilarityTheoryFluxes())
ocean = syn.ocean_state(nx, ny, h, h)
snaps = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
g = orc.make_grid(nx, ny, h, h, 1)
w = dict(separable=True, fi=fi, fj=fj, latitude=phi)
at = orc.interpolate_atmosphere_state(g, snaps, w, 0, 1, 0.37)
fl = orc.compute_atmosphere_ocean_fluxes(g, params, ocean, at, nthreads=0)
it = fl["iterations"][h - 1:h + ny + 1, h - 1:h + nx + 1].astype(np.int64)   # the window the kernel covers (ring 1)
H, W = it.shape
wet = it > 0
print("wet", wet.sum(), "mean trips", it[wet].mean())
jj, ii = np.mgrid[0:H, 0:W]
for tw, th in ((64, 1), (32, 2), (16, 4), (8, 8), (4, 16), (128, 1), (16, 8)):
    key = ((jj // th) * ((W + tw - 1) // tw) + (ii // tw)) * (tw * th) + (jj % th) * tw + (ii % tw)
    order = np.argsort(key[wet], kind="stable")
    t = it[wet][order]
    n = (len(t) // 64) * 64
    mx = t[:n].reshape(-1, 64).max(axis=1)
    print(f"tile {tw:3d} x {th:2d}: mean max over a batch {mx.mean():.3f}  (lane-iterations wasted {1 - t[:n].sum() / (mx.sum() * 64):.3f})")
