"""Would longest-first batch order inside a chunk shorten the solver's workgroups?  Per-cell trip counts of the bench surface
(GPU run), chunks of 768 consecutive wet cells, batches of 64, four waves claiming batches dynamically; a batch costs
A + trips (in iteration units; A = prologue + epilogue + memory phase ~ 5).  Prints the mean / max workgroup makespan for
index order, longest-first by the exact maxima, and longest-first by one-step-old maxima (the other ocean state)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ctx = FluxContext(nx, ny, h, h, ic.flux_params(ic.SimilarityTheoryFluxes(), ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0)), ring=1)
o0 = syn.ocean_state(nx, ny, h, h); o1 = syn.evolved_ocean_state(o0, nx, ny, h, h, 1)
src = {k: ctx.to_device(v) for k, v in syn.jra55_snapshots(4, temporal_correlation=0.95).items()}
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
at, fl, net = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES[:5])
fl["iterations"] = ctx.zeros(torch.int32)
trips = []
for s, o in enumerate((o0, o1)):
    st = {k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")}
    ctx.update_state(src, w, st, at, fl, net, level1=0, level2=1, time_fraction=0.1 + s / 9)
    ctx.sync()
    it = fl["iterations"].cpu().numpy()[h - 1:h + ny + 1, h - 1:h + nx + 1].ravel()
    m = o0["mask"][h - 1:h + ny + 1, h - 1:h + nx + 1].ravel() != 0
    trips.append(it[m].astype(np.int64))
cur, old = trips[1], trips[0]
print("wet cells", cur.size, "mean trips", cur.mean(), "max", cur.max())
A = 5.0
def makespan(costs):
    waves = [0.0] * 4
    for c in costs:
        k = int(np.argmin(waves)); waves[k] += c
    return max(waves), sum(waves) / 4
res = {"index": [], "lpt_exact": [], "lpt_old": []}
for c0 in range(0, cur.size - 767, 768):
    b = cur[c0:c0 + 768].reshape(12, 64).max(axis=1) + A
    bo = old[c0:c0 + 768].reshape(12, 64).max(axis=1) + A
    res["index"].append(makespan(b))
    res["lpt_exact"].append(makespan(b[np.argsort(-b)]))
    res["lpt_old"].append(makespan(b[np.argsort(-bo, kind="stable")]))
print("batch max mean", np.mean([cur[c0:c0 + 768].reshape(12, 64).max(axis=1).mean() for c0 in range(0, cur.size - 767, 768)]))
for k, v in res.items():
    v = np.array(v)
    print(f"{k:10s} mean makespan {v[:,0].mean():7.2f}  p95 {np.percentile(v[:,0],95):7.2f}  max {v[:,0].max():7.2f}   mean wave load {v[:,1].mean():7.2f}")
