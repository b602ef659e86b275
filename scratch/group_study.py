"""Trip-count sorting at group granularity (scratch, CPU): batches of 64 cells made of 64/g groups of g consecutive wet
cells, groups ordered by their largest trip count within chunks of 1280 wet cells.  Prints mean wave trips per batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as orc, util
from coflux import interface_computations as ic
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1440, 560)
case = util.build_case(nx, ny, 7, 7)
params = ic.flux_params()
g_ = orc.make_grid(nx, ny, 7, 7, 1)
def trips(tf):
    atmos = orc.interpolate_atmosphere_state(g_, case["src"], case["weights"], 0, 1, tf)
    fl = orc.compute_atmosphere_ocean_fluxes(g_, params, case["ocean"], atmos, nthreads=0)
    W = (slice(6, 7 + ny + 1), slice(6, 7 + nx + 1))
    it = fl["iterations"][W].ravel()
    return it[it > 0]
t_now = trips(0.37)
t_prev = trips(0.37 - 20.0 / 180.0)   # one 20-minute step earlier (3-hourly snapshots)
print("wet", t_now.size, "mean trip", t_now.mean())
for chunk in (1280,):
    for g in (1, 2, 4, 8, 16, 32, 64):
        for label, key_src in (("exact keys", t_now), ("one-step-old keys", t_prev)):
            tot = 0.0; nb = 0
            for c0 in range(0, t_now.size, chunk):
                tn = t_now[c0:c0 + chunk]; tk = key_src[c0:c0 + chunk]
                n = tn.size // g * g
                if n == 0: continue
                keys = tk[:n].reshape(-1, g).max(axis=1)
                order = np.argsort(-keys, kind="stable")
                cells = tn[:n].reshape(-1, g)[order].ravel()
                m = cells.size // 64 * 64
                if m:
                    b = cells[:m].reshape(-1, 64).max(axis=1)
                    tot += b.sum(); nb += b.size
            print("chunk %d group %2d %-18s trips per batch %.2f" % (chunk, g, label, tot / nb))
