"""Unsorted (index-order) batches with a straggler pool (scratch, CPU model): a batch leaves the loop when at most P
lanes still iterate; those lanes' states go to a pool and are finished later in full waves.  Iteration-equivalents
per 64 cells."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as orc, util
from coflux import interface_computations as ic
nx, ny = 1440, 560
case = util.build_case(nx, ny, 7, 7)
params = ic.flux_params()
g_ = orc.make_grid(nx, ny, 7, 7, 1)
atmos = orc.interpolate_atmosphere_state(g_, case["src"], case["weights"], 0, 1, 0.37)
fl = orc.compute_atmosphere_ocean_fluxes(g_, params, case["ocean"], atmos, nthreads=0)
it = fl["iterations"][6:7 + ny + 1, 6:7 + nx + 1].ravel()
t = it[it > 0]
n = t.size // 64 * 64
B = t[:n].reshape(-1, 64)
print("cells", n, "mean trip %.2f  batch max %.2f" % (t.mean(), B.max(axis=1).mean()))
STASH = 0.35   # iteration-equivalents to write / read one wave's worth of pool entries
for P in (0, 2, 4, 8, 12, 16, 24, 32):
    cost = 0.0
    pool = []
    for row in B:
        s = np.sort(row)[::-1]
        stop = s[P] if P < 64 else 0          # iterations until at most P lanes remain
        stop = max(stop, 0)
        cost += stop
        rem = s[:P] - stop
        rem = rem[rem > 0]
        pool.extend(rem.tolist())
    pool = np.array(pool)
    # pool drained in order of arrival in full waves of 64
    m = pool.size // 64 * 64
    drain = pool[:m].reshape(-1, 64).max(axis=1).sum() + (pool[m:].max() if pool.size > m else 0)
    waves = (pool.size + 63) // 64
    total = cost + drain + STASH * 2 * waves
    print("P=%2d: batch part %.2f + pool %.2f (%.1f%% of lanes pooled) + stash => %.2f per batch" % (P, cost / len(B), drain / len(B), 100.0 * pool.size / n, total / len(B)))
