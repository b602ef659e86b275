"""Predicted stragglers first, the rest in index order (scratch, CPU model on the oracle's trip counts, bench.py's schedule:
two ocean states alternating, JRA55 snapshots correlated 0.95, the clock advancing 20 minutes per step).
A chunk's list = [cells whose trip count LAST step was >= T, longest first] + [the others, index order]; batches of 64.
Prints iteration-equivalents per 64 cells (batch maximum + 2.3 for prologue/epilogue; + PENALTY per batch whose cells are
scattered) against index-ordered batches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as orc, util
from coflux import interface_computations as ic, synthetic as syn
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1440, 560)
conf = sys.argv[3] if len(sys.argv) > 3 else "default"
h = 7
fluxes_cfg = {"default": ic.SimilarityTheoryFluxes, "corrected": ic.corrected_atmosphere_ocean_fluxes}[conf]()
params = ic.flux_params(fluxes_cfg, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
first = syn.ocean_state(nx, ny, h, h)
states = [first, syn.evolved_ocean_state(first, nx, ny, h, h, 1)]
src = syn.jra55_snapshots(4, temporal_correlation=0.95)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
w = dict(separable=True, fi=fi, fj=fj, latitude=phi)
g_ = orc.make_grid(nx, ny, h, h, 1)
inc = 1200.0 / 10800.0
def trips(s):
    tot = s * inc
    l1 = int(tot) % 4
    atmos = orc.interpolate_atmosphere_state(g_, src, w, l1, (l1 + 1) % 4, tot - int(tot))
    fl = orc.compute_atmosphere_ocean_fluxes(g_, params, states[s % 2], atmos, nthreads=0)
    W = (slice(h - 1, h + ny + 1), slice(h - 1, h + nx + 1))
    it = fl["iterations"][W].ravel()
    return it
S0 = int(os.environ.get("STEP", "40"))
t_prev_all, t_now_all = trips(S0 - 1), trips(S0)
wet = t_now_all > 0
t_prev, t_now = t_prev_all[wet], t_now_all[wet]
print("wet", t_now.size, "mean trip %.2f" % t_now.mean(), "hist from 5:", np.bincount(t_now)[5:25])
print("T: P(now>=T | prev>=T), P(prev>=T | now>=T), frac now>=T")
for T in (13, 14, 15, 16):
    a = t_prev >= T; b = t_now >= T
    print(T, "%.3f %.3f %.4f" % ((a & b).sum() / max(a.sum(), 1), (a & b).sum() / max(b.sum(), 1), b.mean()))
PRO, PENALTY = 2.3, float(os.environ.get("PENALTY", "3.5"))
for chunk in (1024, 512):
    base = None
    for frac in (0.0, 0.03, 0.0625, 0.125, 0.1875, 0.25, 1.0):
        for label, key in (("exact", t_now), ("old", t_prev)):
            tot = 0.0; ncells = 0; pen = 0.0
            for c0 in range(0, t_now.size, chunk):
                tn = t_now[c0:c0 + chunk]; tk = key[c0:c0 + chunk]
                # per-chunk threshold: the smallest T with count(key >= T) <= frac * n
                n = tn.size
                T = 99
                for cand in range(30, 0, -1):
                    if (tk >= cand).sum() <= frac * n: T = cand
                    else: break
                s = tk >= T
                si = np.flatnonzero(s)
                si = si[np.argsort(-tk[si], kind="stable")]
                order = np.concatenate([si, np.flatnonzero(~s)])
                cells = tn[order]
                m = (cells.size + 63) // 64
                pad = np.zeros(m * 64, dtype=cells.dtype); pad[:cells.size] = cells
                b = pad.reshape(-1, 64).max(axis=1)
                tot += (b + PRO).sum(); ncells += cells.size
                pen += PENALTY * ((si.size + 63) // 64)
            per64 = tot / ncells * 64; p64 = (tot + pen) / ncells * 64
            if frac == 0.0 and label == "exact": base = per64
            print("chunk %4d stragglers <= %5.1f %% %-5s: %.2f (%+.1f %%), with scatter penalty %.2f (%+.1f %%)" % (chunk, 100 * frac, label, per64, 100 * (per64 / base - 1), p64, 100 * (p64 / base - 1)))
