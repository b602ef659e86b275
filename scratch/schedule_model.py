"""Where does convergence-mode time go?  Replays the solver's schedule on the oracle's trip counts."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "climaocean.jl_amd")]
import numpy as np, util, oracle as orc
from coflux import interface_computations as ic
nx, ny, h = 1440, 560, 7
case = util.build_case(nx, ny, h, h)
g = orc.make_grid(nx, ny, h, h, 1)
for cfgname in ("default", "corrected"):
    fl, vd = util.CONFIGS[cfgname]()
    params = ic.flux_params(fl, velocity_difference=vd)
    at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
    out = orc.compute_atmosphere_ocean_fluxes(g, params, case["ocean"], at, nthreads=8)
    it = util.window(out["iterations"], h, h, nx, ny, 1).ravel()
    wet = util.window(case["ocean"]["mask"], h, h, nx, ny, 1).ravel() != 0
    itw = it[wet]
    print(cfgname, "wet", wet.sum(), "mean it", itw.mean(), "max", itw.max())
    for W in (256, 512, 768):
        n = len(itw) // W
        chunks = itw[: n * W].reshape(n, W)
        srt = -np.sort(-chunks, axis=1)                    # longest first
        batches = srt.reshape(n, W // 64, 64).max(axis=2)  # batch time = slowest lane
        lane_eff = chunks.mean() / batches.mean()
        # LPT on 4 waves
        mk = np.zeros(n)
        for c in range(n):
            waves = np.zeros(4)
            for b in batches[c]:
                waves[np.argmin(waves)] += b
            mk[c] = waves.max()
        wave_eff = batches.sum(axis=1).mean() / 4 / mk.mean()
        print(f"  W={W}: lane eff {lane_eff:.3f}  wave packing {wave_eff:.3f}  WG makespan mean {mk.mean():.1f} max {mk.max():.1f} (max/mean {mk.max()/mk.mean():.3f})")
