"""CPU oracle pass time vs OpenMP thread count on this box (scratch): which setting the cpu_baseline leg should use."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("climaocean.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import oracle as orc, util
from coflux import interface_computations as ic
nx, ny, h = 1440, 560, 7
case = util.build_case(nx, ny, h, h)
params = ic.flux_params()
g = orc.make_grid(nx, ny, h, h, 1)
shape = (ny + 2 * h, nx + 2 * h)
atmos = {n: np.zeros(shape) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
fl = {n: np.zeros(shape) for n in ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature")}
net = {n: np.zeros(shape) for n in ("u", "v", "T", "S", "shortwave_surface_flux")}
print("max threads", orc.max_threads(), "cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64, 96, 128, 192, 256):
    if nt > (os.cpu_count() or 8): break
    best = [1e9, 1e9, 1e9]
    for _ in range(4):
        t0 = time.perf_counter(); orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37, out=atmos)
        t1 = time.perf_counter(); orc.compute_atmosphere_ocean_fluxes(g, params, case["ocean"], atmos, nthreads=nt, scales=False, out=fl)
        t2 = time.perf_counter(); orc.compute_net_ocean_fluxes(g, params, case["ocean"], atmos, fl, weights=case["weights"], out=net)
        t3 = time.perf_counter()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1), min(best[2], t3 - t2)]
    print(nt, "threads: interp %.1f ms, solver %.1f ms, net %.1f ms" % tuple(1e3 * b for b in best), flush=True)
