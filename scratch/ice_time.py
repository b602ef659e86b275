"""Timing of the atmosphere–sea-ice interface kernel on the 1/4° surface (not part of bench.py)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import util
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, FluxContext
nx, ny, h = 1440, 560, 7
case = util.build_case(nx, ny, h, h)
ctx = FluxContext(nx, ny, h, h, ic.flux_params())
dev = ctx.to_device
src = {k: dev(v) for k, v in case["src"].items()}
w = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in case["weights"].items()}
atmos = ctx.field_set(EXCHANGE_NAMES)
ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37); ctx.sync()
at = util.polar_atmosphere({k: v.cpu().numpy() for k, v in atmos.items()})
atmos = {k: dev(at[k]) for k in EXCHANGE_NAMES}
ocean = {k: dev(case["ocean"][k]) for k in ("T", "S", "u", "v", "mask")}
st = {k: dev(v) for k, v in case["ice_state"].items()}
for name in ("sea_ice_corrected", "sea_ice_ncar", "sea_ice_fixed5"):
    f, vd = util.ICE_CONFIGS[name]()
    ctx.set_sea_ice_formulation(ic.flux_params(f, velocity_difference=vd), ic.SeaIceInterfaceProperties().to_params())
    out = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL); out["iterations"] = ctx.zeros(torch.int32)
    for _ in range(3): ctx.compute_atmosphere_sea_ice_fluxes(st, ocean, atmos, out)
    ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ctx.compute_atmosphere_sea_ice_fluxes(st, ocean, atmos, out)
    ctx.sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    it = out["iterations"].cpu().numpy(); wet = case["ocean"]["mask"] != 0
    print(f"{name}: {dt*1e6:.1f} us per launch; mean iterations over wet cells {it[wet].mean():.1f}, at maxiter {np.mean(it[wet] >= 100)*100:.0f} %")
ctx.close()
