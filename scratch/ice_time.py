"""Atmosphere–sea-ice interface kernel on the 1/4-degree surface under a polar atmosphere: time per launch and share of
cells at maxiter for both skin-temperature schemes (explicit / semi-implicit), convergence and FixedIterations(5)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch, ctypes as C
import util
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, FluxContext
nx, ny, h = 1440, 560, 7
case = util.build_case(nx, ny, h, h)
for scheme in (0, 1):
    for label, fl in (("convergence", ic.corrected_atmosphere_sea_ice_fluxes()), ("fixed5", util._fixed(ic.corrected_atmosphere_sea_ice_fluxes(), 5))):
        ctx = FluxContext(nx, ny, h, h, ic.flux_params())
        ctx.set_sea_ice_formulation(ic.flux_params(fl), ic.SeaIceInterfaceProperties(skin_temperature_scheme=scheme).to_params())
        src = {k: ctx.to_device(v) for k, v in case["src"].items()}
        w = {k: (ctx.to_device(v) if hasattr(v, "shape") else v) for k, v in case["weights"].items()}
        atmos = ctx.field_set(EXCHANGE_NAMES)
        ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37); ctx.sync()
        at = util.polar_atmosphere({k: v.cpu().numpy() for k, v in atmos.items()})
        atmos = {k: ctx.to_device(v) for k, v in at.items()}
        ocean = {k: ctx.to_device(case["ocean"][k]) for k in ("T", "S", "u", "v", "mask")}
        st = {k: ctx.to_device(v) for k, v in case["ice_state"].items()}
        out = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL); out["iterations"] = ctx.zeros(torch.int32)
        for _ in range(3): ctx.compute_atmosphere_sea_ice_fluxes(st, ocean, atmos, out)
        ctx.sync(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ctx.compute_atmosphere_sea_ice_fluxes(st, ocean, atmos, out)
        e1.record(); torch.cuda.synchronize()
        it = out["iterations"].cpu().numpy(); wet = case["ocean"]["mask"] != 0
        print(f"scheme {scheme} {label}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch; mean trips {it[wet].mean():.1f}; at maxiter {100 * (it[wet] >= 100).mean():.1f} %")
        ctx.close()
