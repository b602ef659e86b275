"""Config-3 interface solve with the skin temperature carried from step to step (as a coupled run does): reported trip
counts per step and the kernel time once the carried state has settled (scratch)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd")]
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, FluxContext
nx, ny, h = 1440, 560, 7
oc = syn.ocean_state(nx, ny, h, h); si = syn.sea_ice_state(nx, ny, h, h); src_np = syn.jra55_snapshots(4, temporal_correlation=0.95)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
ctx = FluxContext(nx, ny, h, h, ic.flux_params())
ctx.set_sea_ice_formulation(ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes()))
src = {k: ctx.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
ocean = {k: ctx.to_device(oc[k]) for k in ("T", "S", "u", "v", "mask")}
st = dict(concentration=ctx.to_device(oc["ice_concentration"]), **{k: ctx.to_device(si[k]) for k in ("thickness", "top_temperature", "u", "v", "albedo")})
out = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL); out["iterations"] = ctx.zeros(torch.int32)
if os.environ.get("CARRY", "1") == "1":
    out["temperature"].copy_(st["top_temperature"]); st["top_temperature"] = out["temperature"]
atmos = ctx.field_set(EXCHANGE_NAMES)
wet = oc["mask"][h:h + ny, h:h + nx] != 0
for step in range(40):
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, (step / 9.0) % 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    ctx.compute_atmosphere_sea_ice_fluxes(st, ocean, atmos, out)
    e1.record(); torch.cuda.synchronize()
    if step < 4 or step % 6 == 0 or step == 39:
        it = out["iterations"].cpu().numpy()[h:h + ny, h:h + nx][wet]
        print(f"step {step:2d}: {e0.elapsed_time(e1) * 1e3:6.0f} us  mean trips {it.mean():5.1f}  at maxiter {100 * (it >= 100).mean():5.2f} %  "
              f"histogram 0-9..90-99,100: {[int(((it >= a) & (it < a + 10)).sum()) for a in range(0, 100, 10)] + [int((it >= 100).sum())]}", flush=True)
ctx.close()
