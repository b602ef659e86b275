"""One fuzz_stale_mask case re-run with the worst cell printed (scratch)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import util, oracle as orc
from coflux import abi, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, FluxContext
nx, ny, h, opt = 802, 261, 2, 1280
params = ic.flux_params()
case = util.build_case(nx, ny, h, h)
for opt in (1280, None):
    ctx = FluxContext(nx, ny, h, h, params)
    if opt: ctx.set_option(abi.OPT_AO_CHUNK, opt)
    dev = ctx.to_device
    src = {k: dev(v) for k, v in case["src"].items()}
    w = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in case["weights"].items()}
    atmos = ctx.field_set(EXCHANGE_NAMES)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    ocean = {k: dev(case["ocean"][k]) for k in ("T", "S", "u", "v", "mask")}
    g = orc.make_grid(nx, ny, h, h, 1)
    at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
    m = np.ones_like(case["ocean"]["mask"])
    ocean["mask"].copy_(torch.from_numpy(m))
    fluxes = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL); fluxes["iterations"] = ctx.zeros(torch.int32)
    ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes); torch.cuda.synchronize()
    ref = orc.compute_atmosphere_ocean_fluxes(g, params, dict(case["ocean"], mask=m), at, nthreads=0)
    W = lambda a: util.window(a, h, h, nx, ny, 1)
    it_g, it_r = W(fluxes["iterations"].cpu().numpy()), W(ref["iterations"])
    print("opt", opt, "cells with different trip counts:", int((it_g != it_r).sum()), "of", it_g.size)
    for k in ("sensible_heat", "latent_heat", "x_momentum"):
        a, b = W(fluxes[k].cpu().numpy()), W(ref[k])
        err = np.abs(a - b) / np.maximum(np.abs(b), util.FIELD_SCALE[k])
        j, i = np.unravel_index(np.argmax(err), err.shape)
        print(" ", k, "worst", err.max(), "at", (j, i), "gpu", a[j, i], "ref", b[j, i], "trips gpu/ref", it_g[j, i], it_r[j, i],
              "ustar", W(fluxes["friction_velocity"].cpu().numpy())[j, i] if "friction_velocity" in fluxes else None)
        same = it_g == it_r
        print("    worst among equal-trip cells:", err[same].max())
    ctx.close()
