"""How stable is the certified path's set of exact-path cells from one step to the next? (scratch study for the
"known exact-path cells first" list order.)  Runs steps s = 0..S-1 of the bench's surface (two ocean states one step
apart, the atmosphere advancing DT/3 h per step), reads `iterations`, and counts — per group of 752 consecutive wet cells
(a workgroup's share) — the cells that go down the exact path now but did not one step earlier."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext

nx, ny, h = 1440, 560, 3
S = int(os.environ.get("STEPS", "6"))
params = ic.flux_params(ic.SimilarityTheoryFluxes())
src_np = syn.jra55_snapshots(4, temporal_correlation=0.95)
first = syn.ocean_state(nx, ny, h, h)
second = syn.evolved_ocean_state(first, nx, ny, h, h, 1)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
ctx = FluxContext(nx, ny, h, h, params, ring=1, device=0)
ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED)
states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in (first, second)]
states[1]["mask"] = states[0]["mask"]
src = {k: ctx.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
atmos, fl, net = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES[:5])
fl["iterations"] = torch.zeros_like(fl["temperature"], dtype=torch.int32)
inc = 1200.0 / 10800.0
flags = []
for s in range(S):
    tot = s * inc
    l1 = int(tot) % 4
    ctx.update_state(src, w, states[s % 2], atmos, fl, net, level1=l1, level2=(l1 + 1) % 4, time_fraction=tot - int(tot))
    ctx.sync()
    it = fl["iterations"].cpu().numpy()[h - 1:h + ny + 1, h - 1:h + nx + 1].ravel()
    flags.append((it > 0, (it & abi.CERTIFIED_EXACT_FLAG) != 0))
GROUP = int(os.environ.get("GROUP", "752"))
for s in range(1, S):
    wet, ex = flags[s]
    _, prev = flags[s - 1]
    new = ex & ~prev
    gone = prev & ~ex
    idx = np.flatnonzero(wet)
    g = np.arange(idx.size) // GROUP
    ng = g.max() + 1
    new_g = np.bincount(g, weights=new[idx], minlength=ng)
    ex_g = np.bincount(g, weights=ex[idx], minlength=ng)
    print(json.dumps(dict(step=s, exact_share=float(ex[idx].mean()), new_share=float(new[idx].mean()), gone_share=float(gone[idx].mean()),
                          groups=int(ng), groups_without_new=float((new_g == 0).mean()), groups_without_exact=float((ex_g == 0).mean()),
                          mean_new_per_group=float(new_g.mean()), mean_exact_per_group=float(ex_g.mean()),
                          p90_exact_per_group=float(np.percentile(ex_g, 90)), max_exact_per_group=float(ex_g.max()))))
