"""One fuzz_line.py case re-run (same RNG stream) with the worst cell of every flux field against the C oracle printed:
trip counts on both sides, the friction velocity, the cell's wind.  usage: dbg_line_case.py seed case"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util
from coflux import abi, interface_computations as ic
from test_gpu_parity import run_gpu, run_oracle
from test_gpu_random_configs import random_formulation
import random
rng = np.random.default_rng(int(sys.argv[1])); want = int(sys.argv[2])
for n in range(want + 1):
    nx = int(rng.choice([1, 5, 64, 65, 200, 333, 777, 1440, int(rng.integers(1, 1500))]))
    ny = int(rng.choice([1, 3, 40, 70, 97, int(rng.integers(1, 200))]))
    h = int(rng.integers(2, 8)); with_ice = bool(rng.integers(0, 2)); fused = bool(rng.integers(0, 2))
    wkind = str(rng.choice(["latlon", "tripolar"]))
    pat = rng.choice(["as_is", "speckle", "stripes", "all_ocean", "half"])
    if n == want: case = util.build_case(nx, ny, h, h, weights=wkind); m = case["ocean"]["mask"]
    else: m = np.zeros((ny + 2 * h, nx + 2 * h))
    if pat == "speckle": m[...] = (rng.random(m.shape) < rng.choice([0.05, 0.5, 0.95])).astype(m.dtype)
    elif pat == "stripes": m[...] = ((np.arange(m.shape[1])[None, :] // int(rng.integers(1, 90))) % 2).astype(m.dtype)
    elif pat == "all_ocean": m[...] = 1
    elif pat == "half": m[...] = 1; m[:, : m.shape[1] // 2] = 0
    f, vd, extra = random_formulation(random.Random(int(rng.integers(0, 1 << 30))))
print(dict(nx=nx, ny=ny, h=h, pat=str(pat), w=wkind, ice=with_ice, fused=fused)); print(f, vd, extra)
params = ic.flux_params(f, velocity_difference=vd, **extra)
print("solver specialization / stop:", params.maxiter, params.tolerance if hasattr(params, "tolerance") else None)
got = run_gpu(case, params, ice=with_ice, fused=fused)
ref = run_oracle(case, params, ice=with_ice)
W = lambda a: util.window(a, h, h, nx, ny, 1)
it_g, it_r = W(got["fluxes"]["iterations"]), W(ref["fluxes"]["iterations"])
print("cells with different trip counts:", int((it_g != it_r).sum()), "of", it_g.size, "; oracle max trips", it_r.max())
for k in ("sensible_heat", "latent_heat", "x_momentum", "friction_velocity"):
    if k not in got["fluxes"]: continue
    a, b = W(got["fluxes"][k]), W(ref["fluxes"][k])
    err = np.abs(a - b) / np.maximum(np.abs(b), util.FIELD_SCALE[k])
    j, i = np.unravel_index(np.argmax(err), err.shape)
    ua, va = W(got["atmos"]["u"])[j, i], W(got["atmos"]["v"])[j, i]
    print(k, "worst %.3e at" % err.max(), (j, i), "gpu %.15e ref %.15e" % (a[j, i], b[j, i]), "trips", it_g[j, i], it_r[j, i],
          "u* %.6e" % W(ref["fluxes"]["friction_velocity"])[j, i], "wind %.4f %.4f" % (ua, va),
          "count > 1e-9:", int((err > 1e-9).sum()), "> 1e-10:", int((err > 1e-10).sum()))
