"""GPU fuzz, part 8 (round 5): the latency-layout kernels (csrc/coflux_solver_slab.hip, re-scheduled after register
allocation by csrc/tools/gcn_sched.py) forced on (CF_OPT_LATENCY_LAYOUT = 2) against the production kernels
(CF_OPT_LATENCY_LAYOUT = 0) — every output bitwise — on random surface sizes, halos, mask patterns, formulations (both
similarity profiles, random roughness / gustiness / stop rules) with and without the fused net fluxes and sea-ice fields;
every fourth case also against the C oracle.  usage: fuzz_line.py [seed] [cases]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util
from coflux import abi, interface_computations as ic
from test_gpu_parity import run_gpu, run_oracle, compare
from test_gpu_random_configs import random_formulation
import random
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 21)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = took = 0
for n in range(ncases):
    nx = int(rng.choice([1, 5, 64, 65, 200, 333, 777, 1440, int(rng.integers(1, 1500))]))
    ny = int(rng.choice([1, 3, 40, 70, 97, int(rng.integers(1, 200))]))
    h = int(rng.integers(2, 8)); with_ice = bool(rng.integers(0, 2)); fused = bool(rng.integers(0, 2))
    wkind = str(rng.choice(["latlon", "tripolar"]))
    case = util.build_case(nx, ny, h, h, weights=wkind)
    m = case["ocean"]["mask"]
    pat = rng.choice(["as_is", "speckle", "stripes", "all_ocean", "half"])
    if pat == "speckle": m[...] = (rng.random(m.shape) < rng.choice([0.05, 0.5, 0.95])).astype(m.dtype)
    elif pat == "stripes": m[...] = ((np.arange(m.shape[1])[None, :] // int(rng.integers(1, 90))) % 2).astype(m.dtype)
    elif pat == "all_ocean": m[...] = 1
    elif pat == "half": m[...] = 1; m[:, : m.shape[1] // 2] = 0
    f, vd, extra = random_formulation(random.Random(int(rng.integers(0, 1 << 30))))
    try:
        params = ic.flux_params(f, velocity_difference=vd, **extra)
        a = run_gpu(case, params, options=((abi.OPT_LATENCY_LAYOUT, 0),), ice=with_ice, fused=fused)
        b = run_gpu(case, params, options=((abi.OPT_LATENCY_LAYOUT, 2),), ice=with_ice, fused=fused)
        for grp in ("atmos", "fluxes", "net"):
            for k in a[grp]:
                assert np.array_equal(a[grp][k].view(np.uint8), b[grp][k].view(np.uint8)), (grp, k)
        if n % 4 == 0 and nx * ny <= 120000:
            compare(case, b, run_oracle(case, params, ice=with_ice), 1, maxiter=params.maxiter)   # (cells the oracle itself leaves at the cap: 1e-6)
    except Exception as exc:
        bad += 1
        print("FAIL", n, dict(nx=nx, ny=ny, h=h, pat=str(pat), w=wkind, ice=with_ice, fused=fused, f=repr(f)[:200]), repr(exc)[:400], flush=True)
print(f"{ncases - bad} of {ncases} cases passed", flush=True)
