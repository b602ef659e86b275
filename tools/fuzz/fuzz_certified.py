"""GPU fuzz, part 7 (round 5): the certified solver path on random surface sizes, halos, rings, mask patterns and kinds,
chunk plans, fusion and pipelining options — against the C oracle's exact path with tests/test_certified.py's criteria
(six fields and net fluxes <= 1e-6, exact-path cells 1e-9 with the reference's trip counts, land exact), and bitwise
against the same case under another chunk plan."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util
from coflux import abi, interface_computations as ic
from test_gpu_parity import run_gpu, run_oracle
from test_certified import compare_certified, CERTIFIED, certifiable_formulation
import random
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for n in range(ncases):
    nx = int(rng.choice([1, 2, 5, 63, 64, 65, 127, 200, 333, 777, int(rng.integers(1, 1500))]))
    ny = int(rng.choice([1, 2, 3, 7, 40, 97, int(rng.integers(1, 300))]))
    ring = 1; h = int(rng.integers(2, 6))
    wkind = str(rng.choice(["latlon", "tripolar"])); with_ice = bool(rng.integers(0, 2)); fused = bool(rng.integers(0, 2))
    case = util.build_case(nx, ny, h, h, weights=wkind)
    m = case["ocean"]["mask"]
    pat = rng.choice(["as_is", "speckle", "stripes", "all_ocean", "single", "half"])
    if pat == "speckle": m[...] = (rng.random(m.shape) < rng.choice([0.05, 0.5, 0.95])).astype(m.dtype)
    elif pat == "stripes": m[...] = ((np.arange(m.shape[1])[None, :] // int(rng.integers(1, 90))) % 2).astype(m.dtype)
    elif pat == "all_ocean": m[...] = 1
    elif pat == "single": m[...] = 0; m[h + int(rng.integers(0, ny)), h + int(rng.integers(0, nx))] = 1
    elif pat == "half": m[...] = 1; m[:, : m.shape[1] // 2] = 0
    f, vd, extra = certifiable_formulation(random.Random(int(rng.integers(0, 1 << 30))))
    params = ic.flux_params(f, velocity_difference=vd, **extra)
    plans = [(), ((abi.OPT_AO_CHUNK, 256),), ((abi.OPT_AO_CHUNK, 512),), ((abi.OPT_AO_CHUNK, 1280),), ((abi.OPT_MERGED_PREFETCH, 2),)]
    i, j = rng.choice(len(plans), 2, replace=False)
    try:
        got = run_gpu(case, params, ring=ring, options=CERTIFIED + plans[i], ice=with_ice, fused=fused)
        ref = run_oracle(case, params, ring=ring, ice=with_ice)
        if np.any(util.window(ref["fluxes"]["iterations"], h, h, nx, ny, 1) >= params.maxiter):
            continue
        compare_certified(case, got, ref, expect_certified=False, max_exact_share=1.0, label=f"fuzz {n}")
        other = run_gpu(case, params, ring=ring, options=CERTIFIED + plans[j], ice=with_ice, fused=fused)
        for k in got["fluxes"]:
            np.testing.assert_array_equal(got["fluxes"][k], other["fluxes"][k], err_msg=k)
    except Exception as exc:
        bad += 1
        print("FAIL", n, dict(nx=nx, ny=ny, h=h, pat=str(pat), plans=(plans[i], plans[j]), w=wkind, ice=with_ice, fused=fused), repr(exc)[:400], flush=True)
print(f"{ncases - bad} of {ncases} cases passed", flush=True)
