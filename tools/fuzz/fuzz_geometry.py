"""GPU fuzz (not a test: minutes of oracle time): random surface sizes, halos, rings, mask patterns (random speckle, land
stripes, all land, all ocean, a single wet cell), mask kinds and chunk plans — the HIP solver + net fluxes against the
C oracle, identical trip counts, exact zeros on land.  Exercises the start phase's range/list/fingerprint logic on
ragged geometries."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util
from coflux import abi, interface_computations as ic
from test_gpu_parity import run_gpu, run_oracle, compare
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for n in range(ncases):
    nx = int(rng.choice([1, 2, 5, 63, 64, 65, 127, 200, 333, 777, int(rng.integers(1, 1500))]))
    ny = int(rng.choice([1, 2, 3, 7, 40, 97, int(rng.integers(1, 300))]))
    ring = int(rng.integers(0, 2)); h = int(rng.integers(ring + 1, 6))
    wkind = str(rng.choice(["latlon", "tripolar"])); with_ice = bool(rng.integers(0, 2)); fused = bool(rng.integers(0, 2))
    case = util.build_case(nx, ny, h, h, weights=wkind)
    m = case["ocean"]["mask"]
    pat = rng.choice(["as_is", "speckle", "stripes", "all_land", "all_ocean", "single", "half"])
    if pat == "speckle": m[...] = (rng.random(m.shape) < rng.choice([0.05, 0.5, 0.95])).astype(m.dtype)
    elif pat == "stripes": m[...] = ((np.arange(m.shape[1])[None, :] // int(rng.integers(1, 90))) % 2).astype(m.dtype)
    elif pat == "all_land": m[...] = 0
    elif pat == "all_ocean": m[...] = 1
    elif pat == "single": m[...] = 0; m[h + int(rng.integers(0, ny)), h + int(rng.integers(0, nx))] = 1
    elif pat == "half": m[...] = 1; m[:, : m.shape[1] // 2] = 0
    kind = rng.choice(["u8", "bottom", "none"])
    params = ic.flux_params(mask_kind={"u8": abi.MASK_U8, "bottom": abi.MASK_BOTTOM_HEIGHT, "none": abi.MASK_NONE}[kind])
    if kind == "bottom": case["ocean"]["mask"] = np.where(m != 0, -3000.0, 10.0)
    opts = [(), ((abi.OPT_AO_CHUNK, 256),), ((abi.OPT_AO_CHUNK, 1280),), ((abi.OPT_AO_CHUNK, 768),), ((abi.OPT_TRIP_HINTS, 0),)][int(rng.integers(0, 5))]
    try:
        got = run_gpu(case, params, ring=ring, options=opts, ice=with_ice, fused=fused)
        ref = run_oracle(case, params, ring=ring, ice=with_ice)
        compare(case, got, ref, ring)
        W = lambda a: util.window(a, h, h, nx, ny, ring)
        np.testing.assert_array_equal(W(got["fluxes"]["iterations"]), W(ref["fluxes"]["iterations"]))
    except Exception as exc:
        bad += 1
        print("FAIL", n, dict(nx=nx, ny=ny, h=h, ring=ring, pat=str(pat), kind=str(kind), opts=opts, w=wkind, ice=with_ice, fused=fused), repr(exc)[:300], flush=True)
print(f"{ncases - bad} of {ncases} cases passed", flush=True)
