"""GPU fuzz, part 2: (a) interpolation on random sizes with separable and general (rotated) weights, every tile-row
variant reachable through the size and the tile cap, against the oracle at 1e-12; (b) the sea-ice interface solve with
the orbit shortcut on and off — bitwise — on random sizes, both skin schemes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util, oracle as orc
from coflux import abi, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FluxContext
from test_gpu_parity import run_ice
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
n_interp = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n_ice = int(sys.argv[3]) if len(sys.argv) > 3 else 10
bad = 0
for n in range(n_interp):
    nx = int(rng.choice([1, 3, 64, 65, 200, 640, 1440, int(rng.integers(1, 1600))])); ny = int(rng.choice([1, 2, 5, 35, 70, 141, int(rng.integers(1, 400))]))
    ring = int(rng.integers(0, 2)); h = int(rng.integers(ring + 1, 5))
    wkind = rng.choice(["latlon", "tripolar"])
    case = util.build_case(nx, ny, h, h, weights=wkind, n_levels=3)
    cap = int(rng.choice([128, 16, 224, 0, 64])); tf = float(rng.random()); l1, l2 = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    try:
        g = orc.make_grid(nx, ny, h, h, ring)
        ref = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], l1, l2, tf)
        ctx = FluxContext(nx, ny, h, h, ic.flux_params(), ring=ring)
        ctx.set_option(abi.OPT_INTERP_TILE_CAP, cap)
        src = {k: ctx.to_device(v) for k, v in case["src"].items()}
        w = {k: (ctx.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in case["weights"].items()}
        at = ctx.field_set(EXCHANGE_NAMES)
        ctx.interpolate_atmosphere_state(src, w, at, l1, l2, tf); ctx.sync()
        for k in EXCHANGE_NAMES:
            e = util.rel_err(util.window(at[k].cpu().numpy(), h, h, nx, ny, ring), util.window(ref[k], h, h, nx, ny, ring), util.ATMOS_SCALE[k])
            assert e < 1e-12, (k, e)
        ctx.close()
    except Exception as exc:
        bad += 1
        print("FAIL interp", n, dict(nx=nx, ny=ny, h=h, ring=ring, w=str(wkind), cap=cap), repr(exc)[:300], flush=True)
for n in range(n_ice):
    nx = int(rng.choice([5, 64, 90, 333, int(rng.integers(1, 700))])); ny = int(rng.choice([3, 40, 67, int(rng.integers(1, 200))]))
    cfg = str(rng.choice(["sea_ice_corrected", "sea_ice_ncar"])); scheme = int(rng.integers(0, 2))
    try:
        case = util.build_case(nx, ny)
        fast, _ = run_ice(case, cfg, scheme=scheme)
        slow, _ = run_ice(case, cfg, scheme=scheme, options=((abi.OPT_ICE_ORBIT_SHORTCUT, 0),))
        for k in fast: np.testing.assert_array_equal(fast[k], slow[k], err_msg=k)
    except Exception as exc:
        bad += 1
        print("FAIL ice", n, dict(nx=nx, ny=ny, cfg=cfg, scheme=scheme), repr(exc)[:300], flush=True)
print(f"{n_interp + n_ice - bad} of {n_interp + n_ice} cases passed", flush=True)
