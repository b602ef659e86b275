"""GPU fuzz, part 4: the sea-ice interface solve in both workgroup geometries and several chunk plans — bitwise equal
to the default plan — and against the oracle, on random sizes and both skin schemes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util
from coflux import abi
from test_gpu_parity import run_ice, TOL_ICE
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bad = 0
for n in range(ncases):
    nx = int(rng.choice([7, 64, 90, 333, 1440, int(rng.integers(1, 900))])); ny = int(rng.choice([3, 40, 67, 140, int(rng.integers(1, 250))]))
    cfg = str(rng.choice(["sea_ice_corrected", "sea_ice_ncar"])); scheme = int(rng.integers(0, 2))
    try:
        case = util.build_case(nx, ny)
        ref_gpu, ref = run_ice(case, cfg, scheme=scheme)
        try:   # (informational: slow cells of the semi-implicit variant can differ from libm by one trip; the tests pin the bars)
            util.compare_ice_fluxes(ref_gpu, ref, TOL_ICE)
        except AssertionError as exc:
            print("  note: oracle comparison outside the test bars for", dict(nx=nx, ny=ny, cfg=cfg, scheme=scheme), repr(exc)[:120], flush=True)
        for plan in (768, 256, 1280):
            got, _ = run_ice(case, cfg, scheme=scheme, options=((abi.OPT_AO_CHUNK, plan),))
            for k in got: np.testing.assert_array_equal(got[k], ref_gpu[k], err_msg=f"plan {plan} {k}")
    except Exception as exc:
        bad += 1
        print("FAIL", n, dict(nx=nx, ny=ny, cfg=cfg, scheme=scheme), repr(exc)[:300], flush=True)
print(f"{ncases - bad} of {ncases} cases passed", flush=True)
