"""GPU fuzz, part 3: the wet mask rewritten IN PLACE between calls (same pointer: the static lists are stale) on random
sizes, patterns and chunk plans; every call must give the oracle's answer for the mask as it is now."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import util, oracle as orc
from coflux import abi, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, FluxContext
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 25
def pattern(shape, dtype):
    kind = rng.choice(["speckle", "stripes", "all_land", "all_ocean", "rows", "block"])
    m = np.zeros(shape, dtype)
    if kind == "speckle": m[...] = rng.random(shape) < rng.choice([0.03, 0.3, 0.7, 0.97])
    elif kind == "stripes": m[...] = (np.arange(shape[1])[None, :] // int(rng.integers(1, 200))) % 2
    elif kind == "all_ocean": m[...] = 1
    elif kind == "rows": m[:: int(rng.integers(1, 9)), :] = 1
    elif kind == "block": m[shape[0] // 3:, shape[1] // 4: shape[1] // 4 * 3] = 1
    return m, str(kind)
bad = 0
for n in range(ncases):
    nx = int(rng.choice([64, 300, 777, 1440, int(rng.integers(2, 1500))])); ny = int(rng.choice([8, 40, 64, 141, int(rng.integers(2, 300))])); h = int(rng.integers(2, 5))
    params = ic.flux_params()
    case = util.build_case(nx, ny, h, h)
    ctx = FluxContext(nx, ny, h, h, params)
    opt = [None, 256, 1280, 768][int(rng.integers(0, 4))]
    if opt: ctx.set_option(abi.OPT_AO_CHUNK, opt)
    dev = ctx.to_device
    src = {k: dev(v) for k, v in case["src"].items()}
    w = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in case["weights"].items()}
    atmos = ctx.field_set(EXCHANGE_NAMES)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    ocean = {k: dev(case["ocean"][k]) for k in ("T", "S", "u", "v", "mask")}
    g = orc.make_grid(nx, ny, h, h, 1)
    at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
    seq = []
    try:
        for step in range(4):
            m, kind = pattern(case["ocean"]["mask"].shape, case["ocean"]["mask"].dtype); seq.append(kind)
            ocean["mask"].copy_(torch.from_numpy(m))
            fluxes = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL); fluxes["iterations"] = ctx.zeros(torch.int32)
            for rep in range(2):   # twice: the second call runs with hints written through a possibly stale list
                ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
            torch.cuda.synchronize()
            ref = orc.compute_atmosphere_ocean_fluxes(g, params, dict(case["ocean"], mask=m), at, nthreads=0)
            # the stop rule is a threshold: a cell whose drift lands within rounding of 1e-8 may stop one iteration apart from the
            # oracle (seen once in 211 452 cells, round 4: 5e-9 in the fluxes).  Policy: such cells are ≤ 1e-4 of the surface,
            # one trip apart, within 1e-7; every other cell within 1e-9.
            it_g, it_r = util.window(fluxes["iterations"].cpu().numpy(), h, h, nx, ny, 1), util.window(ref["iterations"], h, h, nx, ny, 1)
            same = it_g == it_r
            assert (~same).sum() <= max(1, int(1e-4 * same.size)) and np.abs(it_g - it_r).max() <= 1, ("trip counts", int((~same).sum()))
            for k in FLUX_NAMES + FLUX_OPTIONAL:
                a, b = util.window(fluxes[k].cpu().numpy(), h, h, nx, ny, 1), util.window(ref[k], h, h, nx, ny, 1)
                err = np.abs(a - b) / np.maximum(np.abs(b), util.FIELD_SCALE[k])
                assert err[same].max(initial=0.0) <= 1e-9 and err.max(initial=0.0) <= 1e-7, (k, float(err.max()))
    except Exception as exc:
        bad += 1
        print("FAIL", n, dict(nx=nx, ny=ny, h=h, opt=opt, seq=seq), repr(exc)[:300], flush=True)
    ctx.close()
print(f"{ncases - bad} of {ncases} cases passed", flush=True)
