"""GPU fuzz, part 5: cf_time_steps (C loop, advancing clock, alternating ocean states, split into several calls, with and
without the pipelined interpolation) against the host-driven cf_update_state loop — bitwise — on random sizes, window
lengths, clock increments and step counts."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for n in range(ncases):
    nx = int(rng.choice([33, 192, 640, int(rng.integers(2, 1000))])); ny = int(rng.choice([5, 48, 141, int(rng.integers(2, 200))])); h = int(rng.integers(2, 6))
    n_levels = int(rng.integers(2, 6)); inc = float(rng.choice([1 / 9, 1 / 3, 0.5, 0.37, 1.0])); nsteps = int(rng.integers(1, 30)); pipeline = bool(rng.integers(0, 2))
    cfg = [ic.SimilarityTheoryFluxes, ic.corrected_atmosphere_ocean_fluxes, ic.ncar_atmosphere_ocean_fluxes][int(rng.integers(0, 3))]
    try:
        ctx = FluxContext(nx, ny, h, h, ic.flux_params(cfg()), ring=1)
        o0 = syn.ocean_state(nx, ny, h, h); o1 = syn.evolved_ocean_state(o0, nx, ny, h, h, 1)
        states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in (o0, o1)]
        states[1]["mask"] = states[0]["mask"]
        src = {k: ctx.to_device(v) for k, v in syn.jra55_snapshots(n_levels).items()}
        fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
        w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
        ra, rf, rn = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
        tf0 = float(rng.random())
        for s in range(nsteps):
            tot = tf0 + s * inc
            l1 = int(tot) % n_levels
            ctx.update_state(src, w, states[s % 2], ra, rf, rn, level1=l1, level2=(l1 + 1) % n_levels, time_fraction=tot - int(tot))
        ctx.sync()
        merged = int(rng.integers(0, 3)) if pipeline else 0   # CF_OPT_MERGED_PREFETCH: aux stream / stress launch / solver tail
        ctx.set_option(abi.OPT_MERGED_PREFETCH, merged)
        sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2 if pipeline else 1)]
        fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
        sched = ctx.make_schedule(states, sets, first_level=0, time_fraction=tf0, time_fraction_increment=inc, pipeline=pipeline)
        cut = int(rng.integers(0, nsteps + 1))
        if cut: ctx.time_steps(0, cut, sched, src, w, fl, net)
        if nsteps - cut: ctx.time_steps(cut, nsteps - cut, sched, src, w, fl, net)
        ctx.sync()
        last = sets[(nsteps - 1) % len(sets)]
        for k in EXCHANGE_NAMES: assert torch.equal(last[k], ra[k]), ("atmos", k)
        for k in FLUX_NAMES: assert torch.equal(fl[k], rf[k]), ("fluxes", k)
        for k in NET_NAMES: assert torch.equal(net[k], rn[k]), ("net", k)
        ctx.close()
    except Exception as exc:
        bad += 1
        print("FAIL", n, dict(nx=nx, ny=ny, h=h, n_levels=n_levels, inc=inc, nsteps=nsteps, pipeline=pipeline, merged=merged if pipeline else None, cfg=cfg.__name__), repr(exc)[:300], flush=True)
print(f"{ncases - bad} of {ncases} cases passed", flush=True)
