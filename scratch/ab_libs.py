"""Same-box A/B of library builds (scratch): times the ocean solver stage (identical inputs, warm clocks) and one
evolving-input schedule for every scratch/libcoflux_<tag>.so named (tag 'prod' = the production library), two rounds,
alternating.  usage: ab_libs.py tag [tag ...]   env: CONFIG=default|corrected, STEPS"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
ROOT = os.environ["AB_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
cfg = os.environ.get("CONFIG", "default")
fl = ic.corrected_atmosphere_ocean_fluxes() if cfg == "corrected" else (ic.ncar_atmosphere_ocean_fluxes() if cfg == "ncar" else ic.SimilarityTheoryFluxes())
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
ctx = FluxContext(nx, ny, h, h, ic.flux_params(fl))
for k, v in json.loads(os.environ.get("AB_OPTS", "{}")).items(): ctx.set_option(int(k), int(v))
ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
src = {k: ctx.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
kw = dict(src=src, weights=w, ocean=ocean, atmos=atmos, fluxes=fluxes, net=net, time_fraction=0.37)
for _ in range(4): ctx.time_stage(abi.STAGE_AO_FLUXES, 500, **kw)
for _ in range(int(os.environ.get("AB_REBALANCE", "0"))):   # the trip-weighted chunk table: cut from what the launches above counted
    ctx.ensure_chunk_table(ocean["mask"])
    ctx.time_stage(abi.STAGE_AO_FLUXES, 200, **kw)
ao = min(ctx.time_stage(abi.STAGE_AO_FLUXES, 100, **kw) for _ in range(5))
# evolving inputs: the time fraction moves by 20 min / 3 h per call (interpolation + solver per step, events around the solver only)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * 60)]
tf = 0.0
times = []
for n in range(60):
    tf = (tf + 20.0 / 180.0) % 1.0
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, tf)
    ctx.sync()
    t = ctx.time_stage(abi.STAGE_AO_FLUXES, 1, **kw)
    if n >= 10: times.append(t)
step = min(ctx.time_stage(abi.STAGE_UPDATE_STATE, 100, **kw) for _ in range(3))
print(json.dumps(dict(ao_us=round(ao * 1e3, 2), ao_evolving_us=round(float(np.median(times)) * 1e3, 2), update_state_us=round(step * 1e3, 2))))
ctx.close()
'''
tags = sys.argv[1:] or ["prod"]
best = {}
for rnd in range(2):
    for spec in tags:
        tag, _, opts = spec.partition(":")
        env = dict(os.environ, AB_ROOT=ROOT, COFLUX_EXPERIMENTS="1")
        if tag != "prod": env["LIBCOFLUX"] = os.path.join(ROOT, "scratch", f"libcoflux_{tag}.so")
        opts, _, layers = opts.partition("@")
        if opts: env["AB_OPTS"] = json.dumps(dict(kv.split("=") for kv in opts.split(",")))
        if layers.startswith("r"): env["AB_REBALANCE"] = layers[1:]
        elif layers.startswith("w"): env["COFLUX_SORT_WINDOWS"] = layers[1:]
        elif layers: env["COFLUX_LAYERS"] = layers
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(spec, "FAILED", out.stderr[-800:]); continue
        d = json.loads(line[-1])
        b = best.setdefault(spec, d)
        for k in d: b[k] = min(b[k], d[k])
        print("round", rnd, spec, json.dumps(d), flush=True)
print("best:", json.dumps(best))
