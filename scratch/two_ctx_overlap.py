"""Do the ocean solve and the atmosphere–sea-ice interface solve of config 3 overlap at all when NOTHING orders them?
Two contexts on one device (two streams, no events between them): A launches compute_atmosphere_ocean_fluxes, B launches
compute_atmosphere_sea_ice_fluxes, on the bench's 1440×560 sea-ice state.  Times N launches of A alone, of B alone, and of
both issued alternately (wall time until both streams are idle).  sum ⇒ the hardware runs them one after the other;
max ⇒ the ocean solve fits into the interface solve's shadow."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
params = ic.flux_params(ic.SimilarityTheoryFluxes(), ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))


def make():
    ctx = FluxContext(nx, ny, h, h, params, ring=1)
    ctx.set_sea_ice_formulation(ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes()))
    return ctx


A, B = make(), make()
o = syn.ocean_state(nx, ny, h, h)
si = syn.sea_ice_state(nx, ny, h, h)
src = {k: A.to_device(v) for k, v in syn.jra55_snapshots(2).items()}
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
w = dict(separable=True, fi=A.to_device(fi), fj=A.to_device(fj), latitude=A.to_device(phi))
st = {k: A.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")}
at = A.field_set(EXCHANGE_NAMES)
A.interpolate_atmosphere_state(src, w, at, level1=0, level2=1, time_fraction=0.37)
A.sync()
flA, flB = A.field_set(FLUX_NAMES), B.field_set(FLUX_NAMES)
ice_state = dict(concentration=A.to_device(o["ice_concentration"]), **{k: A.to_device(si[k]) for k in ("thickness", "top_temperature", "u", "v", "albedo")})
flB["temperature"].copy_(ice_state["top_temperature"])
ice_state["top_temperature"] = flB["temperature"]   # the skin temperature carried from launch to launch, as in the bench


def run_a(n):
    for _ in range(n):
        A.compute_atmosphere_ocean_fluxes(st, at, flA)


def run_b(n):
    for _ in range(n):
        B.compute_atmosphere_sea_ice_fluxes(ice_state, st, at, flB)


def timed(f, n):
    A.sync(); B.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    f(n)
    A.sync(); B.sync()
    return (time.perf_counter() - t0) / n * 1e6


for f in (run_a, run_b):
    f(30)
A.sync(); B.sync()
n = 200
ta = min(timed(run_a, n) for _ in range(3))
tb = min(timed(run_b, n) for _ in range(3))


def both(n):
    for _ in range(n):
        B.compute_atmosphere_sea_ice_fluxes(ice_state, st, at, flB)
        A.compute_atmosphere_ocean_fluxes(st, at, flA)


tab = min(timed(both, n) for _ in range(3))
print(f"ocean solve alone {ta:.1f} us, interface solve alone {tb:.1f} us, sum {ta + tb:.1f}; both streams together {tab:.1f} us per pair")
