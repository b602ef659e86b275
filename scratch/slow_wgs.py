"""Which workgroups end the kernel?  Lifetime vs range length / wet count / layer (needs libcoflux_phase.so)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
fl = ic.SimilarityTheoryFluxes()
if os.environ.get('FIXED'): fl.solver_stop_criteria = ic.FixedIterations(int(os.environ['FIXED']))
ctx = FluxContext(nx, ny, h, h, ic.flux_params(fl))
ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
src = {k: ctx.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES)
ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
for _ in range(5): ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
ctx.sync()
NWG = int(os.environ.get('NWG', '760'))
out = (C.c_ulonglong * (NWG * 32))()
ctx.lib.cf_debug_phase_read(out, NWG * 32)
a = np.array(out, dtype=np.uint64).reshape(NWG, 4, 8)
life = (a[:, :, 3].astype(np.float64) - a[:, :, 0].astype(np.float64)).max(axis=1) / 1950.0
cls = (a[:, :, 2].astype(np.float64) - a[:, :, 1].astype(np.float64)).max(axis=1) / 1950.0
rng = (a[:, 0, 7] >> np.uint64(32)).astype(np.int64); nwet = (a[:, 0, 7] & np.uint64(0xffffffff)).astype(np.int64)
layer = np.arange(NWG) // 256
for L in range(3):
    m = layer == L
    print(f"layer {L}: n={m.sum()} wet median {np.median(nwet[m])} range median {np.median(rng[m])} max {rng[m].max()}  lifetime median {np.median(life[m]):.1f} p90 {np.percentile(life[m],90):.1f} max {life[m].max():.1f}")
    top = np.argsort(-life * m)[:6]
    print("   slowest:", [(int(g), int(nwet[g]), int(rng[g]), round(float(cls[g]), 1), round(float(life[g]), 1)) for g in top])
    c = np.corrcoef(np.c_[life[m], rng[m], nwet[m], cls[m]].T)
    print("   corr(life, range) %.2f  corr(life, wet) %.2f  corr(life, classify time) %.2f" % (c[0, 1], c[0, 2], c[0, 3]))
ctx.close()
