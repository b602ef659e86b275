"""Which ψ path the waves of the solver take (needs scratch/libcoflux_count.so: make_variant_build.sh count -DCF_EXP_COUNT)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
for label in ("default", "fixed32"):
    fl = ic.SimilarityTheoryFluxes()
    if label != "default": fl.solver_stop_criteria = ic.FixedIterations(32)
    ctx = FluxContext(nx, ny, h, h, ic.flux_params(fl))
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    for _ in range(3): ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
    out = (C.c_ulonglong * 64)()
    ctx.lib.cf_debug_counts_read(out, 1)
    ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
    ctx.lib.cf_debug_counts_read(out, 1)
    a = np.array(out[:], dtype=np.int64)
    print(label, "wave-iterations: all-small %d, all-table %d, mixed %d; lanes total %d" % (a[0], a[1], a[2], a[40]))
    print("   table-path lanes by iteration:", a[8:40].tolist())
    ctx.close()
