"""Per-wave time stamps of the solver's phases (needs scratch/libcoflux_phase.so built with the STAMP instrumentation)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
for label in ("default", "fixed0", "fixed12"):
    fl = ic.SimilarityTheoryFluxes()
    if label != "default": fl.solver_stop_criteria = ic.FixedIterations(0 if label == "fixed0" else 12)
    ctx = FluxContext(nx, ny, h, h, ic.flux_params(fl))
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    for _ in range(5): ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
    ms = ctx.time_stage(abi.STAGE_AO_FLUXES, 20, ocean=ocean, atmos=atmos, fluxes=fluxes)
    ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
    NWG = int(os.environ.get('NWG', '752'))
    n = NWG * 4 * 8
    out = (C.c_ulonglong * n)()
    ctx.lib.cf_debug_phase_read(out, n)
    raw = np.array(out, dtype=np.float64).reshape(NWG * 4, 8)
    tick = (raw[:, 6] - raw[:, 0]).max() / (ms * 1e3)    # ticks per µs, assuming the longest-lived wave spans the kernel
    print(f"{label}: kernel {ms*1e3:.1f} us; {tick:.1f} ticks/us")
    st = raw[:, [0, 1, 2, 3, 6]]
    d = np.diff(st, axis=1)
    total = st[:, 4] - st[:, 0]
    for q, name in enumerate(("entry -> first barrier (everything requested)", "sort + validation + table DMA landed", "batches (load/prologue/iterate/store)", "(unused)")):
        a_ = d[:, q] / tick
        print(f"   {name:52s} min {a_.min():6.1f}  median {np.median(a_):6.1f}  p90 {np.percentile(a_,90):6.1f}  max {a_.max():6.1f} us")
    for nm, v in (("   entry -> table DMA issued", raw[:, 4] - raw[:, 0]), ("   DMA issued -> list + mask loads issued", raw[:, 5] - raw[:, 4]), ("   wait lgkm + barrier", raw[:, 1] - raw[:, 5])):
        a_ = v / tick
        print(f"   {nm:52s} min {a_.min():6.1f}  median {np.median(a_):6.1f}  p90 {np.percentile(a_,90):6.1f}  max {a_.max():6.1f} us")
    # the tail: how many waves are still alive in the kernel's last microseconds
    t_end = st[:, 4].max(); t_first = st[:, 0].min()
    alive = [(k, float(((st[:, 4] > t_end - k * tick)).mean())) for k in (1, 2, 4, 6, 8, 10, 15, 20)]
    print("   waves still running k us before the end: " + "  ".join(f"{k}us {f*100:.0f}%" for k, f in alive) + f"   (first entry -> last exit {(t_end - t_first)/tick:.1f} us)")
    a_ = total / tick
    print(f"   {'wave lifetime':40s} min {a_.min():6.1f}  median {np.median(a_):6.1f}  p90 {np.percentile(a_,90):6.1f}  max {a_.max():6.1f} us")
    wg_life = (st[:, 4].reshape(NWG, 4).max(axis=1) - st[:, 0].reshape(NWG, 4).min(axis=1)) / tick
    wave_b = d[:, 2].reshape(NWG, 4) / tick
    print("   WG lifetime by XCD (blockIdx % 8): " + " ".join(f"{np.median(wg_life[x::8]):.0f}/{wg_life[x::8].max():.0f}" for x in range(8)))
    order = np.arange(NWG)
    for lo, hi in ((0, 256), (256, 512), (512, NWG)):
        sel = (order >= lo) & (order < hi)
        print(f"   blockIdx {lo:3d}-{hi:3d}: WG lifetime median {np.median(wg_life[sel]):.1f} max {wg_life[sel].max():.1f}; wave batch-phase spread within WG (max-min) median {np.median(wave_b[sel].max(axis=1)-wave_b[sel].min(axis=1)):.1f}")
    # start skew: when did each WG start relative to the earliest WG of its XCD
    s0 = st[:, 0].reshape(NWG, 4).min(axis=1)
    for x in range(2):
        rel = (s0[x::8] - s0[x::8].min()) / tick
        print(f"   XCD {x}: WG start offsets median {np.median(rel):.1f} p90 {np.percentile(rel,90):.1f} max {rel.max():.1f} us")
    ctx.close()
