"""Per-wave time stamps of the lean ocean kernel's phases (needs scratch/libcoflux_leanstamp.so, make_lean_stamp_build.sh).
usage: LIBCOFLUX=scratch/libcoflux_leanstamp.so python scratch/phases_lean.py [config]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FluxContext
nx, ny, h = 1440, int(os.environ.get('NY', 560)), 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
cfgs = sys.argv[1:] or ["default", "fixed0", "fixed12"]
for label in cfgs:
    fl = ic.corrected_atmosphere_ocean_fluxes() if label == "corrected" else ic.SimilarityTheoryFluxes()
    if label.startswith("fixed"): fl.solver_stop_criteria = ic.FixedIterations(int(label[5:]))
    ctx = FluxContext(nx, ny, h, h, ic.flux_params(fl))
    if os.environ.get("CHUNK"): ctx.set_option(abi.OPT_AO_CHUNK, int(os.environ["CHUNK"]))
    if os.environ.get("CERT"):   # certified solver path with this budget (units of 1e-9)
        ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED); ctx.set_option(abi.OPT_CERTIFIED_BUDGET, int(os.environ["CERT"]))
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    for _ in range(3): ctx.time_stage(abi.STAGE_AO_FLUXES, 500, ocean=ocean, atmos=atmos, fluxes=fluxes)
    if os.environ.get("REBAL"):
        ctx.ensure_chunk_table(ocean["mask"])
        ctx.time_stage(abi.STAGE_AO_FLUXES, 200, ocean=ocean, atmos=atmos, fluxes=fluxes)
    ms = min(ctx.time_stage(abi.STAGE_AO_FLUXES, 50, ocean=ocean, atmos=atmos, fluxes=fluxes) for _ in range(3))
    ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
    NWG = int(os.environ.get('NWG', '752')); WPG = int(os.environ.get('WPG', '4'))
    n = NWG * WPG * 8
    out = (C.c_ulonglong * n)()
    ctx.lib.cf_debug_phase_read(out, n)
    flags = (np.array(out, dtype=np.uint64).reshape(NWG * WPG, 8)[:, 7] >> np.uint64(32)).astype(int)
    print("   waves with a valid sorted list:", int((flags & 1).sum()), "of", NWG * WPG, "; sorting:", int(((flags >> 1) & 1).sum()))
    raw = np.array(out, dtype=np.float64).reshape(NWG * WPG, 8)
    raw[:, 7] = np.array(out, dtype=np.uint64).reshape(NWG * WPG, 8)[:, 7] & np.uint64(0xffffffff)
    u6 = np.array(out, dtype=np.uint64).reshape(NWG * WPG, 8)[:, 6]
    raw[:, 6] = u6 & np.uint64((1 << 40) - 1)
    trips = (u6 >> np.uint64(40)).astype(np.float64)
    print("   loop trips per batch (wave max): %.2f over %d batches" % (trips.sum() / raw[:, 7].sum(), int(raw[:, 7].sum())))
    # s_memtime bases differ between CUs: only differences within a wave (or a workgroup) are meaningful
    life = raw[:, 5] - raw[:, 0]
    tick = life.max() / (ms * 1e3)  # ticks per µs, assuming the longest-lived wave spans the kernel
    print(f"{label}: kernel {ms*1e3:.1f} us; {tick:.1f} ticks/us (longest wave = kernel)")
    D = lambda a, b: (raw[:, a] - raw[:, b]) / tick
    def row(name, a):
        print(f"   {name:46s} min {a.min():6.1f}  median {np.median(a):6.1f}  p90 {np.percentile(a,90):6.1f}  max {a.max():6.1f} us")
    row("entry -> everything requested", D(1, 0))
    row("requested -> landed (vmcnt 0)", D(2, 1))
    row("landed -> behind the barrier", D(3, 2))
    row("start phase total", D(3, 0))
    row("batches", D(4, 3))
    row("  of which inside the iteration", raw[:, 6] / tick)
    row("  batches per wave", raw[:, 7])
    row("end phase (retire / sort)", D(5, 4))
    row("wave lifetime", D(5, 0))
    for lo, hi in ((0, 256), (256, 512), (512, NWG)):
        if lo >= NWG: continue
        hi = min(hi, NWG)
        sel = np.repeat((np.arange(NWG) >= lo) & (np.arange(NWG) < hi), WPG)
        wg_life = (raw[sel, 5].reshape(-1, WPG).max(axis=1) - raw[sel, 0].reshape(-1, WPG).min(axis=1)) / tick
        nb = np.maximum(raw[sel, 7], 1)
        print(f"   blockIdx {lo:3d}-{hi:3d}: workgroup lifetime median {np.median(wg_life):.1f} p10 {np.percentile(wg_life,10):.1f} max {wg_life.max():.1f}; batches/wave {raw[sel,7].mean():.2f}; per batch {np.median(D(4,3)[sel]/nb):.1f} us of which iterating {np.median(raw[sel,6]/tick/nb):.1f}")
    # the tail: which workgroups end the kernel, and is it their work (sum of their batches' trips) or their CU's?
    wg_life_all = (raw[:, 5].reshape(-1, WPG).max(axis=1) - raw[:, 0].reshape(-1, WPG).min(axis=1)) / tick
    wg_trips = trips.reshape(-1, WPG).sum(axis=1)
    for lo, hi in ((0, 256), (256, 512), (512, NWG)):
        if lo >= NWG: continue
        hi = min(hi, NWG)
        L_, T_ = wg_life_all[lo:hi], wg_trips[lo:hi]
        print(f"   blockIdx {lo:3d}-{hi:3d}: lifetime p50 {np.percentile(L_,50):.1f} p90 {np.percentile(L_,90):.1f} p99 {np.percentile(L_,99):.1f} max {L_.max():.1f};"
              f" batch-trips per workgroup p10 {np.percentile(T_,10):.0f} p50 {np.percentile(T_,50):.0f} p90 {np.percentile(T_,90):.0f} max {T_.max():.0f};"
              f" corr(lifetime, trips) {np.corrcoef(L_, T_)[0,1]:.2f}; workgroups later than p50+3us: {int((L_ > np.percentile(L_,50) + 3).sum())}")
    # per CU (blockIdx mod 256 on a full surface: one workgroup of each layer): total trips and the latest end
    if NWG >= 752:
        n = min(256, NWG - 512)
        cu_tr = wg_trips[0:n] + wg_trips[256:256 + n] + wg_trips[512:512 + n]
        cu_end = np.maximum.reduce([wg_life_all[0:n], wg_life_all[256:256 + n], wg_life_all[512:512 + n]])
        print(f"   per CU slot (b, b+256, b+512): total batch-trips p10 {np.percentile(cu_tr,10):.0f} p50 {np.percentile(cu_tr,50):.0f} p90 {np.percentile(cu_tr,90):.0f} max {cu_tr.max():.0f};"
              f" latest end p50 {np.percentile(cu_end,50):.1f} p90 {np.percentile(cu_end,90):.1f} max {cu_end.max():.1f}; corr {np.corrcoef(cu_tr, cu_end)[0,1]:.2f}")
    if os.environ.get("PERTRIP"):
        wtr = trips; it_us = raw[:, 6] / tick
        for lo, hi in ((0, 256), (256, NWG)):
            sel = np.repeat((np.arange(NWG) >= lo) & (np.arange(NWG) < hi), WPG) & (wtr > 0)
            if sel.any():
                pt = it_us[sel] / wtr[sel]
                print(f"   blockIdx {lo}-{hi}: waves {int(sel.sum())}; us per trip inside the iteration: p10 {np.percentile(pt,10):.3f} p50 {np.percentile(pt,50):.3f} p90 {np.percentile(pt,90):.3f} max {pt.max():.3f}; "
                      f"wave trips p50 {np.percentile(wtr[sel],50):.0f} max {wtr[sel].max():.0f}; batch phase minus iteration p50 {np.percentile((D(4,3)-it_us)[sel],50):.2f} us; start phase p50 {np.percentile(D(3,0)[sel],50):.2f}")
    ctx.close()
