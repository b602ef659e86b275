"""Where do the solver's workgroups land, and does co-residency explain their lifetimes?  (needs libcoflux_phase.so)"""
import sys, os, ctypes as C, collections
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
NWG = int(os.environ.get('NWG', '757'))
n = NWG * 4 * 8
out = (C.c_ulonglong * n)()
ctx.lib.cf_debug_phase_read(out, n)
a = np.array(out, dtype=np.uint64).reshape(NWG, 4, 8)
life = (a[:, :, 3].astype(np.float64) - a[:, :, 0].astype(np.float64)).max(axis=1) / 1950.0
hw = a[:, :, 4].astype(np.int64); xcc = a[:, :, 5].astype(np.int64) & 0xF
wave_id = hw & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("hw_id sample", [hex(int(v)) for v in hw[0]], "xcc", xcc[0])
key = [(int(xcc[g, 0]), int(se[g, 0]), int(sh[g, 0]), int(cu[g, 0])) for g in range(NWG)]
cnt = collections.Counter(key)
print("distinct CUs used:", len(cnt), " WGs per CU histogram:", collections.Counter(cnt.values()))
simd_of_waves = collections.Counter(tuple(sorted(int(s_) for s_ in simd[g])) for g in range(NWG))
print("SIMD sets of a WG's four waves:", simd_of_waves.most_common(4))
by = collections.defaultdict(list)
for g in range(NWG):
    by[cnt[key[g]]].append(life[g])
for c_, v in sorted(by.items()):
    print(f"  WGs on CUs hosting {c_} WGs: n={len(v)} lifetime median {np.median(v):.1f} min {min(v):.1f} max {max(v):.1f}")
# within a CU: order of arrival (blockIdx) vs lifetime
groups = collections.defaultdict(list)
for g in range(NWG): groups[key[g]].append(g)
first, second, third = [], [], []
for k_, gs in groups.items():
    gs = sorted(gs)
    for lst, g in zip((first, second, third), gs): lst.append(life[g])
print("  by arrival order on the CU: 1st %.1f  2nd %.1f  3rd %.1f (medians)" % (np.median(first), np.median(second), np.median(third) if third else float('nan')))
par = (np.arange(NWG) >> 3) & 1
for order_name, pick in (("1st", 0), ("2nd", 1), ("3rd", 2)):
    ev, od = [], []
    for k_, gs in groups.items():
        gs = sorted(gs)
        if len(gs) > pick:
            (od if par[gs[pick]] else ev).append(life[gs[pick]])
    print(f"  {order_name} on CU: parity-0 WGs median {np.median(ev):.1f} (n={len(ev)})  parity-1 WGs median {np.median(od):.1f} (n={len(od)})")
order_of = {}
for k_, gs in groups.items():
    for o, g in enumerate(sorted(gs)): order_of[g] = o
q = np.arange(NWG) >> 3
tab = collections.defaultdict(collections.Counter)
for g in range(NWG): tab[int(q[g]) // 8][order_of[g]] += 1
print("  arrival order by q//8 (q = blockIdx>>3):", {k_: dict(v) for k_, v in sorted(tab.items())})
print("  xcc of blockIdx 0..15:", [int(xcc[g, 0]) for g in range(16)])
ctx.close()
