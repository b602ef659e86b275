"""CPU study for the latitude-slab regime: which cells take the GENERAL psi at the roughness-length arguments
(|l/L| >= Z0) from the second trip on, how many trips they need, and which waves (64 consecutive wet cells in index order:
the lean kernel's batches) they slow down.  The numpy oracle's iteration (oracle/numpy_oracle.py, logarithmic profile) with
the arguments recorded per trip.  usage: python scratch/general_lane_study.py [ny] [ny_global] [j_offset]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, numpy_oracle as no, util
from coflux import interface_computations as ic
import oracle as orc
nx, h = 1440, 7
ny = int(sys.argv[1]) if len(sys.argv) > 1 else 70
nyg = int(sys.argv[2]) if len(sys.argv) > 2 else ny
j0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
case = util.build_case(nx, ny, h, h, ny_global=nyg, j_offset=j0)
g = orc.make_grid(nx, ny, h, h, 1)
at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
fluxes = ic.SimilarityTheoryFluxes()
th = no.Thermo(ic.AtmosphereThermodynamicsParameters()); sw = ic.SeawaterComposition()
oc = case["ocean"]; W = (slice(h, h + ny), slice(h, h + nx)); E = (slice(h, h + ny), slice(h + 1, h + nx + 1)); N = (slice(h + 1, h + ny + 1), slice(h, h + nx))
uo = 0.5 * (oc["u"][W] + oc["u"][E]); vo = 0.5 * (oc["v"][W] + oc["v"][N]); Ts = oc["T"][W] + 273.15; So = oc["S"][W]
wet = oc["mask"][W] != 0
ua, va, Ta, pa, qa = (at[k][W] for k in ("u", "v", "T", "p", "q"))
A = th.state_pTq(pa, Ta, qa)
qs = no.water_mole_fraction(sw, So) * th.svp_liquid(Ts) / (A["rho"] * th.Rv * Ts)
dq = th.q_vapor(A) - qs; dth = Ta + 9.81 * 10 / th.cp_m(A) - Ts; du, dv = ua - uo, va - vo
Sfc = th.state_pTq(pa, Ts, qs); Tv, qv = th.T_virtual(Sfc), th.q_vapor(Sfc); delta = th.eps - 1.0; kap = 0.4
us = np.full(Ts.shape, 1e-4); ts = us.copy(); qq = us.copy(); dU = np.sqrt(du * du + dv * dv)
active = wet.copy()
trips = np.zeros(Ts.shape, int)
Z0S = (2.0**-10, 2.0**-8, 2.0**-7, 2.0**-6)
general_trips = {z: np.zeros(Ts.shape, int) for z in Z0S}   # trips >= 2 in which the cell is general
for it in range(1, 101):
    b = 9.81 / Tv * (ts * (1 + delta * qv) + delta * Tv * qq); Jb = -us * b
    Ug = np.maximum(np.cbrt(np.maximum(Jb, 0.0) * 600.0), fluxes.minimum_gustiness); U = np.sqrt(du * du + dv * dv + Ug * Ug)
    lu = no.momentum_length(fluxes.momentum_roughness_length, 9.81, us, dU, Ts); lq = no.scalar_length(fluxes.water_vapor_roughness_length, lu, us, Ts)
    invL = kap * b / (us * us)
    zu, zq = np.abs(lu * invL), np.abs(lq * invL)
    if it >= 2:
        for z in Z0S:
            general_trips[z] += active & ~((zu < z) & (zq < z))
    def prof(psi, l):
        r = np.log(10.0 / l) - psi("edson2013", 10.0 * invL) + psi("edson2013", l * invL)
        return np.maximum(r, 1.0)
    nus = kap / prof(no.psi_m, lu) * U; nts = kap / prof(no.psi_h, lq) * dth; nqs = kap / prof(no.psi_h, lq) * dq
    drift = np.abs(nus - us) + np.abs(nts - ts) + np.abs(nqs - qq)
    us = np.where(active, nus, us); ts = np.where(active, nts, ts); qq = np.where(active, nqs, qq)
    trips += active
    active = active & ~(drift < 1e-8)
    if not active.any(): break
t = trips[wet]
print(f"1440x{ny} (global {nyg}, offset {j0}): {wet.sum()} wet cells; trips mean {t.mean():.2f} max {t.max()}; cells with >= max-2 trips: {(t >= t.max() - 2).sum()}")
# waves: 64 consecutive wet cells in index order (an approximation of the chunk lists: chunks are contiguous index ranges)
order = np.flatnonzero(wet.ravel())
nw = (order.size + 63) // 64
pad = np.full(nw * 64 - order.size, -1)
idx = np.concatenate([order, pad]).reshape(nw, 64)
def per_wave(a, fill=0):
    flat = np.concatenate([a.ravel(), [fill]])
    return flat[idx]
wt = per_wave(trips).max(axis=1)
print(f"waves {nw}; wave trips (max over lanes): mean {wt.mean():.2f} max {wt.max()}; waves at the max: {(wt == wt.max()).sum()}")
for z in Z0S:
    gt = general_trips[z]
    cells = (gt[wet] > 0).sum()
    wg = per_wave(gt)                    # per lane: trips >= 2 spent general
    # a wave pays the general block in every trip in which ANY of its still-active lanes is general; bound: max over lanes
    wave_general = wg.max(axis=1)
    cost = wt * 1.0 + wave_general * (0.20 / 0.675)   # in units of a small-only trip (0.675 us), general block +0.20 us
    print(f" Z0 = 2^{int(np.log2(z))}: cells general in some trip >= 2: {cells} ({cells / wet.sum():.3%}); waves with one: {(wave_general > 0).sum()} of {nw}; "
          f"slowest wave in small-trip units: {cost.max():.1f} (its trips {wt[cost.argmax()]}, general trips {wave_general[cost.argmax()]}); "
          f"with no general block at all: {wt.max():.1f}")
