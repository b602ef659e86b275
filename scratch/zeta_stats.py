"""CPU study: distribution of the roughness-length stability arguments ℓᵤ/L★, ℓ_q/L★ per iteration on the synthetic
surface (which lanes can take the small-argument ψ path of coflux_fast.hpp, and how many waves of 64 are uniform)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, numpy_oracle as no, util
from coflux import interface_computations as ic
nx, ny, h = 720, 280, 3
case = util.build_case(nx, ny, h, h, ny_global=280)
import oracle as orc
g = orc.make_grid(nx, ny, h, h, 1)
at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
fluxes = ic.SimilarityTheoryFluxes()
th = no.Thermo(ic.AtmosphereThermodynamicsParameters()); sw = ic.SeawaterComposition(); op = ic.OceanProperties()
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
for it in range(1, 25):
    b = 9.81 / Tv * (ts * (1 + delta * qv) + delta * Tv * qq); Jb = -us * b
    Ug = np.maximum(np.cbrt(np.maximum(Jb, 0.0) * 600.0), fluxes.minimum_gustiness); U = np.sqrt(du * du + dv * dv + Ug * Ug)
    lu = no.momentum_length(fluxes.momentum_roughness_length, 9.81, us, dU, Ts); lq = no.scalar_length(fluxes.water_vapor_roughness_length, lu, us, Ts)
    invL = kap * b / (us * us)
    zu, zq = np.abs(lu * invL)[active], np.abs(lq * invL)[active]
    row = []
    for t in (2.0**-11, 2.0**-10, 2.0**-8, 2.0**-6):
        p = np.mean((zu < t) & (zq < t)); row.append(f"{p:.4f}/{p**64:.2f}")
    print(it, int(active.sum()), " ".join(row), "max zu %.2e zq %.2e" % (zu.max(), zq.max()))
    def prof(psi, l):
        r = np.log(10.0 / l) - psi("edson2013", 10.0 * invL) + psi("edson2013", l * invL)
        return np.maximum(r, 1.0)
    nus = kap / prof(no.psi_m, lu) * U; nts = kap / prof(no.psi_h, lq) * dth; nqs = kap / prof(no.psi_h, lq) * dq
    drift = np.abs(nus - us) + np.abs(nts - ts) + np.abs(nqs - qq)
    us = np.where(active, nus, us); ts = np.where(active, nts, ts); qq = np.where(active, nqs, qq)
    active = active & ~(drift < 1e-8)
    if not active.any(): break
