"""If the FIRST k iterates of the similarity solve carried a relative error eps (single precision: ~1e-7), how far would the
stopped iterate and the trip count move?  The fixed-point map contracts, so an early error decays — by how much before
the stop rule looks?  numpy restatement of the iteration (scratch/lean_study.py), synthetic 1/4-degree surface.
usage: early_precision_study.py [nx ny] [config]"""
import sys
import numpy as np
import lean_study as ls

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1440, 560)
config = sys.argv[3] if len(sys.argv) > 3 else "default"
case, fluxes, params, atmos = ls.prep(nx, ny, config)
c = ls.cell_constants(case, fluxes, atmos)
base_iterate = ls.iterate


def iterate_early(c, fluxes, k_early, eps, rng):
    """ls.iterate with the perturbation applied to iterates 1..k_early only."""
    state = dict(it=0)
    orig_normal = rng.standard_normal

    class R:
        def standard_normal(self, n):
            return orig_normal(n)
    # re-implement by calling the library loop with a hook that switches eps off: simpler to copy the loop's tail
    import numpy_oracle as no
    from coflux import interface_computations as ic
    kap = fluxes.von_karman_constant
    stab = fluxes.stability_functions.name
    coare = isinstance(fluxes.similarity_form, ic.COARELogarithmicSimilarityProfile)
    n = c["Tv"].size
    g, h, h_bl, tol, maxit = 9.81, 10.0, 600.0, 1e-8, 100
    us = np.full(n, 1e-4); ts = us.copy(); qq = us.copy()
    its = np.zeros(n, np.int32); active = np.ones(n, bool)
    dU = np.sqrt(c["du"] ** 2 + c["dv"] ** 2)
    floor = fluxes.similarity_profile_floor
    it = 0
    while active.any() and it < maxit:
        b = g / c["Tv"] * (ts * (1 + c["delta"] * c["qv"]) + c["delta"] * c["Tv"] * qq)
        Jb = -us * b
        Ug = np.maximum(fluxes.gustiness_parameter * np.cbrt(np.maximum(Jb, 0.0) * h_bl), fluxes.minimum_gustiness)
        U = np.sqrt(c["du"] ** 2 + c["dv"] ** 2 + Ug * Ug)
        lu = no.momentum_length(fluxes.momentum_roughness_length, g, us, dU, c["Ts"])
        lq = no.scalar_length(fluxes.water_vapor_roughness_length, lu, us, c["Ts"])
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            invL = np.where(b == 0, 0.0, kap * b / (us * us))
        zh = h * invL

        def prof(psi, l):
            r = np.log(h / l) - psi(stab, zh)
            r = r if coare else r + psi(stab, l * invL)
            return np.maximum(r, floor)
        nus = kap / prof(no.psi_m, lu) * U
        chi = kap / prof(no.psi_h, lq)
        nts, nqs = chi * c["dth"], chi * c["dq"]
        if it < k_early and eps:
            nus = nus * (1 + eps * rng.standard_normal(n))
            f = 1 + eps * rng.standard_normal(n)
            nts, nqs = nts * f, nqs * f
        drift = np.abs(nus - us) + np.abs(nts - ts) + np.abs(nqs - qq)
        us = np.where(active, nus, us); ts = np.where(active, nts, ts); qq = np.where(active, nqs, qq)
        its += active
        it += 1
        active = active & ~(drift < tol)
    return us, ts, qq, its


rng = np.random.default_rng(7)
us, ts, qq, its = iterate_early(c, fluxes, 0, 0.0, rng)
print("wet cells", its.size, "trips mean %.2f min %d max %d" % (its.mean(), its.min(), its.max()))
scale = dict(us=1e-3, ts=1e-3, qq=1e-6)
for eps in (1e-7, 1e-6):
    for k in (1, 2, 3, 4, 5, 6):
        u2, t2, q2, i2 = iterate_early(c, fluxes, k, eps, rng)
        same = i2 == its
        err = np.maximum.reduce([np.abs(u2 - us) / np.maximum(np.abs(us), scale["us"]), np.abs(t2 - ts) / np.maximum(np.abs(ts), scale["ts"]),
                                 np.abs(q2 - qq) / np.maximum(np.abs(qq), scale["qq"])])
        print("eps %.0e on the first %d iterates: trip counts differ in %6d cells (%.1e); worst scaled error all %.2e, same-count cells %.2e, p99.9 %.2e"
              % (eps, k, (~same).sum(), (~same).mean(), err.max(), err[same].max(), np.percentile(err, 99.9)))
