"""Study for the round-3 'lean' solver iteration (scratch; CPU only).
For the synthetic 1/4-degree surface (or a smaller one): runs the :default / :corrected iteration in numpy with
per-iteration hooks and reports
  * trip-count histogram, ζ_h range per iteration,
  * how often a cell's ψ table segment changes from one iteration to the next (per cell and per 64-cell batch
    sorted by trip count): decides whether caching the segment's coefficients in registers pays,
  * the effect of a relative perturbation eps of every iterate on the final result and on the trip counts
    (accuracy budget of the fast primitives).
usage: python scratch/lean_study.py [nx ny] [config]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy_oracle as no  # noqa: E402
import oracle as orc  # noqa: E402
import util  # noqa: E402
from coflux import interface_computations as ic  # noqa: E402


def prep(nx, ny, config):
    case = util.build_case(nx, ny)
    fluxes, vd = util.CONFIGS[config]()
    params = ic.flux_params(fluxes, velocity_difference=vd)
    g = orc.make_grid(nx, ny, case["hx"], case["hy"], 1)
    atmos = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
    return case, fluxes, params, atmos


def cell_constants(case, fluxes, atmos, velocity_difference="relative"):
    hx, hy, nx, ny, ring = case["hx"], case["hy"], case["nx"], case["ny"], 1
    ocean = case["ocean"]
    th = no.Thermo(ic.AtmosphereThermodynamicsParameters())
    js, je, is_, ie = hy - ring, hy + ny + ring, hx - ring, hx + nx + ring
    W = (slice(js, je), slice(is_, ie))
    E = (slice(js, je), slice(is_ + 1, ie + 1))
    N = (slice(js + 1, je + 1), slice(is_, ie))
    uo = 0.5 * (ocean["u"][W] + ocean["u"][E])
    vo = 0.5 * (ocean["v"][W] + ocean["v"][N])
    Ts = ocean["T"][W] + 273.15
    So = ocean["S"][W]
    wet = ocean["mask"][W] != 0
    ua, va, Ta, pa, qa = (atmos[k][W] for k in ("u", "v", "T", "p", "q"))
    A = th.state_pTq(pa, Ta, qa)
    qs = no.water_mole_fraction(ic.SeawaterComposition(), So) * th.svp_liquid(Ts) / (A["rho"] * th.Rv * Ts)
    dq = th.q_vapor(A) - qs
    dth = Ta + 9.81 * 10.0 / th.cp_m(A) - Ts
    du, dv = (ua - uo, va - vo) if velocity_difference == "relative" else (ua, va)
    Sfc = th.state_pTq(pa, Ts, qs)
    Tv, qv = th.T_virtual(Sfc), th.q_vapor(Sfc)
    sel = wet
    return dict(Tv=Tv[sel], qv=qv[sel], dq=dq[sel], dth=dth[sel], du=du[sel], dv=dv[sel], Ts=Ts[sel], delta=th.eps - 1.0)


def iterate(c, fluxes, *, eps=0.0, rng=None, hook=None, g=9.81, h=10.0, h_bl=600.0, maxit=100, tol=1e-8, lq_rel_err=0.0):
    kap = fluxes.von_karman_constant
    stab = fluxes.stability_functions.name
    coare = isinstance(fluxes.similarity_form, ic.COARELogarithmicSimilarityProfile)
    n = c["Tv"].size
    us = np.full(n, 1e-4)
    ts = us.copy()
    qq = us.copy()
    its = np.zeros(n, np.int32)
    active = np.ones(n, bool)
    dU = np.sqrt(c["du"] ** 2 + c["dv"] ** 2)
    it = 0
    floor = fluxes.similarity_profile_floor
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
        if hook:
            hook(it, active, zh, lu * invL, lq * invL, us, ts, qq)
        lq_used = lq * (1.0 + lq_rel_err * (rng.standard_normal(n) if rng is not None and lq_rel_err else 0.0))

        def prof(psi, l, l_used):
            r = np.log(h / l) - psi(stab, zh)
            r = r if coare else r + psi(stab, l_used * invL)
            return np.maximum(r, floor)

        nus = kap / prof(no.psi_m, lu, lu) * U
        chi = kap / prof(no.psi_h, lq, lq_used)
        nts = chi * c["dth"]
        nqs = chi * c["dq"]
        if eps:
            nus = nus * (1 + eps * rng.standard_normal(n))
            nts = nts * (1 + eps * rng.standard_normal(n))
            nqs = nqs * (1 + eps * rng.standard_normal(n))
        drift = np.abs(nus - us) + np.abs(nts - ts) + np.abs(nqs - qq)
        us = np.where(active, nus, us)
        ts = np.where(active, nts, ts)
        qq = np.where(active, nqs, qq)
        its += active
        it += 1
        active = active & ~(drift < tol)
    return us, ts, qq, its


def segment(z):
    x = np.minimum(1.0 + 16.0 * np.abs(z), np.ldexp(1.0, 34) * (1 - 2.0 ** -53))
    m, e = np.frexp(x)  # x = m 2^e, m in [.5, 1)
    k = 4 * (e - 1) + np.floor((2 * m - 1) * 4).astype(np.int64)
    return np.where(z < 0, -k - 1, k)


def main():
    nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (360, 140)
    config = sys.argv[3] if len(sys.argv) > 3 else "default"
    case, fluxes, params, atmos = prep(nx, ny, config)
    c = cell_constants(case, fluxes, atmos)
    n = c["Tv"].size
    print("wet cells", n, "config", config)
    segs = []
    zstats = []

    def hook(it, active, zh, zu, zq, us, ts, qq):
        segs.append((active.copy(), segment(zh)))
        a = np.abs(zh[active])
        zstats.append((it, active.sum(), np.percentile(a, [1, 50, 99, 100]), np.percentile(np.abs(zu[active]), [50, 99, 100]),
                       np.percentile(np.abs(zq[active]), [50, 99, 100])))

    us, ts, qq, its = iterate(c, fluxes, hook=hook)
    print("trip counts: mean %.2f min %d max %d" % (its.mean(), its.min(), its.max()))
    print("histogram:", np.bincount(its)[: its.max() + 1].tolist())
    for it, na, zh, zu, zq in zstats[:25]:
        print("it %2d active %7d |zh| p1/50/99/max %s |zu| p50/99/max %s |zq| %s" % (it, na, np.array2string(zh, precision=3), np.array2string(zu, precision=2), np.array2string(zq, precision=2)))
    # segment changes: per cell, and per batch of 64 cells sorted by trip count (descending)
    order = np.argsort(-its, kind="stable")
    nb = (n + 63) // 64
    total_batch_iters = 0
    reload_batch_iters = 0
    changed_cells = 0
    active_cells = 0
    per_it = []
    for k in range(1, len(segs)):
        act = segs[k][0]
        ch = (segs[k][1] != segs[k - 1][1]) & act
        changed_cells += ch.sum()
        active_cells += act.sum()
        a = np.zeros(nb * 64, bool)
        a[:n] = act[order]
        cb = np.zeros(nb * 64, bool)
        cb[:n] = ch[order]
        ab = a.reshape(nb, 64).any(1)
        rb = cb.reshape(nb, 64).any(1)
        total_batch_iters += ab.sum()
        reload_batch_iters += rb.sum()
        per_it.append((k, int(ab.sum()), int(rb.sum()), int(ch.sum()), int(act.sum())))
    total_batch_iters += nb  # iteration 0 always loads
    reload_batch_iters += nb
    print("cells: segment changed in %.1f%% of cell-iterations (k>=1)" % (100.0 * changed_cells / active_cells))
    print("batches (sorted by exact trip count): reload needed in %d of %d batch-iterations = %.1f%%" % (reload_batch_iters, total_batch_iters, 100.0 * reload_batch_iters / total_batch_iters))
    for k, ab, rb, chc, ac in per_it[:22]:
        print("  it %2d: active batches %6d reloading %6d (%.0f%%)  cells changed %.2f%%" % (k, ab, rb, 100.0 * rb / max(ab, 1), 100.0 * chc / max(ac, 1)))
    # perturbation study
    rng = np.random.default_rng(1)
    scale = dict(us=1e-3, ts=1e-3, qq=1e-6)
    for eps in (1e-15, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9):
        u2, t2, q2, i2 = iterate(c, fluxes, eps=eps, rng=rng)
        same = i2 == its
        e_all = max(np.max(np.abs(u2 - us) / np.maximum(np.abs(us), scale["us"])), np.max(np.abs(t2 - ts) / np.maximum(np.abs(ts), scale["ts"])),
                    np.max(np.abs(q2 - qq) / np.maximum(np.abs(qq), scale["qq"])))
        e_same = max(np.max((np.abs(u2 - us) / np.maximum(np.abs(us), scale["us"]))[same]), np.max((np.abs(t2 - ts) / np.maximum(np.abs(ts), scale["ts"]))[same]),
                     np.max((np.abs(q2 - qq) / np.maximum(np.abs(qq), scale["qq"]))[same]))
        print("eps %.0e: trip counts differ in %d cells (%.2e), worst scaled error all %.2e, same-count cells %.2e" % (eps, (~same).sum(), (~same).mean(), e_all, e_same))
    for lqe in (1e-7, 1e-6):
        u2, t2, q2, i2 = iterate(c, fluxes, rng=rng, lq_rel_err=lqe)
        same = i2 == its
        e_all = max(np.max(np.abs(u2 - us) / np.maximum(np.abs(us), scale["us"])), np.max(np.abs(t2 - ts) / np.maximum(np.abs(ts), scale["ts"])),
                    np.max(np.abs(q2 - qq) / np.maximum(np.abs(qq), scale["qq"])))
        print("lq relative error %.0e in psi_h(lq/L): trip counts differ in %d cells, worst scaled error %.2e" % (lqe, (~same).sum(), e_all))


if __name__ == "__main__":
    main()
