"""CPU study for the certified reduced-iteration solve (round 5): 2-D state (u*, chi), candidate starts and
accelerations, evaluations per cell, the truncation bound of the reference's stop rule and the share of
cells a certificate would send down the exact path.  NumPy oracle primitives; nothing here is product."""
import sys
sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/climaocean.jl_amd')
import numpy as np
import oracle as orc, numpy_oracle as npo
from coflux import synthetic as syn, interface_computations as ic

STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nx, ny, hx, hy = 1440, 560, 3, 3
g = orc.make_grid(nx, ny, hx, hy, 0)
oc = syn.ocean_state(nx, ny, hx, hy)
src = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, hx, hy)
at = orc.interpolate_atmosphere_state(g, src, dict(separable=True, fi=fi, fj=fj))


def setup(fluxes, rel=True):
    th = npo.Thermo(ic.AtmosphereThermodynamicsParameters()); sw = ic.SeawaterComposition()
    W = (slice(hy, hy + ny, STEP), slice(hx, hx + nx, STEP)); E = (slice(hy, hy + ny, STEP), slice(hx + 1, hx + nx + 1, STEP))
    N = (slice(hy + 1, hy + ny + 1, STEP), slice(hx, hx + nx, STEP))
    uo = 0.5 * (oc['u'][W] + oc['u'][E]); vo = 0.5 * (oc['v'][W] + oc['v'][N]); Ts = oc['T'][W] + 273.15; So = oc['S'][W]
    wet = oc['mask'][W] != 0
    ua, va, Ta, pa, qa = (at[k][W] for k in 'uvTpq')
    A = th.state_pTq(pa, Ta, qa); qs = npo.water_mole_fraction(sw, So) * th.svp_liquid(Ts) / (A['rho'] * th.Rv * Ts)
    dq = th.q_vapor(A) - qs; dth = Ta + 9.81 * 10 / th.cp_m(A) - Ts
    du, dv = (ua - uo, va - vo) if rel else (ua, va)
    S = th.state_pTq(pa, Ts, qs); Tv, qv = th.T_virtual(S), th.q_vapor(S); delta = th.eps - 1
    dU = np.sqrt(du * du + dv * dv); kap = 0.4; stab = fluxes.stability_functions.name
    coare = isinstance(fluxes.similarity_form, ic.COARELogarithmicSimilarityProfile)
    rho, cpm, Lv = A['rho'], th.cp_m(A), th.Lv(Ta)

    def F3(x):
        us, ts, qq = x
        b = 9.81 / Tv * (ts * (1 + delta * qv) + delta * Tv * qq); Jb = -us * b
        Ug = np.maximum(fluxes.gustiness_parameter * np.cbrt(np.maximum(Jb, 0) * 600), fluxes.minimum_gustiness)
        U = np.sqrt(du * du + dv * dv + Ug * Ug)
        lu = npo.momentum_length(fluxes.momentum_roughness_length, 9.81, us, dU, Ts)
        lq = npo.scalar_length(fluxes.water_vapor_roughness_length, lu, us, Ts)
        with np.errstate(all='ignore'):
            L = np.where(b == 0, np.inf, us * us / (kap * b))

            def prof(psi, l):
                r = np.log(10 / l) - psi(stab, 10 / L)
                r = r if coare else r + psi(stab, l / L)
                return np.maximum(r, fluxes.similarity_profile_floor)
            return np.array([kap / prof(npo.psi_m, lu) * U, kap / prof(npo.psi_h, lq) * dth, kap / prof(npo.psi_h, lq) * dq])

    def G2(x):   # state (u*, chi)
        y = F3(np.array([x[0], x[1] * dth, x[1] * dq]))
        with np.errstate(all='ignore'):
            chi = np.where(np.abs(dth) >= np.abs(dq), y[1] / np.where(dth == 0, 1, dth), y[2] / np.where(dq == 0, 1, dq))
        return np.array([y[0], chi])
    return dict(F3=F3, G2=G2, wet=wet, dth=dth, dq=dq, dU=dU, rho=rho, cpm=cpm, Lv=Lv, du=du, dv=dv, Ts=Ts, Ta=Ta,
                Tv=Tv, qv=qv, delta=delta, Sabs=np.abs(dth) + np.abs(dq), fluxes=fluxes)


def fluxes_of(c, us, chi):
    ts, qq = chi * c['dth'], chi * c['dq']
    Fv = -c['rho'] * us * qq
    with np.errstate(all='ignore'):
        dirx = np.where(c['dU'] > 0, c['du'] / np.where(c['dU'] > 0, c['dU'], 1), 0); diry = np.where(c['dU'] > 0, c['dv'] / np.where(c['dU'] > 0, c['dU'], 1), 0)
    return dict(Qc=-c['rho'] * c['cpm'] * us * ts, Qv=Fv * c['Lv'], Fv=Fv, tx=-c['rho'] * us * us * dirx, ty=-c['rho'] * us * us * diry)


SCALE = dict(Qc=1.0, Qv=1.0, Fv=1e-6, tx=1e-3, ty=1e-3)


def flux_err(c, a, b):
    fa, fb = fluxes_of(c, *a), fluxes_of(c, *b)
    e = np.zeros(c['wet'].shape)
    for k in fa:
        e = np.maximum(e, np.abs(fa[k] - fb[k]) / np.maximum(np.abs(fb[k]), SCALE[k]))
    return np.where(c['wet'], e, 0)


def reference_path(c, tol=1e-8, maxit=100):
    F3, wet = c['F3'], c['wet']
    x = np.full((3,) + wet.shape, 1e-4); n = np.zeros(wet.shape, int); act = wet.copy()
    for it in range(maxit):
        xn = F3(x); d = np.abs(xn - x).sum(0); x = np.where(act, xn, x); n += act; act = act & ~(d < tol)
        if not act.any(): break
    with np.errstate(all='ignore'):
        chi = np.where(np.abs(c['dth']) >= np.abs(c['dq']), x[1] / np.where(c['dth'] == 0, 1, c['dth']), x[2] / np.where(c['dq'] == 0, 1, c['dq']))
    return x[0], chi, n


def fixed_point(c, x0, n=400):
    x = x0.copy()
    for _ in range(n): x = c['G2'](x)
    return x


def start_neutral(c):
    U0 = np.sqrt(c['dU'] ** 2 + c['fluxes'].minimum_gustiness ** 2)
    return np.array([0.035 * U0, np.full(U0.shape, 0.4 / np.log(10 / 1e-4))])


def norm(c, f):
    return np.abs(f[0]) + np.abs(f[1]) * c['Sabs']


def solve_accel(c, x0, m=2, warm=1, tol=1e-10, maxit=40):
    """Anderson(m) on the 2-D state, m in 0,1,2.  Returns x (the last G(x) of each lane), evals, and the secant Jacobian."""
    G2, wet = c['G2'], c['wet']
    x = x0.copy(); n = np.zeros(wet.shape, int); act = wet.copy(); out = x.copy()
    X, Gs = [], []
    J = np.zeros((2, 2) + wet.shape)
    for it in range(maxit):
        gx = G2(x); f = gx - x
        d = norm(c, f)
        newly = act & (d < tol)
        out = np.where(newly, gx, out)
        n += act; act = act & ~newly
        X.append(x.copy()); Gs.append(gx.copy())
        if not act.any(): break
        k = len(X)
        mk = min(m, k - 1) if it >= warm else 0
        xn = gx
        if mk >= 1:
            dx1 = X[-1] - X[-2]; dg1 = Gs[-1] - Gs[-2]; df1 = dg1 - dx1
            sw = np.array([np.ones(wet.shape), c['Sabs']])   # weights of the reference norm
            if mk == 1:
                den = ((df1 * sw) ** 2).sum(0)
                gam = np.where(den > 0, ((f * sw) * (df1 * sw)).sum(0) / np.where(den > 0, den, 1), 0)
                xn = gx - gam * dg1
            else:
                dx2 = X[-2] - X[-3]; dg2 = Gs[-2] - Gs[-3]; df2 = dg2 - dx2
                det = df1[0] * df2[1] - df1[1] * df2[0]
                ok = np.abs(det) > 1e-12 * (np.abs(df1[0] * df2[1]) + np.abs(df1[1] * df2[0])) + 1e-300
                dets = np.where(ok, det, 1)
                g1 = (f[0] * df2[1] - f[1] * df2[0]) / dets
                g2 = (df1[0] * f[1] - df1[1] * f[0]) / dets
                xa = gx - g1 * dg1 - g2 * dg2
                den = ((df1 * sw) ** 2).sum(0)
                gam = np.where(den > 0, ((f * sw) * (df1 * sw)).sum(0) / np.where(den > 0, den, 1), 0)
                xb = gx - gam * dg1
                xn = np.where(ok, xa, xb)
                # secant Jacobian of G from the two difference pairs:  J [dx1 dx2] = [dg1 dg2]
                dX = dx1[0] * dx2[1] - dx1[1] * dx2[0]
                okJ = ok & (np.abs(dX) > 0)
                dXs = np.where(okJ, dX, 1)
                Jn = np.array([[(dg1[0] * dx2[1] - dg2[0] * dx1[1]) / dXs, (dg2[0] * dx1[0] - dg1[0] * dx2[0]) / dXs],
                               [(dg1[1] * dx2[1] - dg2[1] * dx1[1]) / dXs, (dg2[1] * dx1[0] - dg1[1] * dx2[0]) / dXs]])
                J = np.where(act & okJ, Jn, J)
        bad = (xn[0] <= 0) | (xn[1] <= 0) | ~np.isfinite(xn).all(0) | (xn[0] > 4 * gx[0]) | (xn[0] < 0.25 * gx[0]) | (xn[1] > 4 * gx[1]) | (xn[1] < 0.25 * gx[1])
        xn = np.where(bad, gx, xn)
        x = np.where(act, xn, x)
    return out, n, act, J


def jac_fd(c, x, h=1e-6):
    G2 = c['G2']; g0 = G2(x)
    J = np.zeros((2, 2) + x.shape[1:])
    for k in range(2):
        xp = x.copy(); dx = np.maximum(np.abs(x[k]) * h, 1e-12); xp[k] = x[k] + dx; xm = x.copy(); xm[k] = x[k] - dx
        J[:, k] = (G2(xp) - G2(xm)) / (2 * dx)
    return J


def run(name, fluxes, rel=True):
    c = setup(fluxes, rel); wet = c['wet']
    us_r, chi_r, n_r = reference_path(c)
    xs = fixed_point(c, np.array([us_r, chi_r]))
    e_ref = flux_err(c, (us_r, chi_r), (xs[0], xs[1]))
    print('%s: %d wet cells; reference path: mean %.2f max %d evals; stopped iterate vs fixed point (flux metric): max %.2e, >1e-6: %.1f ppm, >5e-7: %.1f ppm, >2e-7: %.1f ppm'
          % (name, wet.sum(), n_r[wet].mean(), n_r.max(), e_ref.max(), 1e6 * (e_ref[wet] > 1e-6).mean(), 1e6 * (e_ref[wet] > 5e-7).mean(), 1e6 * (e_ref[wet] > 2e-7).mean()))
    # spectral radius of the map at the fixed point
    J = jac_fd(c, xs)
    # scale chi by S to work in the reference norm's units
    tr = J[0, 0] + J[1, 1]; det = J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0]
    disc = tr * tr - 4 * det
    lam1 = np.where(disc >= 0, (tr + np.sign(tr) * np.sqrt(np.abs(disc))) / 2, np.sqrt(np.abs(det)))
    rho = np.abs(lam1)
    print('   spectral radius at the fixed point: median %.3f, 90%% %.3f, 99%% %.3f, max %.3f; complex pairs %.2f%%, negative dominant %.2f%%'
          % (np.median(rho[wet]), np.quantile(rho[wet], .9), np.quantile(rho[wet], .99), rho[wet].max(), 100 * (disc[wet] < 0).mean(), 100 * ((disc >= 0) & (lam1 < 0))[wet].mean()))
    x0 = start_neutral(c)
    e0 = np.abs(x0 - xs) / xs
    print('   neutral start: rel. distance u* median %.2f max %.2f; chi median %.2f max %.2f' % (np.median(e0[0][wet]), e0[0][wet].max(), np.median(e0[1][wet]), e0[1][wet].max()))
    for m in (0, 1, 2):
        for warm in ((1,) if m == 0 else (1, 2)):
            for tol in (1e-9, 1e-10):
                out, n, act, Js = solve_accel(c, x0, m=m, warm=warm, tol=tol)
                e = flux_err(c, (out[0], out[1]), (xs[0], xs[1]))
                hist = np.bincount(n[wet], minlength=12)
                print('   AA(%d) warm %d tol %g: evals mean %.2f max %d (hist %s) unconverged %d; vs fixed point max %.2e' % (m, warm, tol, n[wet].mean(), n[wet].max(), hist[:16].tolist(), act.sum(), e.max()))
    return c, xs, (us_r, chi_r, n_r), J




def bound_from_J(c, J, x, tol=1e-8):
    """Worst-case flux error (test metric) of the reference's stopped iterate relative to the fixed point x, from
    e = J (J - I)^-1 d with |d_u| + S |d_chi| < tol."""
    a, b, cc, d = J[0, 0], J[0, 1], J[1, 0], J[1, 1]
    # (J - I)^-1
    det = (a - 1) * (d - 1) - b * cc
    i11, i12, i21, i22 = (d - 1) / det, -b / det, -cc / det, (a - 1) / det
    M11, M12 = a * i11 + b * i21, a * i12 + b * i22
    M21, M22 = cc * i11 + d * i21, cc * i12 + d * i22
    S = np.maximum(c['Sabs'], 1e-300)
    u, chi = x
    eu = tol * np.maximum(np.abs(M11), np.abs(M12) / S)
    # stress: rho u^2
    tau = c['rho'] * u * u
    e_tau = c['rho'] * 2 * u * eu / np.maximum(tau, 1e-3)
    # scalar fluxes: K * u * chi * D  (K D = rho cp dth, rho Lv dq, rho dq)
    eq = tol * np.maximum(np.abs(chi * M11 + u * M21), np.abs(chi * M12 + u * M22) / S)
    out = e_tau
    for K, D, sc in ((c['rho'] * c['cpm'], c['dth'], 1.0), (c['rho'] * c['Lv'], c['dq'], 1.0), (c['rho'], c['dq'], 1e-6)):
        flux = np.abs(K * D * u * chi)
        out = np.maximum(out, np.abs(K * D) * eq / np.maximum(flux, sc))
    return np.where(c['wet'], out, 0)


def study_bound(name, fluxes, rel=True):
    c, xs, (us_r, chi_r, n_r), Jfd = run(name, fluxes, rel)
    wet = c['wet']
    e_ref = flux_err(c, (us_r, chi_r), (xs[0], xs[1]))
    bfd = bound_from_J(c, Jfd, xs)
    out, n, act, Js = solve_accel(c, start_neutral(c), m=2, warm=1, tol=1e-9)
    bse = bound_from_J(c, Js, out)
    print('   bound (FD Jacobian) / actual error: min %.2f median %.2f;  violations (actual > bound): %d' % ((bfd / np.maximum(e_ref, 1e-300))[wet].min(), np.median((bfd / np.maximum(e_ref, 1e-300))[wet]), (e_ref > bfd)[wet].sum()))
    print('   bound (secant Jacobian): violations %d; ratio to the FD bound: 1%% %.2f median %.2f 99%% %.2f' % ((e_ref > bse)[wet].sum(), np.quantile((bse / bfd)[wet], .01), np.median((bse / bfd)[wet]), np.quantile((bse / bfd)[wet], .99)))
    for budget in (2e-7, 3e-7, 4e-7, 5e-7):
        print('   budget %.0e: actual beyond %.1f ppm; flagged by FD bound %.1f ppm, by secant bound %.1f ppm, by secant bound x1.25 %.1f ppm; worst actual error among unflagged (secant x1.25) %.2e'
              % (budget, 1e6 * (e_ref > budget)[wet].mean(), 1e6 * (bfd > budget)[wet].mean(), 1e6 * (bse > budget)[wet].mean(), 1e6 * (1.25 * bse > budget)[wet].mean(), e_ref[wet & ~(1.25 * bse > budget)].max()))




def flag_anatomy(name, fluxes):
    c = setup(fluxes); wet = c['wet']
    us_r, chi_r, n_r = reference_path(c)
    xs = fixed_point(c, np.array([us_r, chi_r]))
    J = jac_fd(c, xs)
    b = bound_from_J(c, J, xs)
    e_ref = flux_err(c, (us_r, chi_r), (xs[0], xs[1]))
    for budget in (4e-7, 6e-7, 8e-7, 9e-7):
        fl = wet & (b > budget)
        print(name, 'budget %.0e flagged %.2f%%; their median S %.3f, u* %.3f, |dth| %.3f, dU %.2f; ref trips mean %.1f; worst actual unflagged %.2e'
              % (budget, 100 * fl.sum() / wet.sum(), np.median(c['Sabs'][fl]), np.median(xs[0][fl]), np.median(np.abs(c['dth'])[fl]), np.median(c['dU'][fl]), n_r[fl].mean(), e_ref[wet & ~fl].max()))
    fl = wet & (b > 4e-7)
    print('   share of flagged with u*<0.1: %.2f, S<0.5: %.2f' % ((xs[0][fl] < 0.1).mean(), (c['Sabs'][fl] < 0.5).mean()))




def certified_proto(c, tol_ref=1e-8, tol_a=1e-7, budget=8e-7, maxev=12, x0=None, safety=1.25):
    """The algorithm as the kernel would run it: plain, plain, then AA(2) steps; accept the AA step itself (no evaluation
    there) once the relative residual is below tol_a; certificate from the secant Jacobian."""
    G2, wet = c['G2'], c['wet']; S = c['Sabs']
    x = start_neutral(c) if x0 is None else x0.copy()
    n = np.zeros(wet.shape, int); act = wet.copy(); out = x.copy()
    J = np.full((2, 2) + wet.shape, np.nan); haveJ = np.zeros(wet.shape, bool)
    gp = fp = dg2 = df2 = None
    fail = np.zeros(wet.shape, bool)
    for it in range(maxev):
        g = G2(x); f = g - x
        n += act
        xn = g.copy(); aa_ok = np.zeros(wet.shape, bool)
        if it >= 1:
            dg1 = g - gp; df1 = f - fp
        if it >= 2:
            det = df1[0] * df2[1] - df1[1] * df2[0]
            ok = np.abs(det) > 1e-10 * (np.abs(df1[0] * df2[1]) + np.abs(df1[1] * df2[0])) + 1e-300
            dets = np.where(ok, det, 1)
            g1 = (f[0] * df2[1] - f[1] * df2[0]) / dets; g2 = (df1[0] * f[1] - df1[1] * f[0]) / dets
            xa = g - g1 * dg1 - g2 * dg2
            good = ok & np.isfinite(xa).all(0) & (xa[0] > 0.5 * g[0]) & (xa[0] < 2 * g[0]) & (xa[1] > 0.5 * g[1]) & (xa[1] < 2 * g[1])
            xn = np.where(good, xa, g); aa_ok = good
            dx1 = dg1 - df1; dx2 = dg2 - df2
            dX = dx1[0] * dx2[1] - dx1[1] * dx2[0]
            okJ = good & (np.abs(dX) > 1e-10 * (np.abs(dx1[0] * dx2[1]) + np.abs(dx1[1] * dx2[0])) + 1e-300)
            dXs = np.where(okJ, dX, 1)
            Jn = np.array([[(dg1[0] * dx2[1] - dg2[0] * dx1[1]) / dXs, (dg2[0] * dx1[0] - dg1[0] * dx2[0]) / dXs],
                           [(dg1[1] * dx2[1] - dg2[1] * dx1[1]) / dXs, (dg2[1] * dx1[0] - dg1[1] * dx2[0]) / dXs]])
            upd = act & okJ
            J = np.where(upd, Jn, J); haveJ |= upd
        # relative residual
        with np.errstate(all='ignore'):
            r = np.abs(f[0]) / g[0] + np.abs(f[1]) / g[1]
        done = act & (r < tol_a) & (aa_ok | (r < 1e-3 * tol_a)) & haveJ
        out = np.where(done, xn, out)
        act = act & ~done
        if not act.any(): break
        if it >= 1:
            dg2, df2 = dg1, df1
        gp, fp = g, f
        x = np.where(act, xn, x)
    fail = act.copy()
    b = bound_from_J(c, np.where(haveJ, J, 0.5), out, tol_ref) * safety
    with np.errstate(all='ignore'):
        flagged = wet & (fail | ~haveJ | ~(b <= budget))
    return out, n, flagged, b


def proto_report(name, fluxes, rel=True, **kw):
    c = setup(fluxes, rel); wet = c['wet']
    us_r, chi_r, n_r = reference_path(c)
    xs = fixed_point(c, np.array([us_r, chi_r]))
    for tol_a in (1e-5, 1e-6, 1e-7, 1e-8):
        out, n, fl, b = certified_proto(c, tol_a=tol_a, **kw)
        e_fp = flux_err(c, (out[0], out[1]), (xs[0], xs[1]))
        e_rf = flux_err(c, (out[0], out[1]), (us_r, chi_r))
        ok = wet & ~fl
        print('%s tol_a %.0e: evals mean %.2f max %d hist %s; flagged %.2f%%; unflagged: vs fixed point max %.2e, vs reference max %.2e (99.9%% %.2e)'
              % (name, tol_a, n[wet].mean(), n[wet].max(), np.bincount(n[wet], minlength=9)[3:10].tolist(), 100 * fl.sum() / wet.sum(), e_fp[ok].max(), e_rf[ok].max(), np.quantile(e_rf[ok], .999)))




def certified_proto32(c, tol_ref=1e-8, tol_a=1e-7, budget=8e-7, maxev=10, safety=1.25, guard=1e-4, u0c=0.035, chi0=None, x0=None):
    """certified_proto with the Anderson bookkeeping in float32 the way the kernel carries it: step s, previous residual,
    the older difference pair; corrections converted once."""
    f32 = np.float32
    G2, wet = c['G2'], c['wet']
    x = start_neutral(c) if x0 is None else x0.copy()
    if x0 is None:
        x[0] *= u0c / 0.035
        if chi0 is not None: x[1] = chi0
    n = np.zeros(wet.shape, int); act = wet.copy(); out = x.copy()
    Jf = np.zeros((2, 2) + wet.shape, f32); haveJ = np.zeros(wet.shape, bool)
    s = fP = dg2 = df2 = None
    for it in range(maxev):
        g = G2(x); f = g - x
        n += act
        ff = f.astype(f32)
        corr = np.zeros_like(ff); aa_ok = np.zeros(wet.shape, bool)
        with np.errstate(all='ignore'):
            if it >= 1:
                df1 = ff - fP; dg1 = s + df1
            if it >= 2:
                det = df1[0] * df2[1] - df1[1] * df2[0]
                ok = np.abs(det) > f32(guard) * (np.abs(df1[0] * df2[1]) + np.abs(df1[1] * df2[0]))
                rdet = f32(1) / np.where(ok, det, f32(1))
                g1 = (ff[0] * df2[1] - ff[1] * df2[0]) * rdet; g2 = (df1[0] * ff[1] - df1[1] * ff[0]) * rdet
                cr = np.array([g1 * dg1[0] + g2 * dg2[0], g1 * dg1[1] + g2 * dg2[1]])
                good = ok & np.isfinite(cr).all(0) & (np.abs(cr[0]) < 0.5 * g[0]) & (np.abs(cr[1]) < 0.5 * g[1])
                corr = np.where(good, cr, f32(0)); aa_ok = good
                # secant Jacobian:  J [dx1 dx2] = [dg1 dg2],  dx1 = s, dx2 = dg2 - df2
                dx1 = s; dx2 = dg2 - df2
                dX = dx1[0] * dx2[1] - dx1[1] * dx2[0]
                okJ = good & (np.abs(dX) > f32(guard) * (np.abs(dx1[0] * dx2[1]) + np.abs(dx1[1] * dx2[0])))
                rX = f32(1) / np.where(okJ, dX, f32(1))
                Jn = np.array([[(dg1[0] * dx2[1] - dg2[0] * dx1[1]) * rX, (dg2[0] * dx1[0] - dg1[0] * dx2[0]) * rX],
                               [(dg1[1] * dx2[1] - dg2[1] * dx1[1]) * rX, (dg2[1] * dx1[0] - dg1[1] * dx2[0]) * rX]])
                upd = act & okJ
                Jf = np.where(upd, Jn, Jf); haveJ |= upd
            xn = g - corr.astype(np.float64)
            r = np.abs(f[0]) / g[0] + np.abs(f[1]) / g[1]
        done = act & (r < tol_a) & aa_ok & haveJ
        out = np.where(done, xn, out)
        act = act & ~done
        if not act.any(): break
        if it >= 1:
            dg2, df2 = dg1, df1
        fP = ff; s = ff - corr
        x = np.where(act, xn, x)
    fail = act.copy()
    b = bound_from_J(c, np.where(haveJ, Jf.astype(np.float64), 0.5), out, tol_ref) * safety
    with np.errstate(all='ignore'):
        flagged = wet & (fail | ~haveJ | ~(b <= budget))
    return out, n, flagged, b


def proto32_report(name, fluxes, rel=True, **kw):
    c = setup(fluxes, rel); wet = c['wet']
    us_r, chi_r, n_r = reference_path(c)
    xs = fixed_point(c, np.array([us_r, chi_r]))
    for tol_a in (1e-6, 1e-7):
        out, n, fl, b = certified_proto32(c, tol_a=tol_a, **kw)
        e_fp = flux_err(c, (out[0], out[1]), (xs[0], xs[1]))
        e_rf = flux_err(c, (out[0], out[1]), (us_r, chi_r))
        ok = wet & ~fl
        # batch maxima: 64 consecutive wet cells
        nn = n[wet]; nb = len(nn) // 64; bm = nn[:nb * 64].reshape(nb, 64).max(1)
        print('%s f32-AA tol_a %.0e: evals mean %.2f max %d hist %s; batch-max mean %.2f; flagged %.2f%% (failed %d); unflagged: vs fixed point max %.2e, vs reference max %.2e'
              % (name, tol_a, n[wet].mean(), n[wet].max(), np.bincount(n[wet], minlength=9)[3:11].tolist(), bm.mean(), 100 * fl.sum() / wet.sum(), (n >= 10)[wet].sum(), e_fp[ok].max(), e_rf[ok].max()))


def bound_refined(c, J, x, tol=1e-8, r_pow=1.5, eps_sub=1e-3):
    """Directional truncation bound: where J has real, separated eigenvalues (lam2^2 <= |lam1|^(2 r_pow)) the reference's
    last drift lies along the dominant eigenvector up to eps_sub; elsewhere the worst case over all directions."""
    worst = bound_from_J(c, J, x, tol)
    a, b, cc, d = J[0, 0], J[0, 1], J[1, 0], J[1, 1]
    tr, det = a + d, a * d - b * cc
    disc = tr * tr - 4 * det
    sq = np.sqrt(np.maximum(disc, 0))
    l1 = np.where(tr >= 0, (tr + sq) / 2, (tr - sq) / 2)      # dominant (larger |.|)
    with np.errstate(all='ignore'):
        l2 = np.where(l1 != 0, det / np.where(l1 != 0, l1, 1), 0)
    sep = (disc > 0) & (l2 * l2 <= np.abs(l1) ** (2 * r_pow)) & (np.abs(l1) < 0.95)
    # eigenvector of l1: (b, l1 - a) or (l1 - d, cc): the better conditioned
    v1a = np.array([b, l1 - a]); v1b = np.array([l1 - d, cc])
    use_a = (np.abs(v1a[0]) + np.abs(v1a[1])) >= (np.abs(v1b[0]) + np.abs(v1b[1]))
    v = np.where(use_a, v1a, v1b)
    S = np.maximum(c['Sabs'], 1e-300)
    nw = np.abs(v[0]) + S * np.abs(v[1])
    with np.errstate(all='ignore'):
        s = tol / np.where(nw > 0, nw, 1)
        mu = np.abs(l1 / (l1 - 1))
    u, chi = x
    eu = mu * s * np.abs(v[0])
    eq = mu * s * np.abs(chi * v[0] + u * v[1])
    out = c['rho'] * 2 * u * eu / np.maximum(c['rho'] * u * u, 1e-3)
    for K, D, sc in ((c['rho'] * c['cpm'], c['dth'], 1.0), (c['rho'] * c['Lv'], c['dq'], 1.0), (c['rho'], c['dq'], 1e-6)):
        flux = np.abs(K * D * u * chi)
        out = np.maximum(out, np.abs(K * D) * eq / np.maximum(flux, sc))
    refined = out + eps_sub * worst
    ok = sep & (nw > 0) & np.isfinite(refined)
    return np.where(c['wet'], np.where(ok, np.minimum(refined, worst), worst), 0), sep


def refined_report(name, fluxes):
    c = setup(fluxes); wet = c['wet']
    us_r, chi_r, n_r = reference_path(c)
    xs = fixed_point(c, np.array([us_r, chi_r]))
    J = jac_fd(c, xs)
    e_ref = flux_err(c, (us_r, chi_r), (xs[0], xs[1]))
    w = bound_from_J(c, J, xs)
    for r_pow in (1.25, 1.5, 2.0):
        b, sep = bound_refined(c, J, xs, r_pow=r_pow, eps_sub=(1e-6) ** (r_pow - 1))
        viol = (e_ref > b)[wet].sum()
        print('%s r_pow %.2f: separated %.1f%% of wet cells; violations (actual > refined bound) %d, worst actual/bound %.3f' % (name, r_pow, 100 * sep[wet].mean(), viol, (e_ref / np.maximum(b, 1e-300))[wet].max()))
        for budget in (2e-7, 4e-7, 8e-7):
            print('    budget %.0e: flagged worst-case %.3f%%  refined %.4f%%  (x1.25 safety: %.4f%%); actual beyond: %.1f ppm' % (budget, 100 * (w > budget)[wet].mean(), 100 * (b > budget)[wet].mean(), 100 * (1.25 * b > budget)[wet].mean(), 1e6 * (e_ref > budget)[wet].mean()))


if __name__ == '__main__':
    mode = sys.argv[2] if len(sys.argv) > 2 else 'bound'
    cfgs = (('default', ic.SimilarityTheoryFluxes), ('corrected', ic.corrected_atmosphere_ocean_fluxes))
    for name, mk in cfgs:
        if mode == 'bound': study_bound(name, mk())
        elif mode == 'anatomy': flag_anatomy(name, mk())
        elif mode == 'proto': proto_report(name, mk())
        elif mode == 'proto32': proto32_report(name, mk())
        elif mode == 'refined': refined_report(name, mk())
