"""Second, independently written restatement of the flux path in vectorised NumPy.
TEST INFRASTRUCTURE ONLY (cross-checks oracle/coflux_oracle.c and generates tests/golden/*).
PARITY UNPINNED — see the header of coflux_oracle.c: the reference's arithmetic lives in the
un-vendored NumericalEarth.jl (Project.toml:21,31-32); parameters are anchored on
src/OMIPConfigurations/omip_simulation.jl:40-113 and atmosphere.jl:41-44.

Shares no code with the C oracle: parameters are read from the Python configuration objects
(coflux.interface_computations), not from the C struct, and every formula is typed again.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------
# thermodynamics
# ---------------------------------------------------------------------------------------------
class Thermo:
    def __init__(self, th):
        self.Rd = th.gas_constant / th.dry_air_molar_mass
        self.Rv = th.gas_constant / th.water_molar_mass
        self.eps = th.dry_air_molar_mass / th.water_molar_mass
        self.cpd = self.Rd / th.dry_air_adiabatic_exponent
        self.cpv, self.cpl, self.cpi = (th.water_vapor_heat_capacity, th.liquid_water_heat_capacity,
                                        th.water_ice_heat_capacity)
        self.Lv0, self.Ls0 = th.reference_vaporization_enthalpy, th.reference_sublimation_enthalpy
        self.T0, self.Ttr, self.ptr = th.reference_temperature, th.triple_point_temperature, th.triple_point_pressure
        self.Tf, self.Tin, self.pw = th.water_freezing_temperature, th.total_ice_nucleation_temperature, th.ice_nucleation_power

    def liquid_fraction(self, T):
        ramp = np.clip((T - self.Tin) / (self.Tf - self.Tin), 0.0, None) ** self.pw
        return np.where(T > self.Tf, 1.0, np.where(T > self.Tin, ramp, 0.0))

    def svp(self, T, L0, dcp):
        return self.ptr * (T / self.Ttr) ** (dcp / self.Rv) * np.exp(
            (L0 - dcp * self.T0) / self.Rv * (1.0 / self.Ttr - 1.0 / T))

    def svp_liquid(self, T):
        return self.svp(T, self.Lv0, self.cpv - self.cpl)

    def svp_equil(self, T):
        lam = self.liquid_fraction(T)
        return self.svp(T, lam * self.Lv0 + (1 - lam) * self.Ls0,
                        lam * (self.cpv - self.cpl) + (1 - lam) * (self.cpv - self.cpi))

    def Rm(self, q, qc):
        return self.Rd * (1.0 + (self.eps - 1.0) * q - self.eps * qc)

    def state_pTq(self, p, T, q):
        q = np.clip(q, 0.0, 1.0)
        pvs = self.svp_equil(T)
        tiny = np.finfo(np.float64).eps
        with np.errstate(divide="ignore", invalid="ignore"):
            qvs = np.where(p - pvs >= tiny, self.Rd / self.Rv * (1 - q) * pvs / (p - pvs), 1.0 / tiny)
        qc = np.maximum(q - qvs, 0.0)
        rho = p / (self.Rm(q, qc) * T)
        return dict(rho=rho, p=p, T=T, q=q)

    def partition(self, s):
        qvs = self.svp_equil(s["T"]) / (s["rho"] * self.Rv * s["T"])
        qc = np.maximum(s["q"] - qvs, 0.0)
        lam = self.liquid_fraction(s["T"])
        return lam * qc, (1 - lam) * qc

    def cp_m(self, s):
        ql, qi = self.partition(s)
        return self.cpd + (self.cpv - self.cpd) * s["q"] + (self.cpl - self.cpv) * ql + (self.cpi - self.cpv) * qi

    def q_vapor(self, s):
        ql, qi = self.partition(s)
        return np.maximum(0.0, s["q"] - ql - qi)

    def T_virtual(self, s):
        ql, qi = self.partition(s)
        return self.Rm(s["q"], ql + qi) / self.Rd * s["T"]

    def Lv(self, T):
        return self.Lv0 + (self.cpv - self.cpl) * (T - self.T0)


def water_mole_fraction(sw, S):
    s = S / 1000.0
    a = s / (1.0 - s)
    inv = sum(f / m for f, m in zip(sw.constituent_mass_fraction, sw.constituent_molar_mass))
    iw = 1.0 / sw.water_molar_mass
    return iw / (iw + a * inv)


# ---------------------------------------------------------------------------------------------
# stability functions
# ---------------------------------------------------------------------------------------------
def _paulson_m(zm, c=16.0):
    x = (1.0 - c * zm) ** 0.25
    return 2 * np.log((1 + x) / 2) + np.log((1 + x * x) / 2) - 2 * np.arctan(x) + np.pi / 2


def _convective(zm, c):
    y = np.cbrt(1.0 - c * zm)
    return 1.5 * np.log((1 + y + y * y) / 3) - np.sqrt(3) * np.arctan((1 + 2 * y) / np.sqrt(3)) + np.pi / np.sqrt(3)


def psi_m(name, z):
    zm, zp = np.minimum(z, 0.0), np.maximum(z, 0.0)
    if name == "edson2013":
        dz = np.minimum(50.0, 0.35 * zp)
        st = -0.7 * zp - 0.75 * (zp - 5 / 0.35) * np.exp(-dz) - 0.75 * 5 / 0.35
        f = zm ** 2 / (1 + zm ** 2)
        un = (1 - f) * _paulson_m(zm, 15.0) + f * _convective(zm, 10.15)
    elif name == "sheba":
        a, b = 5.0, 5.0 / 6.5
        B = np.cbrt((1 - b) / b)
        x = np.cbrt(1 + zp)
        r3 = np.sqrt(3.0)
        st = -3 * a / b * (x - 1) + a * B / (2 * b) * (
            2 * np.log((x + B) / (1 + B)) - np.log((x * x - x * B + B * B) / (1 - B + B * B))
            + 2 * r3 * (np.arctan((2 * x - B) / (r3 * B)) - np.arctan((2 - B) / (r3 * B))))
        un = _paulson_m(zm)
    else:
        st = -5.0 * zp
        un = _paulson_m(zm)
    return np.where(z < 0, un, st)


def psi_h(name, z):
    zm, zp = np.minimum(z, 0.0), np.maximum(z, 0.0)
    if name == "edson2013":
        dz = np.minimum(50.0, 0.35 * zp)
        st = -(1 + 2 / 3 * zp) ** 1.5 - 2 / 3 * (zp - 14.28) * np.exp(-dz) - 8.525
        f = zm ** 2 / (1 + zm ** 2)
        x = np.sqrt(1 - 15 * zm)
        un = (1 - f) * 2 * np.log((1 + x) / 2) + f * _convective(zm, 34.15)
    elif name == "sheba":
        a, b, c = 5.0, 5.0, 3.0
        B = np.sqrt(c * c - 4)
        st = -b / 2 * np.log(1 + c * zp + zp * zp) + (-a / B + b * c / (2 * B)) * (
            np.log((2 * zp + c - B) / (2 * zp + c + B)) - np.log((c - B) / (c + B)))
        un = 2 * np.log((1 + np.sqrt(1 - 16 * zm)) / 2)
    else:
        st = -5.0 * zp
        un = 2 * np.log((1 + np.sqrt(1 - 16 * zm)) / 2)
    return np.where(z < 0, un, st)


# ---------------------------------------------------------------------------------------------
# roughness
# ---------------------------------------------------------------------------------------------
def _nu(visc, T):
    kind, c = visc.coefficients()
    if kind == 0:
        return c[0] + 0 * T
    Tc = T - 273.15
    return c[0] + c[1] * Tc + c[2] * Tc ** 2 + c[3] * Tc ** 3


def momentum_length(r, g, us, U, Ts):
    if isinstance(r, (int, float)):
        return float(r) + 0 * us
    wf = r.wave_formulation
    alpha = wf if isinstance(wf, (int, float)) else np.maximum(wf.minimum, wf.a1 * np.minimum(U, wf.umax) + wf.a2)
    nu = _nu(r.air_kinematic_viscosity, Ts)
    lm = r.maximum_roughness_length
    with np.errstate(divide="ignore", invalid="ignore"):
        lR = np.where(us == 0, lm, r.laminar_parameter * nu / us)
    return np.minimum(alpha * us * us / g + lR, lm)


def scalar_length(r, lu, us, Ts):
    if isinstance(r, (int, float)):
        return float(r) + 0 * us
    nu = _nu(r.air_kinematic_viscosity, Ts)
    lm = r.maximum_roughness_length
    R = lu * us / nu
    s = r.reynolds_number_scaling_function
    with np.errstate(divide="ignore", invalid="ignore"):
        lq = np.where(R == 0, 0.0, s.A / R ** s.b)
    lq = np.where(us == 0, lm, lq)
    return np.minimum(lq, lm)


def _wind_speed_scale(fluxes, Jb, dU2, h_bl):
    """U of the similarity profiles: |Δu|² + U_G² with U_G = max(β w★, U_G,min), or — shear-aware form, launch.sh:67-72 —
    U_G² = (β w★)² + (c |Δu|)² + U_G,min² when the formulation carries a shear_gustiness_coefficient c > 0."""
    wstar = fluxes.gustiness_parameter * np.cbrt(np.maximum(Jb, 0.0) * h_bl)
    c = getattr(fluxes, "shear_gustiness_coefficient", 0.0)
    if c > 0.0:
        return np.sqrt(dU2 * (1.0 + c * c) + wstar * wstar + fluxes.minimum_gustiness ** 2)
    return np.sqrt(dU2 + np.maximum(wstar, fluxes.minimum_gustiness) ** 2)


# ---------------------------------------------------------------------------------------------
# the solver
# ---------------------------------------------------------------------------------------------
def atmosphere_ocean_fluxes(fluxes, ocean, atmos, *, hx, hy, ring, thermodynamics, seawater,
                            ocean_properties, velocity_difference="relative", h=10.0, h_bl=600.0,
                            g=9.81):
    """ocean/atmos: dicts of halo-inclusive 2-D arrays.  Returns dict of halo-inclusive arrays
    (zeros outside the computed interior+ring window)."""
    from coflux import interface_computations as ic
    th = Thermo(thermodynamics)
    ny, nx = ocean["T"].shape[0] - 2 * hy, ocean["T"].shape[1] - 2 * hx
    js, je, is_, ie = hy - ring, hy + ny + ring, hx - ring, hx + nx + ring
    W = (slice(js, je), slice(is_, ie))
    E = (slice(js, je), slice(is_ + 1, ie + 1))
    N = (slice(js + 1, je + 1), slice(is_, ie))

    uo = 0.5 * (ocean["u"][W] + ocean["u"][E])
    vo = 0.5 * (ocean["v"][W] + ocean["v"][N])
    Ts = ocean["T"][W] + ocean_properties.temperature_offset
    So = ocean["S"][W]
    wet = ocean["mask"][W] != 0 if ocean.get("mask") is not None else np.ones(Ts.shape, bool)
    ua, va, Ta, pa, qa = (atmos[k][W] for k in ("u", "v", "T", "p", "q"))

    A = th.state_pTq(pa, Ta, qa)
    qs = water_mole_fraction(seawater, So) * th.svp_liquid(Ts) / (A["rho"] * th.Rv * Ts)
    dq = th.q_vapor(A) - qs
    dth = Ta + g * h / th.cp_m(A) - Ts
    if velocity_difference == "relative":
        du, dv = ua - uo, va - vo
    else:
        du, dv = ua + 0 * uo, va + 0 * vo
    Sfc = th.state_pTq(pa, Ts, qs)
    Tv, qv = th.T_virtual(Sfc), th.q_vapor(Sfc)
    delta = th.eps - 1.0
    kap = fluxes.von_karman_constant
    if isinstance(fluxes, ic.CoefficientBasedFluxes):
        us, ts, qq, its = _large_yeager(fluxes, du, dv, dth, dq, Tv, qv, delta, kap, g, h, wet)
        return _pack_fluxes(th, A, Ta, Ts, du, dv, us, ts, qq, its, wet, ocean, ocean_properties, W)
    stab = fluxes.stability_functions.name
    coare = isinstance(fluxes.similarity_form, ic.COARELogarithmicSimilarityProfile)
    sc = fluxes.solver_stop_criteria
    fixed = isinstance(sc, ic.FixedIterations)
    maxit = sc.iterations if fixed else sc.maxiter

    us = np.full(Ts.shape, 1e-4)
    ts = us.copy()
    qq = us.copy()
    its = np.zeros(Ts.shape, np.int32)
    active = np.ones(Ts.shape, bool) if fixed else wet.copy()
    dU = np.sqrt(du * du + dv * dv)
    it = 0
    while active.any() and it < maxit:
        b = g / Tv * (ts * (1 + delta * qv) + delta * Tv * qq)
        Jb = -us * b
        U = _wind_speed_scale(fluxes, Jb, du * du + dv * dv, h_bl)
        lu = momentum_length(fluxes.momentum_roughness_length, g, us, dU, Ts)
        lq = scalar_length(fluxes.water_vapor_roughness_length, lu, us, Ts)
        lt = scalar_length(fluxes.temperature_roughness_length, lu, us, Ts)
        with np.errstate(divide="ignore", invalid="ignore"):
            L = np.where(b == 0, np.inf, us * us / (kap * b))  # L★ = u★²/(κ b★): b★ < 0 ⇒ unstable

        def prof(psi, l):
            r = np.log(h / l) - psi(stab, h / L)
            r = r if coare else r + psi(stab, l / L)
            return np.maximum(r, fluxes.similarity_profile_floor)

        nus = kap / prof(psi_m, lu) * U
        nts = kap / prof(psi_h, lt) * dth
        nqs = kap / prof(psi_h, lq) * dq
        drift = np.abs(nus - us) + np.abs(nts - ts) + np.abs(nqs - qq)
        us = np.where(active, nus, us)
        ts = np.where(active, nts, ts)
        qq = np.where(active, nqs, qq)
        its += active
        it += 1
        if not fixed:
            active = active & ~(drift < sc.tolerance)

    dU = None
    return _pack_fluxes(th, A, Ta, Ts, du, dv, us, ts, qq, its, wet, ocean, ocean_properties, W)


def _large_yeager(fluxes, du, dv, dth, dq, Tv, qv, delta, kap, g, h, wet):
    """Large & Yeager (2004, 2009) / NCAR ncar_ocean_fluxes on this package's Δθ, Δq and buoyancy scale."""
    tc = fluxes.transfer_coefficients
    n = fluxes.solver_stop_criteria.iterations

    def cdn10(u):
        poly = (tc.cd[0] / u + tc.cd[1] + tc.cd[2] * u + tc.cd[3] * u ** 6) * 1e-3
        return np.where(u >= tc.high_wind, tc.cd_high * 1e-3, poly)

    U = np.maximum(np.sqrt(du * du + dv * dv), tc.minimum_wind)
    cdn = cdn10(U)
    rt = np.sqrt(cdn)
    cd = cdn
    ce = tc.ce * rt * 1e-3
    ch = np.where(dth > 0, tc.ch_stable, tc.ch_unstable) * rt * 1e-3
    lz = np.log(h / 10.0)
    for _ in range(n):
        cr = np.sqrt(cd)
        us, ts, qq = cr * U, ch / cr * dth, ce / cr * dq
        b = g / Tv * (ts * (1 + delta * qv) + delta * Tv * qq)
        z = kap * b * h / (us * us)
        z = np.sign(z) * np.minimum(np.abs(z), tc.zeta_bound)
        pm, ph = psi_m("large_yeager", z), psi_h("large_yeager", z)
        u10 = U / (1 + rt * (lz - pm) / kap)
        cdn = cdn10(u10)
        rt = np.sqrt(cdn)
        cen = tc.ce * rt * 1e-3
        chn = np.where(z > 0, tc.ch_stable, tc.ch_unstable) * rt * 1e-3
        cd = cdn / (1 + rt * (lz - pm) / kap) ** 2
        xx = (lz - ph) / kap
        r = np.sqrt(cd / cdn)
        ch = chn / (1 + chn * xx / rt) * r
        ce = cen / (1 + cen * xx / rt) * r
    cr = np.sqrt(cd)
    return cr * U, ch / cr * dth, ce / cr * dq, np.full(du.shape, n, np.int32)


def atmosphere_sea_ice_fluxes(fluxes, iprops, ice, ocean, atmos, *, hx, hy, ring, thermodynamics, ocean_properties,
                              velocity_difference="relative", h=10.0, h_bl=600.0, g=9.81, sigma=5.67e-8):
    """Atmosphere–sea-ice interface with SkinTemperature(ConductiveFlux); second, independent restatement."""
    from coflux import interface_computations as ic
    th = Thermo(thermodynamics)
    ny, nx = ocean["T"].shape[0] - 2 * hy, ocean["T"].shape[1] - 2 * hx
    W = (slice(hy - ring, hy + ny + ring), slice(hx - ring, hx + nx + ring))
    Ti = iprops.freshwater_melting_temperature - iprops.liquidus_slope * ocean["S"][W]  # ice bottom on the liquidus
    wet = ocean["mask"][W] != 0 if ocean.get("mask") is not None else np.ones(Ti.shape, bool)
    ua, va, Ta, pa, qa, Qs, Ql = (atmos[k][W] for k in ("u", "v", "T", "p", "q", "Qs", "Ql"))
    ui = ice["u"][W] if ice.get("u") is not None else 0.0
    vi = ice["v"][W] if ice.get("v") is not None else 0.0
    alb = ice["albedo"][W] if ice.get("albedo") is not None else iprops.albedo
    hi = ice["thickness"][W]
    Ts = ice["top_temperature"][W] + iprops.temperature_offset
    A = th.state_pTq(pa, Ta, qa)
    rho, cp, qav = A["rho"], th.cp_m(A), th.q_vapor(A)
    Ls = th.Ls0 + (th.cpv - th.cpi) * (Ta - th.T0)
    Tm = iprops.freshwater_melting_temperature
    heff = np.maximum(hi, iprops.consolidation_thickness)
    if velocity_difference == "relative":
        du, dv = ua - ui, va - vi
    else:
        du, dv = ua + 0 * Ti, va + 0 * Ti
    dU = np.sqrt(du * du + dv * dv)
    delta, kap = th.eps - 1.0, fluxes.von_karman_constant
    stab = fluxes.stability_functions.name
    coare = isinstance(fluxes.similarity_form, ic.COARELogarithmicSimilarityProfile)
    sc = fluxes.solver_stop_criteria
    fixed = isinstance(sc, ic.FixedIterations)
    maxit = sc.iterations if fixed else sc.maxiter
    us = np.full(Ti.shape, 1e-4)
    ts, qq = us.copy(), us.copy()
    its = np.zeros(Ti.shape, np.int32)
    active = np.ones(Ti.shape, bool) if fixed else wet.copy()
    it = 0
    while active.any() and it < maxit:
        Qnet = -rho * Ls * us * qq + iprops.emissivity * sigma * Ts ** 4 - rho * cp * us * ts \
            - (1 - alb) * Qs - iprops.emissivity * Ql
        Tstar = Ti - Qnet * heff / iprops.conductivity
        if getattr(iprops, "skin_temperature_scheme", 0) == 1:   # semi-implicit: longwave linearised about the previous Ts
            rest = Qnet - iprops.emissivity * sigma * Ts ** 4
            Tstar = (Ti - rest * heff / iprops.conductivity) / (1 + heff / iprops.conductivity * iprops.emissivity * sigma * Ts ** 3)
        Tn = np.minimum(Ts + np.clip(Tstar - Ts, -iprops.maximum_temperature_change, iprops.maximum_temperature_change), Tm)
        qs = th.svp(Tn, th.Ls0, th.cpv - th.cpi) / (rho * th.Rv * Tn)
        dq, dth = qav - qs, Ta + g * h / cp - Tn
        S = th.state_pTq(pa, Tn, qs)
        Tv, qv = th.T_virtual(S), th.q_vapor(S)
        b = g / Tv * (ts * (1 + delta * qv) + delta * Tv * qq)
        U = _wind_speed_scale(fluxes, -us * b, du * du + dv * dv, h_bl)
        lu = momentum_length(fluxes.momentum_roughness_length, g, us, dU, Tn)
        lq = scalar_length(fluxes.water_vapor_roughness_length, lu, us, Tn)
        lt = scalar_length(fluxes.temperature_roughness_length, lu, us, Tn)
        with np.errstate(divide="ignore", invalid="ignore"):
            L = np.where(b == 0, np.inf, us * us / (kap * b))

            def prof(psi, l):
                r = np.log(h / l) - psi(stab, h / L)
                r = r if coare else r + psi(stab, l / L)
                return np.maximum(r, fluxes.similarity_profile_floor)

            nus, nts, nqs = kap / prof(psi_m, lu) * U, kap / prof(psi_h, lt) * dth, kap / prof(psi_h, lq) * dq
        drift = np.abs(nus - us) + np.abs(nts - ts) + np.abs(nqs - qq)
        us, ts, qq, Ts = (np.where(active, n_, o_) for n_, o_ in ((nus, us), (nts, ts), (nqs, qq), (Tn, Ts)))
        its += active
        it += 1
        if not fixed:
            active = active & ~(drift < sc.tolerance)
    zero = ~wet
    us, ts, qq = (np.where(zero, 0.0, a) for a in (us, ts, qq))
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(dU == 0, 0.0, -us * us * du / dU)
        ty = np.where(dU == 0, 0.0, -us * us * dv / dU)
    res = dict(sensible_heat=-rho * cp * us * ts, latent_heat=-rho * us * qq * Ls, water_vapor=-rho * us * qq,
               x_momentum=rho * tx, y_momentum=rho * ty,
               temperature=np.where(zero, 0.0, Ts) - iprops.temperature_offset,
               friction_velocity=us, temperature_scale=ts, humidity_scale=qq)
    out = {}
    for k, a in res.items():
        full = np.zeros(ocean["T"].shape)
        full[W] = np.where(a == 0, 0.0, a)
        out[k] = full
    full = np.zeros(ocean["T"].shape, np.int32)
    full[W] = its
    out["iterations"] = full
    return out


def _pack_fluxes(th, A, Ta, Ts, du, dv, us, ts, qq, its, wet, ocean, ocean_properties, W):
    dU = np.sqrt(du * du + dv * dv)
    zero = ~wet
    us, ts, qq = (np.where(zero, 0.0, a) for a in (us, ts, qq))
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(dU == 0, 0.0, -us * us * du / dU)
        ty = np.where(dU == 0, 0.0, -us * us * dv / dU)
    rho, cp, Lv = A["rho"], th.cp_m(A), th.Lv(Ta)
    res = dict(sensible_heat=-rho * cp * us * ts, latent_heat=-rho * us * qq * Lv,
               water_vapor=-rho * us * qq, x_momentum=rho * tx, y_momentum=rho * ty,
               temperature=np.where(zero, 0.0, Ts) - ocean_properties.temperature_offset,
               friction_velocity=us, temperature_scale=ts, humidity_scale=qq)
    out = {}
    for k, a in res.items():
        full = np.zeros(ocean["T"].shape)
        full[W] = np.where(a == 0, 0.0, a)  # normalise -0.0
        out[k] = full
    full = np.zeros(ocean["T"].shape, np.int32)
    full[W] = its
    out["iterations"] = full
    return out


# ---------------------------------------------------------------------------------------------
# interpolation and net fluxes
# ---------------------------------------------------------------------------------------------
def interpolate_atmosphere_state(src, fi2d, fj2d, level1, level2, tf, cos_rot=None, sin_rot=None):
    """fi2d/fj2d: 2-D fractional indices of the window to fill (already restricted)."""
    nsx = src["tas"].shape[2]
    nsy = src["tas"].shape[1]
    ti, tj = np.trunc(fi2d), np.trunc(fj2d)
    xi, eta = np.mod(fi2d, 1.0), np.mod(fj2d, 1.0)   # ξ = mod(fractional_idx, 1) ∈ [0, 1), also for negative indices
    i0, j0 = ti.astype(np.int64), tj.astype(np.int64)
    i1 = i0 + np.sign(fi2d).astype(np.int64)
    j1 = j0 + np.sign(fj2d).astype(np.int64)
    i0, i1 = np.mod(i0, nsx), np.mod(i1, nsx)
    j0, j1 = np.clip(j0, 0, nsy - 1), np.clip(j1, 0, nsy - 1)

    def one(name):
        vals = []
        for lv in (level1, level2):
            d = src[name][lv].astype(np.float64)
            vals.append((1 - xi) * (1 - eta) * d[j0, i0] + (1 - xi) * eta * d[j1, i0]
                        + xi * (1 - eta) * d[j0, i1] + xi * eta * d[j1, i1])
        return vals[1] * tf + vals[0] * (1 - tf)

    u, v = one("uas"), one("vas")
    if cos_rot is not None:
        u, v = u * cos_rot + v * sin_rot, -u * sin_rot + v * cos_rot
    return dict(u=u, v=v, T=one("tas"), p=one("psl"), q=one("huss"), Qs=one("rsds"), Ql=one("rlds"),
                Mp=one("prra") + one("prsn"))


def net_ocean_fluxes(ocean, atmos, fl, *, hx, hy, ocean_properties, albedo, emissivity=1.0,
                     sigma=5.67e-8, min_salinity=0.0, penetrating=True, ice=None, latitude2d=None, land=None):
    ny, nx = ocean["T"].shape[0] - 2 * hy, ocean["T"].shape[1] - 2 * hx
    C = (slice(hy, hy + ny), slice(hx, hx + nx))
    Wst = (slice(hy, hy + ny), slice(hx - 1, hx + nx - 1))
    Sth = (slice(hy - 1, hy + ny - 1), slice(hx, hx + nx))
    z = np.zeros((ny, nx))
    aice = ice["concentration"] if ice else None
    a_c = aice[C] if ice else z
    a_w = aice[Wst] if ice else z
    a_s = aice[Sth] if ice else z
    wet = (ocean["mask"][C] != 0) if ocean.get("mask") is not None else np.ones((ny, nx), bool)
    So = ocean["S"][C]
    Ts = fl["temperature"][C] + ocean_properties.temperature_offset
    if hasattr(albedo, "diffuse"):
        alb = albedo.diffuse - albedo.direct * np.cos(2 * np.deg2rad(latitude2d[C]))
    else:
        alb = float(albedo)
    Qu = emissivity * sigma * Ts ** 4
    Qal = -emissivity * atmos["Ql"][C]
    Qts = -(1 - alb) * atmos["Qs"][C] * (1 - a_c)
    Qss = 0.0 if penetrating else Qts
    SQ = (Qu + fl["sensible_heat"][C] + fl["latent_heat"][C] + Qal) * (1 - a_c) + Qss
    rfi = 1.0 / ocean_properties.freshwater_density
    roi = 1.0 / ocean_properties.reference_density
    co = ocean_properties.heat_capacity
    SF = -atmos["Mp"][C] * rfi + fl["water_vapor"][C] * rfi
    SFs = np.where((So < min_salinity) & (SF < 0), 0.0, SF)
    SFl = -(land[C] if land is not None else z) * rfi             # JRA55PrescribedLand: rivers + calving, not ice-masked
    SFls = np.where((So < min_salinity) & (SFl < 0), 0.0, SFl)
    Qio = ice["interface_heat"][C] if ice else z
    Jsio = ice["salt_flux"][C] if ice else z
    txio = ice["x_stress"][C] if ice else z
    tyio = ice["y_stress"][C] if ice else z
    txao = 0.5 * (fl["x_momentum"][Wst] + fl["x_momentum"][C]) * roi
    tyao = 0.5 * (fl["y_momentum"][Sth] + fl["y_momentum"][C]) * roi
    ax, ay = 0.5 * (a_w + a_c), 0.5 * (a_s + a_c)
    res = dict(u=(1 - ax) * txao + ax * txio, v=(1 - ay) * tyao + ay * tyio,
               T=SQ * roi / co + Qio * roi / co, S=(1 - a_c) * (-So * SFs) + Jsio + (-So * SFls),
               shortwave_surface_flux=Qts * roi / co, upwelling_longwave=Qu,
               downwelling_longwave=-Qal, downwelling_shortwave=-Qts)
    out = {}
    for k, a in res.items():
        full = np.zeros(ocean["T"].shape)
        full[C] = np.where(wet, a, 0.0)
        out[k] = full
    return out


# ---------------------------------------------------------------------------------------------
# sea-ice albedo (CCSM3) and the three-equation ice–ocean exchange — second, independently typed restatements
# ---------------------------------------------------------------------------------------------
def sea_ice_albedo(hi, hs, Ts, *, ice=(0.78, 0.36), snow=(0.98, 0.70), ocean=0.06, h_ref=0.3, dT=1.0, d_ice=0.075,
                   d_snow=(0.10, 0.15), patch=0.02, visible_fraction=0.5, T_melt=0.0):
    """SeaIceAlbedo(hi, hs, Ts): Briegleb et al. 2004 / CICE `ccsm3`."""
    hs = np.zeros_like(hi) if hs is None else hs
    fh = np.minimum(np.arctan(4 * hi) / np.arctan(4 * h_ref), 1.0)
    fT = np.minimum((T_melt - Ts) / dT - 1.0, 0.0)
    bands = []
    for a_i, a_s, ds in zip(ice, snow, d_snow):
        bare = np.maximum(a_i * fh + ocean * (1 - fh) + d_ice * fT, ocean)
        cover = np.where(hs > 0, hs / (hs + patch), 0.0)
        bands.append(bare * (1 - cover) + (a_s + ds * fT) * cover)
    return visible_fraction * bands[0] + (1 - visible_fraction) * bands[1]


def sea_ice_ocean_fluxes(To, So, conc, tau_x_c, tau_y_c, *, rho_o=1026.0, c_o=3991.86795711963, alpha_h=0.0095,
                         alpha_s=0.0095 / 35, us_min=0.0, L=334000.0, S_i=4.0, m=0.054, dz=10.0, dt=0.0):
    """Three-equation interface model + frazil on cell-centre inputs (kinematic stress components at the centre).
    Returns Q_io, Js_io, Q_frazil, u★, S_b."""
    Tf = -m * So
    frz = (To < Tf) & (dt > 0)
    Qfr = np.where(frz, rho_o * c_o * dz * (To - Tf) / (dt if dt > 0 else 1.0), 0.0)
    T = np.where(frz, Tf, To)
    us = np.maximum(np.sqrt(np.hypot(tau_x_c, tau_y_c)), us_min)
    g = c_o * alpha_h / L
    # roots of g·m·Sb² + (g·T − g·m·S_i + α_s)·Sb − (g·T·S_i + α_s·So) = 0 via numpy's polynomial solver per cell is slow;
    # the closed form with the numerically stable choice of root:
    A, B, Cc = g * m, g * T - g * m * S_i + alpha_s, g * T * S_i + alpha_s * So
    disc = np.sqrt(B * B + 4 * A * Cc)
    Sb = np.where(B >= 0, 2 * Cc / (B + disc), (-B + disc) / (2 * A))
    Tb = -m * Sb
    ice = conc > 0
    Qio = np.where(ice, conc * rho_o * c_o * alpha_h * us * (T - Tb), 0.0)
    Js = np.where(ice, conc * alpha_s * us * (So - Sb), 0.0)
    return Qio, Js, Qfr, np.where(ice, us, 0.0), Sb
