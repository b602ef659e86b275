"""ctypes wrapper of the CPU oracle (oracle/coflux_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
PARITY UNPINNED — see the header of coflux_oracle.c.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
from coflux import abi  # noqa: E402  (struct layouts of include/coflux.h)

LIB = os.path.join(HERE, "liboracle_coflux.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "coflux_oracle.c")
    hdr = os.path.join(ROOT, "include", "coflux.h")
    stale = (not os.path.exists(LIB)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB) for s in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "-B", "liboracle_coflux.so"],
                              stdout=subprocess.DEVNULL)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        for n in ("oracle_psi_momentum", "oracle_psi_scalar"):
            getattr(lib, n).restype = C.c_double
            getattr(lib, n).argtypes = [C.c_int, C.c_double]
        lib.oracle_saturation_vapor_pressure_liquid.restype = C.c_double
        lib.oracle_saturation_vapor_pressure_liquid.argtypes = [C.POINTER(abi.FluxParams), C.c_double]
        lib.oracle_saturation_vapor_pressure_ice.restype = C.c_double
        lib.oracle_saturation_vapor_pressure_ice.argtypes = [C.POINTER(abi.FluxParams), C.c_double]
        lib.oracle_water_mole_fraction.restype = C.c_double
        lib.oracle_water_mole_fraction.argtypes = [C.POINTER(abi.FluxParams), C.c_double]
        lib.oracle_air_density.restype = C.c_double
        lib.oracle_air_density.argtypes = [C.POINTER(abi.FluxParams)] + [C.c_double] * 3
        lib.oracle_solve_cell.argtypes = ([C.POINTER(abi.FluxParams)] + [C.c_double] * 9 +
                                          [C.c_int, C.POINTER(C.c_double)])
        _lib = lib
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def make_grid(nx, ny, hx, hy, ring=1):
    return abi.Grid(nx, ny, hx, hy, ring, 0)


def _shape(g):
    return (g.ny + 2 * g.hy, g.nx + 2 * g.hx)


def _exchange_struct(d):
    s = abi.ExchangeFields()
    for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp"):
        setattr(s, n, _ptr(d[n]))
    return s


def _ocean_struct(o, keep):
    s = abi.OceanSurface()
    for n in ("T", "S", "u", "v"):
        a = _f64(o[n])
        keep.append(a)
        setattr(s, n, _ptr(a))
    m = o.get("mask")
    if m is not None:
        m = np.ascontiguousarray(m)
        keep.append(m)
        s.mask = _ptr(m)
    return s


def _source_struct(src, level1, level2, tf, keep):
    s = abi.AtmosSource()
    first = None
    for k, name in enumerate(abi.JRA55_VARIABLES):
        a = np.ascontiguousarray(src[name], dtype=np.float32)
        keep.append(a)
        s.data[k] = a.ctypes.data
        first = a
    s.n_levels, s.ns_y, s.ns_x = first.shape
    s.level1, s.level2, s.time_fraction = level1, level2, tf
    return s


def _weights_struct(w, keep):
    s = abi.InterpWeights()
    if w is None:
        return s
    s.separable = 1 if w.get("separable", True) else 0
    for n in ("fi", "fj", "cos_rot", "sin_rot", "latitude"):
        a = w.get(n)
        if a is not None:
            a = _f64(a)
            keep.append(a)
            setattr(s, n, _ptr(a))
    return s


def interpolate_atmosphere_state(g, src, weights, level1=0, level2=1, time_fraction=0.37, out=None):
    lib, keep = load(), []
    if out is None:   # (bench.py's timed CPU leg passes preallocated outputs)
        out = {n: np.zeros(_shape(g)) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
    s = _source_struct(src, level1, level2, time_fraction, keep)
    w = _weights_struct(weights, keep)
    e = _exchange_struct(out)
    rc = lib.oracle_interpolate_atmosphere_state(C.byref(g), C.byref(s), C.byref(w), C.byref(e))
    assert rc == 0
    return out


def compute_atmosphere_ocean_fluxes(g, params, ocean, atmos, nthreads=1, scales=True, out=None):
    lib, keep = load(), []
    names = ["sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature"]
    if scales:
        names += ["friction_velocity", "temperature_scale", "humidity_scale"]
    if out is None:
        out = {n: np.zeros(_shape(g)) for n in names}
        if scales:
            out["iterations"] = np.zeros(_shape(g), np.int32)
    o = _ocean_struct(ocean, keep)
    a = {n: _f64(atmos[n]) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp") if n in atmos}
    for n in ("Qs", "Ql", "Mp"):
        a.setdefault(n, np.zeros(_shape(g)))
    e = _exchange_struct(a)
    f = abi.InterfaceFluxes()
    for n, arr in out.items():
        setattr(f, n, _ptr(arr))
    rc = lib.oracle_compute_atmosphere_ocean_fluxes(C.byref(g), C.byref(params), C.byref(o),
                                                    C.byref(e), C.byref(f), nthreads)
    assert rc == 0
    return out


def compute_net_ocean_fluxes(g, params, ocean, atmos, fluxes, ice=None, weights=None, out=None, land=None):
    lib, keep = load(), []
    land = None if land is None else _f64(land)
    lib.oracle_set_land_freshwater(_ptr(land))
    names = ["u", "v", "T", "S", "shortwave_surface_flux", "upwelling_longwave",
             "downwelling_longwave", "downwelling_shortwave"]
    if out is None:
        out = {n: np.zeros(_shape(g)) for n in names}
    o = _ocean_struct(ocean, keep)
    a = {n: _f64(atmos[n]) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
    e = _exchange_struct(a)
    f = abi.InterfaceFluxes()
    for n in ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature"):
        arr = _f64(fluxes[n])
        keep.append(arr)
        setattr(f, n, _ptr(arr))
    ice_s = None
    if ice is not None:
        ice_s = abi.SeaIceFields()
        for n in ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress"):
            if ice.get(n) is not None:
                arr = _f64(ice[n])
                keep.append(arr)
                setattr(ice_s, n, _ptr(arr))
    w = _weights_struct(weights, keep)
    nf = abi.NetOceanFluxes()
    for n, arr in out.items():
        setattr(nf, n, _ptr(arr))
    rc = lib.oracle_compute_net_ocean_fluxes(C.byref(g), C.byref(params), C.byref(o), C.byref(e),
                                             C.byref(f), C.byref(ice_s) if ice_s else None,
                                             C.byref(w), C.byref(nf))
    lib.oracle_set_land_freshwater(None)
    assert rc == 0
    return out


def interpolate_land_freshwater(g, friver, licalvf, weights, level1=0, level2=1, time_fraction=0.0):
    """JRA55PrescribedLand: friver (+ licalvf) [n_levels, ns_y, ns_x] float32 → one ocean-grid field."""
    lib, keep = load(), []
    s = abi.LandSource()
    fr = np.ascontiguousarray(friver, dtype=np.float32)
    s.friver = fr.ctypes.data
    if licalvf is not None:
        lc = np.ascontiguousarray(licalvf, dtype=np.float32)
        keep.append(lc)
        s.licalvf = lc.ctypes.data
    s.n_levels, s.ns_y, s.ns_x = fr.shape
    s.level1, s.level2, s.time_fraction = level1, level2, time_fraction
    w = _weights_struct(weights, keep)
    out = np.zeros(_shape(g))
    assert lib.oracle_interpolate_land_freshwater(C.byref(g), C.byref(s), C.byref(w), _ptr(out)) == 0
    return out


def compute_atmosphere_sea_ice_fluxes(g, params, ice_params, ice, ocean, atmos):
    """ice: dict with thickness, top_temperature (°C) and optionally u, v, albedo, concentration."""
    lib, keep = load(), []
    names = ["sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature",
             "friction_velocity", "temperature_scale", "humidity_scale"]
    out = {n: np.zeros(_shape(g)) for n in names}
    out["iterations"] = np.zeros(_shape(g), np.int32)
    o = _ocean_struct(ocean, keep)
    a = {n: _f64(atmos[n]) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
    e = _exchange_struct(a)
    st = abi.SeaIceState()
    for n in ("concentration", "thickness", "top_temperature", "u", "v", "albedo", "snow_thickness"):
        if ice.get(n) is not None:
            arr = _f64(ice[n])
            keep.append(arr)
            setattr(st, n, _ptr(arr))
    f = abi.InterfaceFluxes()
    for n, arr in out.items():
        setattr(f, n, _ptr(arr))
    rc = lib.oracle_compute_atmosphere_sea_ice_fluxes(C.byref(g), C.byref(params), C.byref(ice_params), C.byref(st),
                                                      C.byref(o), C.byref(e), C.byref(f))
    assert rc == 0
    return out


def compute_net_sea_ice_fluxes(g, params, ice_params, ice, ocean, atmos, ai_fluxes, frazil=None, interface_heat=None):
    lib, keep = load(), []
    o = _ocean_struct(ocean, keep)
    a = {n: _f64(atmos[n]) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
    e = _exchange_struct(a)
    st = abi.SeaIceState()
    for n in ("concentration", "albedo"):
        if ice.get(n) is not None:
            arr = _f64(ice[n])
            keep.append(arr)
            setattr(st, n, _ptr(arr))
    f = abi.InterfaceFluxes()
    for n in ("sensible_heat", "latent_heat", "temperature"):
        arr = _f64(ai_fluxes[n])
        keep.append(arr)
        setattr(f, n, _ptr(arr))
    fr = None if frazil is None else _f64(frazil)
    ih = None if interface_heat is None else _f64(interface_heat)
    top, bottom = np.zeros(_shape(g)), np.zeros(_shape(g))
    rc = lib.oracle_compute_net_sea_ice_fluxes(C.byref(g), C.byref(params), C.byref(ice_params), C.byref(st), C.byref(o),
                                               C.byref(e), C.byref(f), _ptr(fr), _ptr(ih), _ptr(top), _ptr(bottom))
    assert rc == 0
    return dict(top_heat=top, bottom_heat=bottom)


def sea_ice_albedo(params, thickness, snow_thickness, top_temperature):
    """SeaIceAlbedo(hi, hs, Ts), CCSM3."""
    lib = load()
    hi, Ts = _f64(thickness), _f64(top_temperature)
    hs = None if snow_thickness is None else _f64(snow_thickness)
    out = np.zeros(hi.shape)
    rc = lib.oracle_sea_ice_albedo(C.byref(params), C.c_long(hi.size), _ptr(hi), _ptr(hs), _ptr(Ts), _ptr(out))
    assert rc == 0
    return out


def sea_ice_ocean_fluxes(g, params, ice_ocean_params, ocean, concentration, x_stress=None, y_stress=None):
    """compute_sea_ice_ocean_fluxes!: dict interface_heat, salt_flux, frazil_heat, friction_velocity."""
    lib, keep = load(), []
    o = _ocean_struct(ocean, keep)
    conc = None if concentration is None else _f64(concentration)
    tx = None if x_stress is None else _f64(x_stress)
    ty = None if y_stress is None else _f64(y_stress)
    out = {n: np.zeros(_shape(g)) for n in ("interface_heat", "salt_flux", "frazil_heat", "friction_velocity")}
    rc = lib.oracle_sea_ice_ocean_fluxes(C.byref(g), C.byref(params), C.byref(ice_ocean_params), C.byref(o), _ptr(conc),
                                         _ptr(tx), _ptr(ty), _ptr(out["interface_heat"]), _ptr(out["salt_flux"]),
                                         _ptr(out["frazil_heat"]), _ptr(out["friction_velocity"]))
    assert rc == 0
    return out


def normalize_salinity_flux(g, params, flux, mask, additional=None, area=None):
    """Returns (normalised flux copy, mean)."""
    lib = load()
    lib.oracle_normalize_salinity_flux.restype = C.c_double
    f = _f64(flux).copy()
    add = None if additional is None else _f64(additional)
    ar = None if area is None else _f64(area)
    m = np.ascontiguousarray(mask)
    mean = lib.oracle_normalize_salinity_flux(C.byref(g), C.byref(params), _ptr(f), _ptr(add), _ptr(ar), _ptr(m))
    return f, mean


def solve_cell(params, ua, va, Ta, pa, qa, uo, vo, To, So, wet=1):
    lib = load()
    out = (C.c_double * 10)()
    lib.oracle_solve_cell(C.byref(params), ua, va, Ta, pa, qa, uo, vo, To, So, wet, out)
    keys = ("Qc", "Qv", "Fv", "rho_tau_x", "rho_tau_y", "Ts", "ustar", "theta_star", "q_star", "iterations")
    return dict(zip(keys, list(out)))


def max_threads():
    return load().oracle_max_threads()
