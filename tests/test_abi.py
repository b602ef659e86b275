"""CPU-side checks of the drop-in boundary: libcoflux.so loads, exports every symbol that
include/coflux.h declares, agrees with the ctypes mirror on struct sizes, and refuses to run
without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from coflux import abi
from coflux import interface_computations as ic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "coflux.h")).read()
    return set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", text))


def test_header_and_mirror_list_the_same_symbols():
    assert _header_symbols() == set(abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.cf_version() == abi.ABI_VERSION


def test_library_exports_no_undeclared_cf_symbol():
    """Every C-linkage `cf_*` symbol libcoflux.so exports — what a binding could bind — is declared in include/coflux.h
    (cf_ensure_chunk_table was exported, documented in INTEGRATION.md and not declared; it was in fact a
    C++-mangled helper then).  C++ helpers shared between the library's translation units are mangled and not part of
    the ABI; none of them may be NAMED like an undeclared entry point that INTEGRATION.md lists."""
    import re
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set()
    for line in out.splitlines():
        name = line.split()[-1]
        if " T " in line and re.match(r"^cf_[a-z0-9_]+$", name):
            exported.add(name)
    allowed = {"cf_debug_phase_read"}  # per-wave stamp reader of the instrumented scratch builds (absent from the product)
    assert exported - _header_symbols() - allowed == set(), exported - _header_symbols() - allowed
    listed = set(re.findall(r"`(cf_[a-z0-9_]+)`", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    types = set(re.findall(r"\b(cf_[a-z0-9_]+)\b(?=\s*[;{]|\s+\w+[;,)\[])", open(os.path.join(ROOT, "include", "coflux.h")).read()))
    functions = {n for n in listed if not n.endswith(("_params", "_fields", "_fluxes", "_surface", "_state", "_source", "_weights", "_schedule", "_ctx", "_grid", "_roughness"))}
    assert functions - types <= _header_symbols(), functions - types - _header_symbols()


def test_struct_layout_matches_library():
    lib = abi.load_library()
    p = abi.FluxParams()
    assert lib.cf_default_flux_params(C.byref(p)) == 0
    assert p.struct_size == C.sizeof(abi.FluxParams)
    # the library's ":default" block equals the Python mirror's defaults field by field
    q = ic.flux_params()
    for name, _ in abi.FluxParams._fields_:
        a, b = getattr(p, name), getattr(q, name)
        if isinstance(a, (C.Structure, C.Array)):
            assert bytes(a) == bytes(b), name
        else:
            assert a == b, name


def test_presets_lower_to_valid_blocks():
    for make in (ic.corrected_atmosphere_ocean_fluxes, ic.corrected_atmosphere_sea_ice_fluxes,
                 ic.ncar_atmosphere_sea_ice_fluxes):
        p = ic.flux_params(make(), velocity_difference=ic.WindVelocity())
        assert p.velocity_difference == abi.VELOCITY_WIND
        assert p.similarity_form == abi.SIMILARITY_COARE_LOGARITHMIC  # omip_simulation.jl:43,65,108
    p = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes())
    assert p.momentum_roughness.kind == abi.ROUGHNESS_WIND_CHARNOCK and p.minimum_gustiness == 0.5
    p = ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes())
    assert (p.momentum_roughness.constant_length, p.temperature_roughness.constant_length) == (5e-4, 5e-5)
    assert p.stability_functions == abi.STABILITY_SHEBA and p.minimum_gustiness == 0.2
    p = ic.flux_params(ic.ncar_atmosphere_sea_ice_fluxes())
    assert p.gustiness_parameter == 0.0 and p.stability_functions == abi.STABILITY_LARGE_YEAGER


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = abi.load_library()
    h = C.c_void_p()
    g = abi.Grid(8, 8, 2, 2, 1, 0)
    p = ic.flux_params()
    rc = lib.cf_create(C.byref(h), 0, C.byref(g), C.byref(p))
    assert rc == -3 and not h.value
    assert b"no CPU backend" in lib.cf_last_error(None)
    from coflux.runtime import CofluxError, FluxContext
    with pytest.raises(CofluxError):
        FluxContext(8, 8, 2, 2, p)


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(abi.CofluxLibraryMissing):
        abi.load_library(str(tmp_path / "libcoflux.so"))


def test_solver_chunk_plan_covers_every_surface():
    """The layered chunk plan of the flux solver (host arithmetic, coflux_solver.hip::plan_chunk_rounds): for any
    surface cost and CU count the rounds cover the whole cost range with whole chunks, chunk sizes never grow along
    the dispatch order, only 256/512/768/1024/1280 wet cells occur, and the 1/4° surface gets 1024 / 768 / 512."""
    lib = abi.load_library()
    out = (C.c_int * 32)()

    def plan(total, cus, forced=0):
        unit = lib.cf_debug_chunk_plan(total, cus, forced, out, 32)
        assert unit == 64
        return [(out[1 + 2 * r], out[2 + 2 * r]) for r in range(out[0])], unit

    rounds, unit = plan(577498 * 64 + 232906, 256)           # 1440×560 synthetic surface: wet·64 + land
    assert [w for w, _ in rounds] == [1024, 768, 512] and all(0 < n <= 256 for _, n in rounds)
    assert sum(n for _, n in rounds) <= 3 * 256
    import random
    rng = random.Random(5)
    for _ in range(2000):
        cus = rng.choice([1, 2, 8, 32, 64, 256, 304])
        total = rng.choice([1, 63, 64, 65, rng.randrange(1, 10 ** 5), rng.randrange(1, 10 ** 7), rng.randrange(1, 2 * 10 ** 9)])
        rounds, unit = plan(total, cus)
        sizes = [w for w, _ in rounds]
        assert rounds and all(w in (256, 512, 768, 1024, 1280) for w in sizes[:-1]) and sizes == sorted(sizes, reverse=True), (total, cus, rounds)
        # (64 / 128 / 192: the last dispatch layer of a surface of one to three 256-cell chunks per CU, cut so that every CU
        # gets at most one piece of it)
        assert sizes[-1] in (256, 512, 768, 1024, 1280) or (sizes == [256, sizes[-1]] and sizes[-1] in (64, 128, 192)
                                                            and rounds[0][1] in (cus, 2 * cus) and rounds[1][1] <= cus + 1), (total, cus, rounds)   # (+ 1: the export counts the end slack of one wet cell as a chunk)
        assert all(n > 0 for _, n in rounds), (total, cus, rounds)
        covered = sum(w * unit * n for w, n in rounds)
        assert covered > total, (total, cus, rounds)                     # every cost prefix falls into some chunk
        assert covered - total <= max(sizes) * unit + unit, (total, cus, rounds)   # and no empty chunk at the end
        if len(rounds) > 1:
            assert all(n <= cus or w == 1024 or (w == 256 and n == 2 * cus) for w, n in rounds[:-1]), (total, cus, rounds)
    # the 1/8 latitude slab of the 1/4° surface (71 803 wet cells): 256 whole workgroups and one-wave pieces for the rest
    rounds, _ = plan(71803 * 64 + 32021, 256)
    assert rounds == [(256, 256), (64, 106)], rounds      # (98 of wet cells + the land cells' share of the cost)
    assert plan(65000 * 64, 256)[0][0][0] == 256 and len(plan(65000 * 64, 256)[0]) == 1      # fits one layer: uniform
    assert [w for w, _ in plan(100000 * 64, 256)[0]] == [256, 192]
    assert plan(143600 * 64, 256)[0] == [(256, 512), (64, 196)]     # the quarter slab: two whole layers, then pieces
    rounds, _ = plan(10 ** 6, 256, forced=512)
    assert len(rounds) == 1 and rounds[0][0] == 512
    # AO_PLAN_TAIL (-2): the plan of a launch that carries tail workgroups — three EQUAL chunks per CU exactly where the surface
    # needs all three arrival layers (its workgroups then retire staggered); every other size keeps the automatic plan
    full = 577498 * 64 + 232906
    rounds, _ = plan(full, 256, forced=-2)
    assert [w for w, _ in rounds] == [768, 768, 768] and sum(n for _, n in rounds) <= 3 * 256
    for total in (full // 8, full // 2, full * 3 // 4, full * 3):
        assert plan(total, 256, forced=-2)[0] == plan(total, 256)[0], total
    for _ in range(500):
        cus = rng.choice([8, 64, 256, 304])
        total = rng.randrange(1, 2 * 10 ** 9)
        rounds, unit = plan(total, cus, forced=-2)
        covered = sum(w * unit * n for w, n in rounds)
        assert rounds and all(n > 0 for _, n in rounds) and covered > total and covered - total <= max(w for w, _ in rounds) * unit + unit


def test_loader_refuses_a_library_built_from_other_sources(tmp_path, monkeypatch):
    """cf_build_stamp (sha256 of the sources at build time) against the tree the Python mirror sits in: a stale or foreign
    libcoflux.so — an A/B build left in scratch/, one `LIBCOFLUX=` away from the tests — is refused unless the caller says
    so."""
    lib = abi.load_library()
    assert lib.cf_build_stamp().decode() == abi.source_stamp() and len(abi.source_stamp()) == 16
    monkeypatch.setattr(abi, "source_stamp", lambda: "0123456789abcdef")
    with pytest.raises(abi.CofluxLibraryMissing, match="built from other sources"):
        abi.load_library(abi.LIB_PATH)
    monkeypatch.setenv("COFLUX_ALLOW_STALE_LIBRARY", "1")
    assert abi.load_library(abi.LIB_PATH).cf_version() == abi.ABI_VERSION
