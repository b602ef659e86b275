"""oracle/upstream_probe.py with the interpreter faked both ways: a box without Julia reports "absent", a
box with Julia but without the reference reports that, and a box where `using ClimaOcean` works runs the dump script, times the
reference's CPU path and labels its numbers with the reference's version — none of which can happen in the build image."""
import json
import os
import stat
import textwrap

import numpy as np
import pytest

import upstream_probe as up


def _fake_julia(tmp_path, body):
    exe = tmp_path / "julia"
    exe.write_text("#!/bin/bash\n" + textwrap.dedent(body))
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    return str(exe)


@pytest.fixture(autouse=True)
def _clean(monkeypatch):
    up._cache.clear()
    monkeypatch.delenv("COFLUX_JULIA_PROJECT", raising=False)
    yield
    up._cache.clear()


def test_no_julia_is_absent(monkeypatch, tmp_path):
    monkeypatch.delenv("COFLUX_JULIA", raising=False)
    monkeypatch.setenv("PATH", str(tmp_path))          # an empty directory: no `julia`
    info = up.probe()
    assert info["status"] == "absent" and info["julia"] is None
    assert up.time_reference_cpu(dict(ocean={}, src={}), 8, 4, 1) is None
    assert up.ensure_upstream_vectors()["ran"] is False


def test_julia_without_the_reference(monkeypatch, tmp_path):
    monkeypatch.setenv("COFLUX_JULIA", _fake_julia(tmp_path, """
        echo "ERROR: ArgumentError: Package ClimaOcean not found in current path." >&2
        exit 1
    """))
    info = up.probe()
    assert info["status"] == "julia_without_reference" and "ClimaOcean not found" in info["detail"]
    assert up.time_reference_cpu(dict(ocean={}, src={}), 8, 4, 1) is None


def test_julia_with_the_reference_dumps_times_and_labels(monkeypatch, tmp_path):
    upstream = tmp_path / "upstream"
    monkeypatch.setattr(up, "UPSTREAM", str(upstream))
    monkeypatch.setenv("COFLUX_JULIA", _fake_julia(tmp_path, f"""
        for a in "$@"; do last="$a"; done
        case "$*" in
          *oracle_dump.jl*)
            mkdir -p {upstream}
            python3 -c "import numpy as np; np.save('{upstream}/default_sensible_heat.npy', np.zeros((2, 2)))"
            printf 'ocean_fluxes: ok\\nsea_ice: ok\\n' > {upstream}/STATUS.txt
            printf 'ClimaOcean 0.10.0\\nNumericalEarth 0.8.1\\n' > {upstream}/VERSION.txt ;;
          *reference_cpu_baseline.jl*)
            test -f "$last/ocean_T.npy" -a -f "$last/jra_tas.npy" -a -f "$last/shape.npy" || exit 3
            test "$JULIA_NUM_THREADS" -ge 1 || exit 4
            echo '{{"seconds_per_pass": 0.25, "passes": 7, "threads": 16}}' ;;
          *"using ClimaOcean"*)
            echo "COFLUX_VERSIONS ClimaOcean=0.10.0,NumericalEarth=0.8.1,Oceananigans=0.110.2" ;;
          *) exit 2 ;;
        esac
    """))
    info = up.probe()
    assert info["status"] == "present" and info["versions"]["NumericalEarth"] == "0.8.1"
    assert up.reference_label(info) == "NumericalEarth 0.8.1 (ClimaOcean 0.10.0, Oceananigans 0.110.2)"
    assert not up.upstream_vectors_present()
    res = up.ensure_upstream_vectors()
    assert res["ran"] and res["ok"] and "sea_ice: ok" in res["status"] and up.upstream_vectors_present()
    assert up.ensure_upstream_vectors()["ran"] is False            # vectors are there: nothing to run again
    nx, ny, h = 8, 4, 1
    case = dict(ocean={k: np.zeros((ny + 2 * h, nx + 2 * h)) for k in ("T", "S", "u", "v", "mask")},
                src={v: np.zeros((2, 320, 640), np.float32) for v in ("tas", "huss", "psl", "uas", "vas", "rlds", "rsds", "prra", "prsn")})
    rec = up.time_reference_cpu(case, nx, ny, h, seconds=1.0)
    assert rec["kind"] == "reference" and rec["cores"] == 16 and rec["unit"] == "cells/s"
    assert rec["value"] == pytest.approx(nx * ny / 0.25) and "NumericalEarth 0.8.1" in rec["sample"]
    json.dumps(rec)                                                   # goes into bench.py's JSON line as is


def test_the_julia_scripts_are_shipped_and_share_their_npy_reader():
    jl = os.path.join(up.ROOT, "climaocean.jl_amd", "julia")
    for name in ("oracle_dump.jl", "reference_cpu_baseline.jl", "npy_io.jl"):
        assert os.path.exists(os.path.join(jl, name)), name
    timer = open(up.TIMER).read()
    for needle in ("OceanSeaIceModel", "PrescribedAtmosphere", "update_state!", "CPU()", "seconds_per_pass", 'include(joinpath(@__DIR__, "npy_io.jl"))'):
        assert needle in timer, needle
    assert 'include(joinpath(@__DIR__, "npy_io.jl"))' in open(up.DUMP).read()
