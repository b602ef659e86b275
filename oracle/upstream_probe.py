"""TEST / MEASUREMENT INFRASTRUCTURE — not part of the product (see oracle.py's header).

The upstream pin's switch (SURVEY.md §8c "Conditional upgrade").  The path's
arithmetic lives in NumericalEarth.jl, which the build image cannot run.  This module is what makes the FIRST box that has
Julia + ClimaOcean flip the parity status by itself:

  probe()                   is `julia -e 'using ClimaOcean'` possible here?  "absent" / "julia_without_reference" / "present"
  ensure_upstream_vectors() on "present": run climaocean.jl_amd/julia/oracle_dump.jl (the reference's public API on the committed
                            inputs of tests/golden/upstream_inputs/) into tests/golden/upstream/ — tests/test_upstream_pin.py then
                            stops skipping
  time_reference_cpu()      on "present": time the reference's own CPU() update_state! on bench.py's inputs with
                            JULIA_NUM_THREADS = the host's cores (climaocean.jl_amd/julia/reference_cpu_baseline.jl) — bench.py's
                            cpu_baseline then reads kind = "reference"

tests/conftest.py and bench.py call these; tests/test_upstream_probe.py fakes the interpreter both ways.  COFLUX_JULIA names the
interpreter (default: `julia` on PATH); COFLUX_JULIA_PROJECT adds --project=…."""
import json
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "climaocean.jl_amd", "julia", "oracle_dump.jl")
TIMER = os.path.join(ROOT, "climaocean.jl_amd", "julia", "reference_cpu_baseline.jl")
UPSTREAM = os.path.join(ROOT, "tests", "golden", "upstream")

_cache = {}


def _julia():
    exe = os.environ.get("COFLUX_JULIA") or shutil.which("julia")
    if not exe:
        return None
    cmd = [exe]
    if os.environ.get("COFLUX_JULIA_PROJECT"):
        cmd.append("--project=" + os.environ["COFLUX_JULIA_PROJECT"])
    return cmd


def probe(timeout=900, refresh=False):
    """dict(status, julia, versions, detail).  Cached per interpreter for the process (the first `using` precompiles)."""
    cmd = _julia()
    key = tuple(cmd) if cmd else None
    if not refresh and key in _cache:
        return _cache[key]
    if cmd is None:
        out = dict(status="absent", julia=None, versions=None, detail="no `julia` on PATH (COFLUX_JULIA unset)")
    else:
        code = ('using ClimaOcean; import Pkg; '
                'v = Dict(string(p.name) => string(p.version) for p in values(Pkg.dependencies()) '
                'if p.name in ("ClimaOcean", "NumericalEarth", "Oceananigans", "ClimaSeaIce")); '
                'println("COFLUX_VERSIONS ", join(("$(k)=$(v[k])" for k in sort(collect(keys(v)))), ","))')
        try:
            r = subprocess.run(cmd + ["-e", code], capture_output=True, text=True, timeout=timeout)
            if r.returncode == 0:
                versions = {}
                for line in r.stdout.splitlines():
                    if line.startswith("COFLUX_VERSIONS "):
                        versions = dict(kv.split("=", 1) for kv in line.split(" ", 1)[1].split(",") if "=" in kv)
                out = dict(status="present", julia=cmd[0], versions=versions, detail=None)
            else:
                out = dict(status="julia_without_reference", julia=cmd[0], versions=None,
                           detail=(r.stderr or r.stdout).strip().splitlines()[-1][:300] if (r.stderr or r.stdout).strip() else f"exit code {r.returncode}")
        except Exception as exc:  # noqa: BLE001 — a probe reports, it never raises
            out = dict(status="julia_without_reference", julia=cmd[0], versions=None, detail=f"{type(exc).__name__}: {exc}"[:300])
    _cache[key] = out
    return out


def reference_label(info=None):
    """What a report may say about parity once the vectors exist: 'NumericalEarth 0.8.1 (ClimaOcean 0.10.0)'."""
    v = (info or probe()).get("versions") or {}
    if os.path.exists(os.path.join(UPSTREAM, "VERSION.txt")) and not v:
        v = dict(line.split(" ", 1) for line in open(os.path.join(UPSTREAM, "VERSION.txt")).read().splitlines() if " " in line)
    if not v:
        return None
    lead = "NumericalEarth " + v["NumericalEarth"] if "NumericalEarth" in v else "ClimaOcean " + v.get("ClimaOcean", "?")
    rest = ", ".join(f"{k} {v[k]}" for k in ("ClimaOcean", "Oceananigans") if k in v and not lead.startswith(k))
    return lead + (f" ({rest})" if rest else "")


def upstream_vectors_present():
    return os.path.isdir(UPSTREAM) and any(f.endswith(".npy") for f in os.listdir(UPSTREAM))


def ensure_upstream_vectors(timeout=3600):
    """Runs oracle_dump.jl where the reference can run and the vectors are not there yet.  Returns dict(ran, ok, status, detail):
    `status` is STATUS.txt's content (which sections of the dump succeeded) when the script got that far."""
    if upstream_vectors_present():
        return dict(ran=False, ok=True, status=_read_status(), detail="tests/golden/upstream/ already holds vectors")
    info = probe()
    if info["status"] != "present":
        return dict(ran=False, ok=False, status=None, detail=f"julia_probe: {info['status']}")
    try:
        r = subprocess.run(_julia() + [DUMP], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except Exception as exc:  # noqa: BLE001
        return dict(ran=True, ok=False, status=_read_status(), detail=f"{type(exc).__name__}: {exc}"[:300])
    ok = r.returncode == 0 and upstream_vectors_present()
    return dict(ran=True, ok=ok, status=_read_status(), detail=None if ok else (r.stderr or r.stdout)[-600:])


def _read_status():
    p = os.path.join(UPSTREAM, "STATUS.txt")
    return open(p).read() if os.path.exists(p) else None


def time_reference_cpu(case_np, nx, ny, h, seconds=20.0, timeout=3600):
    """The reference's CPU() architecture on bench.py's inputs (the 1440×560 surface: ocean state, two JRA55 snapshots, ñ = 0.37):
    full update_state! passes for about `seconds`, best pass reported.  Returns the cpu_baseline record (kind = "reference") or
    None where the reference cannot run; never raises."""
    info = probe()
    if info["status"] != "present":
        return None
    import numpy as np
    threads = os.cpu_count() or 1
    with tempfile.TemporaryDirectory(prefix="coflux_ref_") as d:
        try:
            for k in ("T", "S", "u", "v", "mask"):
                np.save(os.path.join(d, f"ocean_{k}.npy"), np.ascontiguousarray(case_np["ocean"][k], dtype=np.float64))
            for v, a in case_np["src"].items():
                np.save(os.path.join(d, f"jra_{v}.npy"), np.ascontiguousarray(a[:2], dtype=np.float64))
            np.save(os.path.join(d, "shape.npy"), np.array([nx, ny, h, 0.37, seconds], dtype=np.float64))
            env = dict(os.environ, JULIA_NUM_THREADS=str(threads))
            r = subprocess.run(_julia() + ["--threads", str(threads), TIMER, d], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
            if r.returncode != 0 or line is None:
                return None
            t = json.loads(line)
            return dict(value=nx * ny / float(t["seconds_per_pass"]), unit="cells/s", cores=int(t.get("threads", threads)), kind="reference",
                        sample=f"{t.get('passes', '?')} update_state! passes of the reference's CPU() architecture over the {nx}x{ny} surface "
                               f"(best pass {float(t['seconds_per_pass']) * 1e3:.1f} ms), JULIA_NUM_THREADS = {t.get('threads', threads)}; "
                               f"{reference_label(info)}; climaocean.jl_amd/julia/reference_cpu_baseline.jl")
        except Exception:  # noqa: BLE001 — a baseline that cannot be taken is reported as absent, not as a crash of the bench
            return None
