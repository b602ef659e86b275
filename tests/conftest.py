import os
import sys

# the schedule-invariance tests set the library's experiment options (CF_OPT_AO_CHUNK, CF_OPT_INTERP_TILE_CAP), which it accepts
# only in a process started with this (include/coflux.h); no experiment knob of the environment is set by the suite
os.environ.setdefault("COFLUX_EXPERIMENTS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build libcoflux.so and the oracle once, the
    same way __graft_entry__.build() does.  The product itself never builds or falls back — a missing library stays
    a hard error there (tests/test_abi.py::test_missing_library_is_loud)."""
    lib = os.path.join(ROOT, "climaocean.jl_amd", "csrc", "libcoflux.so")
    orc = os.path.join(ROOT, "oracle", "liboracle_coflux.so")
    _upstream_pin()
    if os.path.exists(lib) and os.path.exists(orc):
        return
    import subprocess
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "climaocean.jl_amd", "csrc")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(orc):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _upstream_pin():
    """The first box with Julia + ClimaOcean flips the parity status by itself: where
    `julia -e 'using ClimaOcean'` works and tests/golden/upstream/ is still empty, oracle_dump.jl runs before collection and
    tests/test_upstream_pin.py compares instead of skipping.  Anywhere else this is one `which julia`."""
    try:
        import upstream_probe
        if upstream_probe.probe()["status"] == "present" and not upstream_probe.upstream_vectors_present():
            res = upstream_probe.ensure_upstream_vectors()
            print(f"[conftest] oracle_dump.jl: ran={res['ran']} ok={res['ok']} {res['detail'] or ''}", file=sys.stderr)
    except Exception as exc:  # noqa: BLE001 — the pin is an upgrade, never a reason for the suite not to start
        print(f"[conftest] upstream probe failed: {exc}", file=sys.stderr)
