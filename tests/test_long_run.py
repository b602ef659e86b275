"""bench.py --days: the flux path stepped through a long run (BASELINE configs[3]/[4]'s run length,
examples/sixth_degree_tripolar_ocean_sea_ice.jl:22,52, examples/one_degree_tripolar_ocean_sea_ice.jl:47) at reduced length:
the clock advances through a repeat-year record that wraps, a sliding window in HBM whose slots are rewritten several
times, ocean states in turn; every check step is compared with the CPU oracle at 1e-9 and hashed against an un-pipelined
host-driven loop over the same steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("grid,size,dt,days,record,every", [
    ("tripolar", ("360", "180"), 1200, 3.0, 8, 72),      # 216 steps of 20 min, 24 snapshot intervals: the record wraps 3 times
    ("latlon", ("360", "120"), 300, 1.0, 3, 144),        # 288 steps of 5 min, 8 intervals through a 3-snapshot record
])
def test_long_run_wraps_the_window_and_matches_the_oracle_and_the_host_loop(grid, size, dt, days, record, every):
    line = _run("--grid", grid, "--nx", size[0], "--ny", size[1], "--flux-configuration", "corrected" if grid == "tripolar" else "default",
                "--days", str(days), "--dt", str(dt), "--record-snapshots", str(record), "--window-slots", "4", "--ocean-states", "3",
                "--check-every", str(every))
    steps = int(round(days * 86400 / dt))
    assert line["steps"] == steps and line["unit"] == "s per simulated day" and line["higher_is_better"] is False
    assert line["simulated_days"] == pytest.approx(days) and line["value"] > 0 and line["flux_only_sypd_ceiling"] > 0
    # the record wrapped at least twice and every window slot was rewritten more than once
    intervals = steps * dt / 10800
    assert intervals / record >= 2 and intervals / 4 >= 2
    assert len(line["checks"]) == steps // every and line["checks"][-1]["step"] == steps - 1
    assert line["parity_ok"] and line["worst_scaled_error_vs_oracle"] <= 1e-9, line["checks"]
    assert line["host_loop"]["hashes_equal_at_every_check"] is True, (line["checks"], line["host_loop"]["checks"])
    assert len({c["sha256"] for c in line["checks"]}) == len(line["checks"])      # the state does change from check to check


def test_long_run_arguments_are_validated_before_any_device_work():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--days", "1", "--dt", "700"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and ("whole number of steps" in r.stderr or "HIP device" in r.stderr)
