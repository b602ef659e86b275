"""Runs bench.py with the given arguments and prints the headline numbers of its JSON line (scratch)."""
import json, subprocess, sys
if not sys.stdin.isatty() and sys.stdin.read(1):
    sys.exit("bench_brief.py runs bench.py itself: pass bench.py's arguments to it, do not pipe a second bench into it")
out = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1])
r = d["roofline"]
print("ms_per_step %.4f  value %.3e  solver in-schedule %.1f us (frac %.3f)  same-inputs %.1f  sorted-by-hints %s  stages %s" % (
    d["ms_per_step"], d["value"], r["avg_launch_ms"] * 1e3, r["frac"], r.get("avg_launch_ms_back_to_back_same_inputs", 0) * 1e3,
    r.get("avg_launch_ms_batches_sorted_by_trip_hints"), json.dumps({k: (round(v * 1e3, 1) if v else v) for k, v in d["stages_ms"].items()})))
