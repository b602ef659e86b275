"""Strong-scaling floor on one GPU: the step time of one rank's latitude slab of the 1440x560 surface for 1, 2, 4, 8
ranks (ny = 560, 280, 140, 70), from bench.py's own timed region (scratch; writes gpurun_out/<tag>/slab_curve.json).
usage: slab_curve.py <tag>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
cfg = sys.argv[2] if len(sys.argv) > 2 else "default"   # flux configuration; "corrected" writes slab_curve_corrected.json
out = {}
for ranks, ny in ((1, 560), (2, 280), (4, 140), (8, 70)):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ny", str(ny), "--no-cpu-baseline", "--no-sorted-pass", "--flux-configuration", cfg], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(ny, "FAILED", r.stderr[-500:]); continue
    d = json.loads(line[-1])
    out[str(ranks)] = dict(ny=ny, ms_per_step=d["ms_per_step"], samples=d["ms_per_step_samples"], stages_ms=d["stages_ms"],
                           solver_ms=d["roofline"]["avg_launch_ms"], solver_path=d["config"].get("solver_path"),
                           solver_paths_ms_per_step=d.get("solver_paths_ms_per_step"))
    print(ranks, ny, d["ms_per_step"], json.dumps(d["stages_ms"]), flush=True)
base = out["1"]["ms_per_step"]
proj = {k: base / v["ms_per_step"] for k, v in out.items()}
res = dict(note="one rank's slab of the 1440x560 surface timed alone on one MI355X (python bench.py --ny N): the step time an N-rank "
                "strong-scaled run cannot beat, BEFORE its halo rows (a peer-direct exchange is one more ~3-5 us kernel per step); "
                "projected speed-up = T(560) / T(560 / N)", slabs=out, projected_speedup_before_halos=proj)
os.makedirs(os.path.join(ROOT, "gpurun_out", tag), exist_ok=True)
res["flux_configuration"] = cfg
json.dump(res, open(os.path.join(ROOT, "gpurun_out", tag, "slab_curve.json" if cfg == "default" else f"slab_curve_{cfg}.json"), "w"), indent=1)
print(json.dumps(proj))
