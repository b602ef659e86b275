"""gpurun_out/<tag>/pmc_traffic_raw.json + pmc_sq_raw.json (tools/round_profile.sh) → profiles/<round>_pmc_traffic.json and
profiles/<round>_pmc_sq.json, the files bench.py cites.  usage: make_pmc_profiles.py <tag> <round> (e.g. r03a r03)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
raw = json.load(open(os.path.join(ROOT, "gpurun_out", tag, "pmc_traffic_raw.json")))
copy = next((v for k, v in raw.items() if k.startswith("copy_kernel")), None)
kernels = {}
for k, v in raw.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    # gfx950: FETCH_SIZE tallies the 128-B requests of 8/16-B-per-lane streams at 64 B (calibrated on copy_kernel in the same
    # run: 256 MiB copied); the interpolation's 4-B gathers are counted unscaled
    scale = 1 if k.startswith("interpolate") else 2
    kernels[k] = dict(FETCH_SIZE_KB=v["FETCH_SIZE"], WRITE_SIZE_KB=v["WRITE_SIZE"], fetch_scale=scale, launches=v.get("launches"),
                      hbm_bytes_per_launch=(scale * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0)
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (counters only) over `python bench.py --steps 20 --warmup 2 "
        "--repetitions 1 --no-cpu-baseline` (tools/round_profile.sh %s); KB per launch, averaged over all launches of the run. gfx950 "
        "correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies 128-B requests at 64 B — calibrated in the same run on copy_kernel "
        "(256 MiB copied: FETCH_SIZE %s KB, WRITE_SIZE %s KB) — so fetch bytes = 2 x FETCH_SIZE for 8/16-B-per-lane streams; the "
        "interpolation's 4-B gathers are counted unscaled. The counters sit on the fabric side of L2 and include what the 256 MB "
        "Infinity Cache absorbs." % (tag, copy and round(copy["FETCH_SIZE"]), copy and round(copy["WRITE_SIZE"])))
json.dump(dict(note=note, commit=commit, kernels=kernels), open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json"), "w"), indent=1)
sq = json.load(open(os.path.join(ROOT, "gpurun_out", tag, "pmc_sq_raw.json")))
out = {}
for k, v in sq.items():
    if "SQ_INSTS_VALU" not in v:
        continue
    rec = dict(v)
    if v.get("SQ_WAVE_CYCLES"):
        # quad-cycles in which a wave's VALU instruction issues ÷ wave lifetime, × waves per SIMD = SIMD VALU-busy while waves live
        rec["valu_busy"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"] * 3.0
        rec["valu_busy_note"] = "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x 3 waves per SIMD"
    out[k] = rec
json.dump(dict(note="rocprofv3 --pmc SQ passes (counters only) over the same command; per launch, averaged over all launches of the run "
                    "(tools/round_profile.sh %s)" % tag, commit=commit, kernels=out),
          open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_sq.json"), "w"), indent=1)
print("wrote profiles/%s_pmc_traffic.json, profiles/%s_pmc_sq.json at %s" % (rnd, rnd, commit))
