#!/usr/bin/env python
"""bench.py — surface-flux hot path throughput on MI355X (BASELINE.json metric).

A "step" is one time_step! of the coupled model's flux path over one synthetic surface: (N > 1: halo rows of
the ocean surface state) → JRA55 interpolation → Monin–Obukhov solve → net ocean fluxes, all inside libcoflux
(cf_time_steps: the step loop runs in C, nothing returns to Python inside the timed region).  The clock advances
every step (JRA55 time fraction += 20 min / 3 h through a 4-snapshot window) and the ocean surface alternates
between two states one step apart, so the solver's trip-count hints are one step old, as in a coupled run.

Workload: BASELINE.json configs[1] — the 1/4° 1440×560 surface, JRA55 atmosphere, SimilarityTheory fluxes +
Radiation, Float64, inputs resident in HBM before the timed region.  `--gpus N` shards THAT surface into N
latitude slabs (strong scaling, the north star's 1/2/4/8-GPU figure; `--scaling weak` gives every rank its own
1440×560 slab instead).  Without torchrun's environment `--gpus N` starts the N ranks itself.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Algorithmic bytes per surface cell (derivation in DESIGN.md §5)
BYTES_AO = 128.0                       # compute_atmosphere_ocean_fluxes!: 80 read + 48 written
BYTES_AO_FUSED = 128.0 + 24.0 + 24.0   # + the cell-local net fluxes in its epilogue: Qs, Qℓ, Mp read; JT, JS, SW written
BYTES_STRESS = 16.0 + 8.0 + 16.0       # the face-stress kernel of the fused form: ρτx, ρτy + mask read; τx, τy written
FP64_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4  # wave-instructions/s: 1024 SIMDs, one FP64 VALU instruction per 4 cycles at the 2.4 GHz spec clock
BYTES_INTERP = 18.3 + 64.0             # JRA55 window amortised + 8 exchange fields written
BYTES_NET = 88.0 + 40.0
SNAPSHOT_INTERVAL, DT = 3 * 3600.0, 20 * 60.0   # JRA55 is 3-hourly; Δt = 20 min (README.md:76)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 1000, median of 5 repetitions)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 50)")
    ap.add_argument("--repetitions", type=int, default=None, help="timed repetitions of K steps; the median is reported")
    ap.add_argument("--latency-layout", choices=("auto", "never", "always"), default="auto",
                    help="CF_OPT_LATENCY_LAYOUT: the exact path's kernels for one or two waves per SIMD (auto: chunk plans of at most two workgroups per CU)")
    ap.add_argument("--ao-chunk", type=int, default=0, help="CF_OPT_AO_CHUNK (0 = the library's plan): wet cells per solver workgroup — an experiment "
                                                            "option, needs COFLUX_EXPERIMENTS=1 in the environment")
    ap.add_argument("--nx", type=int, default=1440)
    ap.add_argument("--ny", type=int, default=560)
    ap.add_argument("--halo", type=int, default=7)  # README.md:58 halo=(7,7,7)
    ap.add_argument("--grid", choices=("latlon", "tripolar"), default="latlon",
                    help="tripolar: BASELINE configs[3]/[4] — general 2-D interpolation weights, wind rotation, and the fold on "
                         "the last rank (use --nx 360 --ny 180 for the 1-degree grid, --nx 2160 --ny 1080 for the 1/6-degree one)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--halo-backend", choices=("auto", "rccl", "peer", "torch"), default="auto")
    ap.add_argument("--halo-in-launch", choices=("auto", "on", "off"), default="auto",
                    help="CF_OPT_HALO_IN_SOLVER_LAUNCH at N > 1 with the peer-direct backend: the step's halo rows as rider workgroups of its "
                         "solver launch.  auto = on if, on this machine, six steps with it leave the bits of six steps with the exchange "
                         "kernel on every rank (halo rows poisoned first); the line says which ran (config.halo_in_solver_launch)")
    ap.add_argument("--flux-configuration", choices=("default", "corrected", "ncar"), default="default")
    ap.add_argument("--config", choices=("ocean", "sea_ice"), default="ocean",
                    help="ocean: BASELINE configs[1]; sea_ice: configs[2] (atmosphere–sea-ice interface + partition)")
    ap.add_argument("--pipeline", choices=("auto", "on", "off", "merged", "tail"), default="auto",
                    help="where the NEXT step's interpolate_atmosphere_state! runs (two sets of exchange fields).  tail = "
                         "CF_OPT_MERGED_PREFETCH 2: tail workgroups of this step's solver launch (with sea ice: of the interface "
                         "solve's, with this step's face stresses), 1440x560 0.0914 -> 0.0864 ms/step (0.0839 once the per-step event was gone); merged = 1: inside this "
                         "step's face-stress launch (-1 %%); on = the auxiliary stream (measured slower: 0.119 vs 0.092); off = "
                         "the un-pipelined three-launch step; auto = tail wherever the solver kernel can carry it (the round-3 "
                         "ocean kernel, CoefficientBasedFluxes), else off")
    ap.add_argument("--net-diagnostics", action="store_true",
                    help="also write the three optional radiation diagnostics of cf_net_ocean_fluxes (24 B/cell beyond "
                         "the 88 + 40 B/cell contract of compute_net_ocean_fluxes!, SURVEY.md §8d)")
    ap.add_argument("--ice-free-cells", choices=("iterate", "zero"), default="iterate",
                    help="--config sea_ice: CF_OPT_ICE_FREE_CELLS — iterate (default: the interface solve runs on every wet cell) or "
                         "zero (opt-in: open water gets zero_interface_state; reported beside the default, never instead of it)")
    ap.add_argument("--ice-orbit-shortcut", type=int, choices=(0, 1), default=1,
                    help="--config sea_ice: CF_OPT_ICE_ORBIT_SHORTCUT (0 = iterate every abandoned cell to maxiter)")
    ap.add_argument("--share-device", action="store_true",
                    help="REHEARSAL of the N-rank code path on a box with fewer devices: every rank uses device 0, the host-side "
                         "collectives go over gloo, and only the halo backends that can run two ranks on one device are "
                         "tried (peer-direct through HIP IPC; RCCL refuses duplicate devices).  The line is marked "
                         "'rehearsal' and is not a scaling number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trip-hints", type=int, choices=(0, 1, 2, 3), default=2,
                    help="CF_OPT_TRIP_HINTS for the timed region: 2 = the library's default (index-ordered batches in the round-3 ocean "
                         "kernel), 1 = batches sorted by last call's trip counts over the whole chunk, 3 = within quarter-chunk windows")
    ap.add_argument("--solver-path", choices=("auto", "exact", "certified"), default="auto",
                    help="CF_OPT_SOLVER_PATH.  `value` is ALWAYS the path named here: exact (= auto: the library's default, the "
                         "reference's own iteration) or certified (opt-in: the reduced-iteration solve with per-cell exact-path "
                         "fallback, include/coflux.h).  auto at N = 1 also times the certified path beside it, interleaved, and "
                         "reports it as `value_certified` with its parity against the CPU oracle — never as `value`.  At N > 1 "
                         "only the named path runs, so a SCALE series is one algorithm at every N")
    ap.add_argument("--days", type=float, default=None,
                    help="LONG RUN instead of the throughput line: this many simulated days of the flux path at --dt seconds per step "
                         "(BASELINE configs[3]/[4]: --grid tripolar --nx 2160 --ny 1080 --days 30 --dt 300 is sixth_degree_tripolar_ocean_sea_ice.jl:52's "
                         "8 640 steps; --nx 360 --ny 180 --dt 1200 is one_degree_tripolar_ocean_sea_ice.jl:47).  cf_time_steps runs one "
                         "3-hourly snapshot interval per call through a RepeatYearJRA55-style record of --record-snapshots snapshots held in host "
                         "memory, a --window-slots sliding window in HBM (cf_window_*) and --ocean-states surface states; every --check-every'th "
                         "step is compared with the CPU oracle at 1e-9, and the fields at those steps are hashed and compared with an un-pipelined "
                         "host-driven loop over the same steps.  Prints one JSON line: flux-path seconds per simulated day and the flux-only SYPD ceiling")
    ap.add_argument("--dt", type=float, default=300.0, help="--days: coupled time step in seconds (3 h must be a whole number of steps)")
    ap.add_argument("--record-snapshots", type=int, default=16, help="--days: 3-hourly snapshots in the repeat-year record (16 = two days: a 30-day run wraps it 15 times)")
    ap.add_argument("--window-slots", type=int, default=4, help="--days: time_indices_in_memory of the sliding window (>= 4: three resident + one being refilled)")
    ap.add_argument("--ocean-states", type=int, default=4, help="--days: distinct ocean surface states the steps cycle through")
    ap.add_argument("--check-every", type=int, default=720, help="--days: compare with the CPU oracle every this many steps (a whole number of snapshot intervals)")
    ap.add_argument("--no-host-loop", action="store_true", help="--days: skip the un-pipelined host-driven loop (its hashes are then not compared)")
    ap.add_argument("--selftest", action="store_true",
                    help="no timing: verify the halo backends, run 10 steps, gather the surface on rank 0 and compare it with the "
                         "CPU oracle; prints one JSON line, exits non-zero naming the failing stage")
    ap.add_argument("--certified-budget", type=int, default=800, help="CF_OPT_CERTIFIED_BUDGET in units of 1e-9")
    ap.add_argument("--no-sorted-pass", action="store_true",
                    help="skip the informational pass with CF_OPT_TRIP_HINTS = 1 (profiling runs: only the default configuration's launches)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample: passes that fit this wall time")
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks (one per GPU) ourselves."""
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < a.gpus and not (a.share_device and ndev >= 1):
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} HIP device(s) visible on this node; "
                         f"refusing to run (an N-GPU number needs N devices)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(case_np, params, nx, ny, h, seconds):
    """The CPU oracle (C restatement, OpenMP over rows, one row per grab) timed on this box's host cores on the
    same workload: full passes of interpolate + solver + net fluxes over the surface, outputs preallocated
    outside the timer.  As many passes as fit `seconds` of wall time (at least 3), best pass reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as orc
    g = orc.make_grid(nx, ny, h, h, 1)
    cores = orc.max_threads()
    shape = (ny + 2 * h, nx + 2 * h)
    atmos = {n: np.zeros(shape) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
    fl = {n: np.zeros(shape) for n in ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature")}
    net = {n: np.zeros(shape) for n in ("u", "v", "T", "S", "shortwave_surface_flux")}

    def one_pass(nt):
        t0 = time.perf_counter()
        orc.interpolate_atmosphere_state(g, case_np["src"], case_np["weights"], 0, 1, 0.37, out=atmos)
        orc.compute_atmosphere_ocean_fluxes(g, params, case_np["ocean"], atmos, nthreads=nt, scales=False, out=fl)
        orc.compute_net_ocean_fluxes(g, params, case_np["ocean"], atmos, fl, weights=case_np["weights"], out=net)
        return time.perf_counter() - t0

    # How many threads: all the box shows is not always what the job may use (a container with a CPU quota runs 256
    # threads slower than 32: measured 304 ms vs 95 ms per pass on a 2 x 64-core host, scratch/cpu_threads.py) — one
    # calibration pass per candidate count, the fastest is the baseline's `cores`
    candidates = sorted({c for c in (cores, cores // 2, cores // 4, cores // 8, 64, 32) if 1 <= c <= cores}, reverse=True)
    one_pass(candidates[-1])   # (untimed: first touch of the output arrays)
    calib = {c: min(one_pass(c), one_pass(c)) for c in candidates}
    threads = min(calib, key=calib.get)
    t_first = one_pass(threads)
    repeats = min(100, max(3, int(seconds / max(t_first, 1e-3))))
    t_best = min([t_first] + [one_pass(threads) for _ in range(repeats - 1)])
    rec = dict(value=nx * ny / t_best, unit="cells/s", cores=threads, kind="port",
               sample=f"{repeats} full update_state passes over the {nx}x{ny} surface (best of {repeats}, "
                      f"{t_best * 1e3:.1f} ms); oracle/coflux_oracle.c, OpenMP over rows (dynamic, 1 row), {threads} threads "
                      f"(the fastest of {candidates} on this box: {', '.join(f'{c}: {calib[c] * 1e3:.0f} ms' for c in candidates)}; "
                      f"{cores} hardware threads visible)")
    return rec, dict(atmos=atmos, fluxes=fl, net=net)


# field scales of the parity metric |Δ| ≤ tol · max(|ref|, scale) (tests/util.py::FIELD_SCALE)
PARITY_SCALE = dict(sensible_heat=1.0, latent_heat=1.0, water_vapor=1e-6, x_momentum=1e-3, y_momentum=1e-3, temperature=1.0,
                    u=1e-6, v=1e-6, T=1e-6, S=1e-7, shortwave_surface_flux=1e-6)


def oracle_reference(case_np, params, nx, ny, h):
    """One untimed pass of the CPU oracle over this rank's surface (the checker of the solver-path decision; the timed
    cpu_baseline leg runs the same three calls)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    g = orc.make_grid(nx, ny, h, h, 1)
    atmos = orc.interpolate_atmosphere_state(g, case_np["src"], case_np["weights"], 0, 1, 0.37)
    fl = orc.compute_atmosphere_ocean_fluxes(g, params, case_np["ocean"], atmos, nthreads=min(32, orc.max_threads()), scales=False)
    net = orc.compute_net_ocean_fluxes(g, params, case_np["ocean"], atmos, fl, weights=case_np["weights"])
    return dict(atmos=atmos, fluxes={k: fl[k] for k in ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature")},
                net={k: net[k] for k in ("u", "v", "T", "S", "shortwave_surface_flux")})


def measured_parity(ctx, ref, dev_case, nx, ny, h, shares=None):
    """One cf_update_state on the inputs the CPU baseline ran on (time fraction 0.37, snapshot levels 0/1), compared
    with the oracle's outputs of that leg: worst |Δ| / max(|ref|, field scale) per field over interior + ring (fluxes)
    / interior (net fluxes).  The oracle is the checker here, never the thing measured."""
    import numpy as np
    from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES
    import torch
    from coflux import abi
    atmos, fl, net = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
    if shares is not None:  # (the diagnostic output: which cells the certified path sent down the exact one)
        fl["iterations"] = torch.zeros_like(fl["temperature"], dtype=torch.int32)
    ctx.update_state(dev_case["src"], dev_case["weights"], dev_case["ocean"], atmos, fl, net, time_fraction=0.37)
    ctx.sync()
    if shares is not None:
        it = fl.pop("iterations").cpu().numpy()
        solved = it > 0
        exact = (it & abi.CERTIFIED_EXACT_FLAG) != 0
        shares["cells_solved"] = int(solved.sum())
        shares["exact_path_share"] = float(exact.sum() / max(1, solved.sum()))
        shares["mean_map_evaluations_certified_cells"] = float(it[solved & ~exact].mean()) if (solved & ~exact).any() else None
        shares["mean_trips_exact_path_cells"] = float((it[exact] & 0xff).mean()) if exact.any() else None
    out = {}
    W1 = (slice(h - 1, h + ny + 1), slice(h - 1, h + nx + 1))
    W0 = (slice(h, h + ny), slice(h, h + nx))
    for name, got, want, W in (("fluxes", fl, ref["fluxes"], W1), ("net", net, ref["net"], W0)):
        for k in want:
            g = got[k].cpu().numpy()[W]
            r = want[k][W]
            out[f"{name}.{k}"] = float(np.max(np.abs(g - r) / np.maximum(np.abs(r), PARITY_SCALE[k])))
    return out


def long_run(a, ctx, params, nx, ny, h, tripolar, ocean_np, w_np, states, w, atmos_sets, fl, net, pipeline, mode):
    """--days: `days` simulated days of time_step!'s flux path on one GPU (see the option's help).  The loop a host-language driver
    would write around the C ABI: per 3-hourly snapshot interval, make the window hold snapshots k, k+1 (this interval) and k+2 (the
    request the pipelined loop's last step makes for the next interval's first step), then ONE cf_time_steps call for the interval's
    steps.  Nothing synchronises with the host except the pinned staging buffers' reuse and the check steps."""
    import hashlib
    import numpy as np
    import torch
    from coflux import abi, synthetic as syn
    from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, SnapshotWindow
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    spi = SNAPSHOT_INTERVAL / a.dt
    if abs(spi - round(spi)) > 1e-9 or spi < 1:
        raise SystemExit(f"bench.py --days: the 3-hourly snapshot interval must be a whole number of steps (--dt {a.dt})")
    spi = int(round(spi))
    n_steps = int(round(a.days * 86400.0 / a.dt))
    n_intervals = (n_steps + spi - 1) // spi
    n_steps = n_intervals * spi                     # whole intervals
    if a.check_every % spi or a.check_every <= 0:
        raise SystemExit(f"bench.py --days: --check-every must be a whole number of snapshot intervals ({spi} steps)")
    R, slots, n_states = a.record_snapshots, a.window_slots, len(states)
    if slots < 4 or R < 2:
        raise SystemExit("bench.py --days: --window-slots >= 4 and --record-snapshots >= 2")
    inc = a.dt / SNAPSHOT_INTERVAL
    record = syn.jra55_snapshots(R, temporal_correlation=0.95)   # the repeat-year record, in host memory: snapshot counter k is record[k mod R]
    g = orc.make_grid(nx, ny, h, h, 1)
    W1 = (slice(h - 1, h + ny + 1), slice(h - 1, h + nx + 1))
    W0 = (slice(h, h + ny), slice(h, h + nx))
    fold_args = ([abi.FOLD_CENTER, abi.FOLD_CENTER, abi.FOLD_X_FACE, abi.FOLD_Y_FACE], [1.0, 1.0, -1.0, -1.0])

    def clock(step):
        """(snapshot counter, time fraction) of a step — the arithmetic of cf_time_steps::source_at, in the same order"""
        total = 0.0 + float(step) * inc
        whole = int(np.floor(total))
        return whole, total - whole

    def new_window():
        win = SnapshotWindow(ctx, syn.JRA55_NX, syn.JRA55_NY, slots)
        return win

    def ensure(win, k):
        if win.find(k) >= 0:
            return
        slot = k % slots
        win.wait_slot(slot)
        for v in abi.JRA55_VARIABLES:
            np.copyto(win.host_view(slot, v), record[v][k % R])
        win.commit(slot, k)

    def digest(fields_f, fields_n):
        hh = hashlib.sha256()
        for k in FLUX_NAMES:
            hh.update(fields_f[k].cpu().numpy().tobytes())
        for k in fields_n:
            hh.update(fields_n[k].cpu().numpy().tobytes())
        return hh.hexdigest()[:16]

    def check_against_oracle(step, fields_f, fields_n):
        k, frac = clock(step)
        atm = orc.interpolate_atmosphere_state(g, record, w_np, k % R, (k + 1) % R, frac)
        st = ocean_np[step % n_states]
        ref_f = orc.compute_atmosphere_ocean_fluxes(g, params, st, atm, nthreads=min(32, orc.max_threads()), scales=False)
        ref_n = orc.compute_net_ocean_fluxes(g, params, st, atm, ref_f, weights=w_np)
        worst = 0.0
        for name, got, want, Wn in [(k_, fields_f[k_], ref_f[k_], W1) for k_ in FLUX_NAMES] + [(k_, fields_n[k_], ref_n[k_], W0) for k_ in fields_n]:
            gg, rr = got.cpu().numpy()[Wn], want[Wn]
            worst = max(worst, float(np.max(np.abs(gg - rr) / np.maximum(np.abs(rr), PARITY_SCALE[name]))))
        return worst

    # ---- pass 1: the C step loop, pipelined (the next step's interpolation in the solver launch's tail workgroups) ------------------
    win = new_window()
    sched = ctx.make_schedule(states, atmos_sets, first_level=0, time_fraction=0.0, time_fraction_increment=inc,
                              pipeline=abi.PIPELINE_CONTINUING if pipeline else 0, fold_north=tripolar)
    checks, wall, t_group = [], 0.0, None
    per_interval = []
    for k in range(n_intervals):
        if t_group is None:
            ctx.sync()
            t_group = time.perf_counter()
        for kk in (k, k + 1, k + 2):
            ensure(win, kk)
        src_struct = win.source(k, k + 1, 0.0)        # (orders the stream behind the uploads of the slots this interval reads …
        win.source(k + 1, k + 2, 0.0)                 #  … and of the one its last step's request reads)
        ctx.time_steps(k * spi, spi, sched, src_struct, w, fl, net)
        last = (k + 1) * spi - 1
        if (last + 1) % a.check_every == 0 or k == n_intervals - 1:
            ctx.sync()
            dt_group = time.perf_counter() - t_group
            wall += dt_group
            t_group = None
            checks.append(dict(step=last, worst_scaled_error_vs_oracle=check_against_oracle(last, fl, net), sha256=digest(fl, net)))
    win.close()
    ctx.discard_prefetched_atmosphere_state()

    # ---- pass 2: the same steps as an un-pipelined host-driven loop (three launches per step, one exchange set) ---------------------
    host = None
    if not a.no_host_loop:
        ctx.set_option(abi.OPT_MERGED_PREFETCH, 0)
        win = new_window()
        fl2, net2 = ctx.field_set(FLUX_NAMES), ctx.field_set(tuple(net))
        one = ctx.field_set(EXCHANGE_NAMES)
        host, t0 = [], time.perf_counter()
        want_steps = {c["step"] for c in checks}
        for step in range(n_steps):
            k, frac = clock(step)
            ensure(win, k)
            ensure(win, k + 1)
            st = states[step % n_states]
            if tripolar:
                ctx.fold_north_halo([st[f] for f in ("T", "S", "u", "v")], *fold_args, rows=2)
            ctx.update_state(win.source(k, k + 1, frac), w, st, one, fl2, net2)
            if step in want_steps:
                ctx.sync()
                host.append(dict(step=step, sha256=digest(fl2, net2)))
        ctx.sync()
        host_wall = time.perf_counter() - t0
        win.close()

    sim_days = n_steps * a.dt / 86400.0
    sec_per_day = wall / sim_days
    worst = max(c["worst_scaled_error_vs_oracle"] for c in checks)
    same = None if host is None else all(hc["sha256"] == c["sha256"] for hc, c in zip(host, checks)) and len(host) == len(checks)
    grid_name = f"TripolarGrid surface {nx}x{ny} (synthetic mesh with the real fold)" if tripolar else f"LatitudeLongitudeGrid surface {nx}x{ny}"
    return dict(metric="flux-path wall seconds per simulated day (time_step!'s update_state!: JRA55 interp + similarity-theory fluxes + net fluxes)",
                value=sec_per_day, unit="s per simulated day", higher_is_better=False, n_gpus=1, steps=n_steps, warmup=0,
                ms_per_step=wall / n_steps * 1e3, scaling="strong", vs_baseline=None, dtype="f64", data="synthetic",
                config=dict(workload=f"{grid_name}, {sim_days:g} simulated days at dt = {a.dt:g} s ({n_steps} steps of cf_time_steps, one call per "
                                     f"3-hourly snapshot interval = {spi} steps), RepeatYearJRA55-style record of {R} snapshots in host memory "
                                     f"(wrapped {n_intervals / R:.1f} times) through a {slots}-slot window in HBM (cf_window_*; every slot rewritten "
                                     f"{n_intervals / slots:.0f} times), {n_states} ocean surface states in turn, "
                                     f"SimilarityTheoryFluxes(:{a.flux_configuration}) + Radiation, halo {h}, ring 1",
                            pipeline_mode=mode, solver_path="exact", cells=nx * ny),
                simulated_days=sim_days, wall_seconds=wall, cells_per_s=nx * ny * n_steps / wall,
                flux_only_sypd_ceiling=86400.0 / sec_per_day / 365.0,
                sypd_note="simulated years per wall day if NOTHING but this path ran: the ceiling the flux path alone puts on the coupled model's "
                          "SYPD (the ocean dynamical core, which sets the real figure, is out of scope: DESIGN.md section 1)",
                checks=checks, checks_every_steps=a.check_every, worst_scaled_error_vs_oracle=worst, tolerance=1e-9,
                parity_ok=bool(worst <= 1e-9),
                host_loop=(None if host is None else dict(kind="un-pipelined: cf_update_state per step from the host, one exchange set, CF_OPT_MERGED_PREFETCH = 0",
                                                          wall_seconds=host_wall, ms_per_step=host_wall / n_steps * 1e3, hashes_equal_at_every_check=same,
                                                          checks=host)),
                parity="vs reference: unpinned (the oracle is a restatement; see DESIGN.md)")


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a)
    import numpy as np
    import torch
    import torch.distributed as dist

    from coflux import abi, synthetic as syn
    from coflux import interface_computations as ic
    from coflux.distributed import SlabHaloExchanger, slab_bounds
    from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the line would misreport n_gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the flux path has no CPU backend")
    if a.share_device:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if a.share_device else "cuda"   # where the tensors of the host-side collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.share_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    default_protocol = a.steps is None
    steps = a.steps if a.steps is not None else 1000
    warmup = a.warmup if a.warmup is not None else 50
    # --steps K: at least 3 repetitions of exactly K steps, more while they stay inside ≈ 30 ms per solver path (K = 20: 9
    # regions of 1.7 ms) — a short region is exposed to whatever the host does for a few hundred microseconds (one evidence run
    # read 0.085, 0.095, 0.096 ms per step in three consecutive 20-step regions whose kernels ran at their usual 76 µs)
    reps = a.repetitions if a.repetitions is not None else (5 if default_protocol else (max(3, min(9, int(0.030 / (steps * 0.085e-3)))) | 1))   # (odd: the median is one of the samples)

    h = a.halo
    if a.scaling == "weak":
        ny_global, (j0, j1) = a.ny * world, (rank * a.ny, (rank + 1) * a.ny)
    else:
        ny_global = a.ny
        j0, j1 = slab_bounds(a.ny, rank, world)
    nx, ny = a.nx, j1 - j0
    if ny < 3:
        raise SystemExit(f"bench.py: slab of {ny} rows on rank {rank} is too thin")

    fluxes_cfg = {"default": ic.SimilarityTheoryFluxes, "corrected": ic.corrected_atmosphere_ocean_fluxes,
                  "ncar": ic.ncar_atmosphere_ocean_fluxes}[a.flux_configuration]()
    params = ic.flux_params(fluxes_cfg, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))

    # ---- synthetic inputs, resident in HBM before the timed region ---------------------------------
    n_levels = 4
    # consecutive 3-hourly snapshots correlated 0.95: the atmosphere changes by ≈ 3 % of its variability per 20-min step
    src_np = syn.jra55_snapshots(n_levels, temporal_correlation=0.95)
    tripolar = a.grid == "tripolar"
    if tripolar and a.scaling != "strong":
        raise SystemExit("bench.py: --grid tripolar shards ONE folded surface (strong scaling)")

    def make_case(r0, r1, n_states=2):
        """`n_states` ocean states (one step apart each) and the interpolation weights of rows [r0, r1) of the global surface:
        every field is a function of the GLOBAL cell index, so slabs and the whole surface agree where they overlap."""
        rows = r1 - r0
        if tripolar:
            tc = syn.tripolar_case(nx, ny_global, h, h, j0=r0, j1=r1)
            out = [dict(tc["ocean"])]
            base = syn.ocean_state(nx, ny_global, h, h, latitude=(-80.0, 90.0))
            for n in range(1, n_states):
                evolved = syn.evolved_ocean_state(base, nx, ny_global, h, h, n)
                nxt = {}
                for k in ("T", "S", "u", "v"):          # the later states fold like the first
                    gfull = evolved[k].copy()
                    syn.fold_north(gfull, nx, ny_global, h, h, 2, syn.FOLD_LOCATION[k], syn.FOLD_SIGN[k])
                    nxt[k] = np.ascontiguousarray(gfull[r0:r1 + 2 * h])
                nxt["mask"] = tc["ocean"]["mask"]
                out.append(nxt)
            return out, tc["weights"]
        first = syn.ocean_state(nx, rows, h, h, ny_global=ny_global, j_offset=r0)
        out = [first] + [syn.evolved_ocean_state(first, nx, rows, h, h, n, ny_global=ny_global, j_offset=r0) for n in range(1, n_states)]
        fi, fj, phi = syn.latlon_fractional_indices(nx, rows, h, h, ny_global=ny_global, j_offset=r0)
        return out, dict(separable=True, fi=fi, fj=fj, latitude=phi)

    if a.days is not None and (world != 1 or a.config != "ocean"):
        raise SystemExit("bench.py --days: the long run is a one-GPU, ocean-only (configs without prognostic sea ice) measurement")
    ocean_np, w_np = make_case(j0, j1, max(2, a.ocean_states) if a.days is not None else 2)

    ctx = FluxContext(nx, ny, h, h, params, ring=1, device=local_rank)
    if a.trip_hints != 2:
        ctx.set_option(abi.OPT_TRIP_HINTS, a.trip_hints)
    if a.ao_chunk:
        ctx.set_option(abi.OPT_AO_CHUNK, a.ao_chunk)
    if a.latency_layout != "auto":
        ctx.set_option(abi.OPT_LATENCY_LAYOUT, {"never": 0, "always": 2}[a.latency_layout])
    ctx.set_option(abi.OPT_CERTIFIED_BUDGET, a.certified_budget)
    if a.solver_path == "certified":
        ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED)
    ring_rows = ctx.grid.ring + 1
    states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in ocean_np]
    for st in states[1:]:
        st["mask"] = states[0]["mask"]   # one static wet mask (the solver's chunk table is keyed on it)
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = {k: (ctx.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in w_np.items()}
    # --pipeline auto: the next step's interpolation as tail workgroups of the solver launch (CF_OPT_MERGED_PREFETCH = 2) wherever
    # the round-3 ocean kernel runs with its fused epilogue — measured −3 % per step on the 1/4-degree surface, −9 … −16 % on
    # slabs and on the 1/6-degree surface (profiles/r04_experiments.md §6); the other solver kernels keep the three launches
    lean0, fused0 = ctx.solver_path()
    mode = a.pipeline
    if mode == "auto":
        mode = "tail" if (fused0 == 1 and (lean0 or a.flux_configuration == "ncar")) else "off"
    pipeline = mode in ("on", "merged", "tail")
    if mode in ("merged", "tail"):
        ctx.set_option(abi.OPT_MERGED_PREFETCH, 1 if mode == "merged" else 2)
    tail_mode = mode == "tail" and fused0 == 1 and (lean0 or a.flux_configuration == "ncar")
    atmos_sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2 if pipeline else 1)]
    fl = ctx.field_set(FLUX_NAMES)
    # compute_net_ocean_fluxes! writes five fields (τx, τy, Jᵀ, Jˢ, penetrating shortwave: the 40 B/cell of the contract);
    # the ABI's three radiation diagnostics are optional outputs and off unless asked for
    net = ctx.field_set(NET_NAMES if a.net_diagnostics else NET_NAMES[:5])

    ice = ice_state = ai = net_ice = None
    if a.config == "sea_ice":
        ice_cfg = ic.corrected_atmosphere_sea_ice_fluxes() if a.flux_configuration != "ncar" else ic.ncar_atmosphere_sea_ice_fluxes()
        ctx.set_sea_ice_formulation(ic.flux_params(ice_cfg))
        ctx.set_option(abi.OPT_ICE_ORBIT_SHORTCUT, a.ice_orbit_shortcut)
        ctx.set_option(abi.OPT_ICE_FREE_CELLS, abi.ICE_FREE_ZERO if a.ice_free_cells == "zero" else abi.ICE_FREE_ITERATE)
        si_np = syn.sea_ice_state(nx, ny, h, h, ny_global=ny_global, j_offset=j0)
        if a.ice_free_cells == "zero":   # the opt-in's definition of open water is ℵ = 0 AND hᵢ = 0: no thickness where there is no ice
            si_np["thickness"] = np.where(ocean_np[0]["ice_concentration"] > 0, si_np["thickness"], 0.0)
        ice = {k: ctx.to_device(ocean_np[0]["ice_" + k]) for k in ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")}
        ice_state = dict(concentration=ice["concentration"], **{k: ctx.to_device(si_np[k]) for k in ("thickness", "top_temperature", "u", "v", "albedo")})
        ai = ctx.field_set(FLUX_NAMES)
        net_ice = ctx.field_set(("top_heat", "bottom_heat"))
        # the skin temperature the interface solve finds IS the sea-ice model's top_surface_temperature — the next step's
        # first guess (atmosphere.jl:34-39; coflux/models.py::update_state): one buffer for both, as in a coupled run
        ai["temperature"].copy_(ice_state["top_temperature"])
        ice_state["top_temperature"] = ai["temperature"]

    if a.days is not None:
        out = long_run(a, ctx, params, nx, ny, h, tripolar, ocean_np, w_np, states, w, atmos_sets, fl, net, pipeline, mode)
        print(json.dumps(out), flush=True)
        ctx.close()
        return

    # ---- halo rows: prove each backend on this machine before timing it ------------------------------
    # The synthetic state is a function of the GLOBAL cell index, so every rank knows what its neighbours' boundary
    # rows must be: wipe the halo rows, exchange, compare.
    halo_fields = [states[0][k] for k in ("T", "S", "u", "v")]

    def verify(exchanger):
        rows = []
        if rank > 0:
            rows += list(range(h - ring_rows, h))
        if rank < world - 1:
            rows += list(range(h + ny, h + ny + ring_rows))
        want = [[f[r].clone() for r in rows] for f in halo_fields]
        for f in halo_fields:
            for r in rows:
                f[r].fill_(float("nan"))
        good = True
        try:
            exchanger(halo_fields)
            ctx.sync()
            torch.cuda.synchronize()
            good = all(torch.equal(f[r], x) for f, xs in zip(halo_fields, want) for r, x in zip(rows, xs))
        except Exception as exc:  # a backend that fails is a backend that is not used
            print(f"[bench] rank {rank}: halo backend {exchanger.backend} failed: {exc}", file=sys.stderr)
            good = False
        for f, xs in zip(halo_fields, want):      # restore either way
            for r, x in zip(rows, xs):
                f[r].copy_(x)
        flag = torch.tensor([1 if good else 0], device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    exchangers, halo_verified, rccl_comm_ranks = {}, {}, None
    if world > 1:
        wanted = ("rccl", "peer", "torch") if a.halo_backend == "auto" else (a.halo_backend,)
        if a.share_device:
            wanted = tuple(n for n in wanted if n == "peer")
        for name in wanted:
            ok = torch.tensor([1], device=coll_dev)
            try:
                ex = SlabHaloExchanger(ctx, ny, h, rows=ring_rows, backend=name)
            except Exception as exc:
                print(f"[bench] rank {rank}: halo backend {name} unavailable: {exc}", file=sys.stderr)
                ex, ok = None, torch.tensor([0], device=coll_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            halo_verified[name] = bool(ok.item()) and verify(ex)
            if halo_verified[name]:
                exchangers[name] = ex
            if ex is not None and ex.backend == "rccl" and rccl_comm_ranks is None:
                try:   # RCCL's own count of the ranks it connected, beside WORLD_SIZE
                    rccl_comm_ranks = ctx.comm_count()[0]
                except Exception as exc:
                    print(f"[bench] rank {rank}: cf_comm_count: {exc}", file=sys.stderr)
            if name == "torch" and len(exchangers) > 1:
                del exchangers["torch"]   # only the fallback when no native path verified
        if not exchangers and not a.selftest:   # (--selftest reports it as its first stage)
            raise SystemExit(f"bench.py: no halo backend reproduces the neighbours' boundary rows ({halo_verified}); refusing to time")

    inc = DT / SNAPSHOT_INTERVAL
    backend_code = {"rccl": abi.HALO_RCCL, "peer": abi.HALO_PEER}

    def schedule_for(name):
        return ctx.make_schedule(states, atmos_sets, first_level=0, time_fraction=0.0, time_fraction_increment=inc,
                                 pipeline=abi.PIPELINE_CONTINUING if pipeline else 0,   # one loop over many calls
                                 halo_backend=backend_code.get(name, abi.HALO_NONE),
                                 halo_rows=ring_rows if name in backend_code else 0,
                                 fold_north=tripolar and rank == world - 1)

    def run_steps(name, sched, first, n):
        if a.config == "sea_ice" or name == "torch":   # host-driven steps (five-launch sea-ice step; torch P2P halo)
            for s in range(first, first + n):
                st = states[s % 2]
                if name == "torch":
                    exchangers["torch"]([st[k] for k in ("T", "S", "u", "v")])
                if tripolar and rank == world - 1:
                    ctx.fold_north_halo([st[k] for k in ("T", "S", "u", "v")],
                                        [abi.FOLD_CENTER, abi.FOLD_CENTER, abi.FOLD_X_FACE, abi.FOLD_Y_FACE], [1.0, 1.0, -1.0, -1.0], rows=2)
                tot = s * inc
                l1 = int(tot) % n_levels
                kw = dict(level1=l1, level2=(l1 + 1) % n_levels, time_fraction=tot - int(tot))
                cur = atmos_sets[s % len(atmos_sets)]
                if tail_mode and len(atmos_sets) == 2:
                    # the host-driven loop asks for the next step's atmosphere itself (cf_time_steps does the same in C): it
                    # becomes the tail workgroups of this step's ocean-solver launch
                    nxt = (s + 1) * inc
                    l1n = int(nxt) % n_levels
                    ctx.prefetch_atmosphere_state(src, w, atmos_sets[(s + 1) % 2], level1=l1n, level2=(l1n + 1) % n_levels,
                                                  time_fraction=nxt - int(nxt))
                if a.config == "sea_ice":
                    ctx.update_state_sea_ice(src, w, st, cur, fl, net, ice, ice_state, ai, net_ice, **kw)
                else:
                    ctx.update_state(src, w, st, cur, fl, net, **kw)
        else:
            ctx.time_steps(first, n, sched, src, w, fl, net)

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    # ---- the halo rows inside the solver launch: proven on this machine against the exchange kernel before it is timed -------------
    def prove_halo_in_launch():
        """True where six steps with CF_OPT_HALO_IN_SOLVER_LAUNCH leave, on every rank, the bits of six steps with the exchange kernel
        (halo rows poisoned first, the riders really used); the option is left on exactly then."""
        if not (world > 1 and "peer" in exchangers and tail_mode and a.config == "ocean" and a.halo_in_launch != "off"):
            return False

        def six_steps(flag):
            """(fields, exchanges that rode, ok): every collective sits OUTSIDE the part that can fail, so that a rank whose
            riders time out (≈ 5 s, surfacing as an error of its next cf_sync) stays in step with the others"""
            dist.barrier()
            fields, rode, ok = None, 0, True
            try:
                ctx.set_option(abi.OPT_HALO_IN_SOLVER_LAUNCH, flag)
                for st in states:                        # rows a neighbour must deliver (the synthetic state is a function of the global index)
                    for k in ("T", "S", "u", "v"):
                        if rank > 0:
                            st[k][h - ring_rows:h] = float("nan")
                        if rank < world - 1:
                            st[k][h + ny:h + ny + ring_rows] = float("nan")
                torch.cuda.synchronize()
                before = ctx.peer_halo_stats()
                run_steps("peer", schedule_for("peer"), 0, 6)
                ctx.sync()
                torch.cuda.synchronize()
                ctx.discard_prefetched_atmosphere_state()
                rode = ctx.peer_halo_stats()[1] - before[1]
                fields = [fl[k].clone() for k in FLUX_NAMES] + [net[k].clone() for k in net]
            except Exception as exc:  # noqa: BLE001 — a form that fails on this machine is a form that is not used
                print(f"[bench] rank {rank}: six steps with CF_OPT_HALO_IN_SOLVER_LAUNCH = {flag} failed: {exc}", file=sys.stderr)
                ok = False
                try:
                    ctx.discard_prefetched_atmosphere_state()
                    ctx.sync()
                except Exception:  # noqa: BLE001
                    pass
            dist.barrier()
            return fields, rode, ok
        ref_fields, _, ok0 = six_steps(0)
        got_fields, rode, ok1 = six_steps(1)
        good = (ok0 and ok1 and rode > 0 and all(torch.equal(x, y) for x, y in zip(ref_fields, got_fields)) and
                all(bool(torch.isfinite(x[h:h + ny, h:h + nx]).all()) for x in got_fields))
        flag = torch.tensor([1 if good else 0], device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        proven = bool(flag.item())
        ctx.set_option(abi.OPT_HALO_IN_SOLVER_LAUNCH, 1 if proven else 0)
        return proven

    # ---- --selftest: first contact with an N-GPU node, stage by stage -----------------------------------------------
    if a.selftest:
        report = dict(selftest="running", n_gpus=world, rank_rows=ny, halo_verified=halo_verified if world > 1 else None,
                      rccl_comm_ranks=rccl_comm_ranks, stages=[])

        def stage(name, fn):
            """Runs fn on every rank; the ranks agree on the outcome before anyone moves on (or exits)."""
            err = None
            try:
                fn()
            except BaseException as exc:  # noqa: BLE001 — reported, then every rank leaves together
                err = f"rank {rank}: {type(exc).__name__}: {exc}"
            bad = torch.tensor([0 if err is None else 1], device=coll_dev)
            if world > 1:
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if err is not None:
                print(f"[bench --selftest] stage {name} FAILED on {err}", file=sys.stderr, flush=True)
            if bool(bad.item()):
                if rank == 0:
                    report.update(selftest="failed", failed_stage=name, error=err or "another rank failed: see its stderr")
                    print(json.dumps(report), flush=True)
                sys.exit(3)
            report["stages"].append(name)

        def check_halos():
            if world > 1 and not exchangers:
                raise RuntimeError(f"no halo backend verified: {halo_verified}")
            if world > 1 and rccl_comm_ranks is not None and rccl_comm_ranks != world:
                raise RuntimeError(f"RCCL connected {rccl_comm_ranks} ranks, WORLD_SIZE is {world}")
        stage("halo_backends", check_halos)
        name0 = next(iter(exchangers), "none")
        nself = 10
        stage("ten_steps[" + name0 + "]", lambda: (run_steps(name0, schedule_for(name0), 0, nself), barrier()))
        got = {}

        def gather():
            fields = [("fluxes." + k, fl[k]) for k in FLUX_NAMES] + [("net." + k, net[k]) for k in list(net)[:5]]
            for key, t in fields:
                mine = t[h:h + ny, h:h + nx].contiguous().cpu()
                if world > 1:
                    parts = [None] * world
                    dist.all_gather_object(parts, mine.numpy())
                    got[key] = np.concatenate(parts, axis=0)
                else:
                    got[key] = mine.numpy()
        stage("gather_surface", gather)

        def compare():
            if rank != 0:
                return
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as orc
            oceans_g, w_g = make_case(0, ny_global)
            last = nself - 1
            tot = last * (DT / SNAPSHOT_INTERVAL)
            l1 = int(tot) % n_levels
            g = orc.make_grid(nx, ny_global, h, h, 1)
            atm = orc.interpolate_atmosphere_state(g, src_np, w_g, l1, (l1 + 1) % n_levels, tot - int(tot))
            ref_f = orc.compute_atmosphere_ocean_fluxes(g, params, oceans_g[last % 2], atm, nthreads=min(32, orc.max_threads()), scales=False)
            ref_n = orc.compute_net_ocean_fluxes(g, params, oceans_g[last % 2], atm, ref_f, weights=w_g)
            worst = {}
            for key, arr in got.items():
                grp, k = key.split(".")
                r = (ref_f if grp == "fluxes" else ref_n)[k][h:h + ny_global, h:h + nx]
                worst[key] = float(np.max(np.abs(arr - r) / np.maximum(np.abs(r), PARITY_SCALE[k])))
            report["worst_scaled_error_vs_oracle"] = worst
            tol = 1e-9 if ctx.solver_iteration_path() == abi.SOLVER_PATH_EXACT else 2e-6
            if max(worst.values()) > tol:
                raise RuntimeError(f"gathered surface differs from the single-domain oracle: {worst}")
        stage("compare_with_oracle", compare)
        proven = prove_halo_in_launch()      # (reported, never fatal: where the riders do not reproduce the exchange kernel it is simply not used)
        report["halo_in_solver_launch_proven"] = proven if world > 1 else None
        if rank == 0:
            report["selftest"] = "ok"
            print(json.dumps(report), flush=True)
        ctx.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    halo_in_launch = prove_halo_in_launch()
    if a.halo_in_launch == "on" and world > 1 and not halo_in_launch:
        raise SystemExit("bench.py --halo-in-launch on: the riders do not reproduce the exchange kernel's steps on this machine")

    settle, per_rank = {}, {}

    PATH_CODE = {"exact": abi.SOLVER_PATH_EXACT, "certified": abi.SOLVER_PATH_CERTIFIED}
    SWITCH_STEPS = 4   # untimed steps after a change of CF_OPT_SOLVER_PATH inside the repetition loop

    def timed(name, paths):
        """Times `steps` steps `reps` times for every solver path in `paths` on halo backend `name`.  With two paths the
        repetitions INTERLEAVE (exact, certified, exact, certified, …): the paths differ by 3 %, a device drifts by as much
        within a second of running, and whichever is timed second would carry the drift (measured: the same path 0.0825 →
        0.0845 ms over three consecutive runs of this script on one box; profiles/r05_experiments.md)."""
        sched = schedule_for(name)
        # Clocks first: a cold device needs tens of milliseconds of work before it holds its sustained clock, and a
        # 20-step region is 3 ms.  Untimed steps until ≈ 0.15 s have passed, then the W warm-up steps the caller asked
        # for, then exactly K timed steps — so that --steps 20 and --steps 1000 measure the same machine state.
        # (every rank must run the SAME number of steps — the halo rows pair up step by step — so the ranks agree on
        # when to stop: all of them have been busy for 0.15 s)
        ctx.set_option(abi.OPT_SOLVER_PATH, PATH_CODE[paths[0]])
        done, t_start = 0, time.perf_counter()
        while True:
            run_steps(name, sched, done, 50)
            ctx.sync()
            done += 50
            enough = time.perf_counter() - t_start >= 0.15
            if world > 1:
                flag = torch.tensor([1 if enough else 0], device=coll_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                enough = bool(flag.item())
            if enough:
                break
        settle[name] = done
        run_steps(name, sched, done, warmup)
        first = done + warmup
        samples, by_rank = {q: [] for q in paths}, {q: [] for q in paths}
        switches = 0
        for _ in range(reps):
            for q in paths:
                if len(paths) > 1:
                    ctx.set_option(abi.OPT_SOLVER_PATH, PATH_CODE[q])
                    run_steps(name, sched, first, SWITCH_STEPS)
                    first += SWITCH_STEPS
                    switches += 1
                barrier()
                t0 = time.perf_counter()
                run_steps(name, sched, first, steps)
                barrier()
                dt = time.perf_counter() - t0
                if world > 1:
                    every = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
                    dist.all_gather(every, torch.tensor([dt], dtype=torch.float64, device=coll_dev))
                    by_rank[q].append([float(x.item()) for x in every])
                    dt = max(by_rank[q][-1])           # the MAX over ranks is the step time of the job
                samples[q].append(dt)
                first += steps
        switch_steps[name] = switches * SWITCH_STEPS
        out = {}
        for q in paths:
            med = statistics.median(samples[q])
            nearest = min(range(len(samples[q])), key=lambda n: abs(samples[q][n] - med))   # (even counts: the sample next to the median)
            out[q] = (med, samples[q], sched, first, by_rank[q][nearest] if by_rank[q] else None)
        return out

    # ---- the two solver paths (CF_OPT_SOLVER_PATH) ---------------------------------------------------------------------
    # `value` is the path the command line names — exact unless --solver-path certified — on every verified halo backend.
    # With auto at N = 1 the certified path is timed beside it (interleaved repetition by repetition on the same schedule),
    # checked against the CPU oracle, and reported as `value_certified`: a second number, never the number of record
    # (VERDICT r5 / ADVICE r5: the library and ComponentInterfaces default to the exact path, so that is what `value` measures;
    # at N > 1 one path only, so that the points of a scaling series are the same algorithm).
    first_path = "certified" if a.solver_path == "certified" else "exact"
    ctx.set_option(abi.OPT_SOLVER_PATH, PATH_CODE[first_path])
    if first_path == "certified" and ctx.solver_iteration_path() != abi.SOLVER_PATH_CERTIFIED:
        first_path = "exact"    # (the option does not apply to this formulation / geometry: the exact path ran)
    paths = [first_path]
    if a.solver_path == "auto" and a.config == "ocean" and world == 1:
        ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED)
        if ctx.solver_iteration_path() == abi.SOLVER_PATH_CERTIFIED:
            paths.append("certified")
        ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_EXACT)
    switch_steps = {}
    results = {}
    for name in (list(exchangers) or ["none"]):
        results[name] = timed(name, paths)
    best = min(results, key=lambda k: results[k][first_path][0])
    path_results = {q: results[best][q][:4] for q in paths}
    path_per_rank = {q: results[best][q][4] for q in paths}
    for name in results:
        per_rank[name] = results[name][first_path][4]
    results = {name: results[name][first_path][:4] for name in results}   # (halo_paths_ms_per_step reads these)
    path_parity, chosen = {}, first_path
    if "certified" in paths and first_path == "exact":
        ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED)
        ref_rank = oracle_reference(dict(ocean=ocean_np[0], src=src_np, weights=w_np), params, nx, ny, h)
        shares = {}
        worst = measured_parity(ctx, ref_rank, dict(src=src, weights=w, ocean=states[0]), nx, ny, h, shares=shares)
        # every field parity_measured reports — the six flux fields (the north star's statement) and the net fluxes
        # assembled from them (J_S ∝ F_v − P and the face stresses can cancel: measured, not bounded per cell)
        six = max(v for k, v in worst.items() if k.startswith("fluxes."))
        path_parity["certified"] = dict(max=max(worst.values()), max_six_flux_fields=six,
                                        within_half_the_tolerance=bool(max(worst.values()) <= 5e-7),
                                        worst_scaled_error=worst, exact_path_cells=shares,
                                        metric="|got - oracle exact path| / max(|oracle|, field scale), one cf_update_state")
    ctx.set_option(abi.OPT_SOLVER_PATH, PATH_CODE[chosen])
    elapsed, samples, sched, next_step = path_results[chosen]
    next_step = max(v[3] for v in path_results.values())

    cells_total = nx * (ny_global if a.scaling == "weak" else a.ny) if world > 1 else nx * ny
    cells_rank = (nx + 2) * (ny + 2)  # a launch covers the ring as the reference does
    value = cells_total * steps / elapsed

    # ---- per-kernel times: a SEPARATE, event-bracketed pass over the same schedule (the event records between
    # dependent kernels cost stream time, so they stay out of the region `value` is measured on) ------------------
    def instrumented(n, hints):
        ctx.set_option(abi.OPT_TRIP_HINTS, hints)
        # settle (and rebuild hints) after the switch — and bring the clocks back up: the solver-path decision above ran the
        # CPU oracle for a second with the device idle, and a cold device measured this pass's launches 10 % slow (85.9 vs
        # 78.1 µs, profiles/r05_experiments.md §4).  The step count of the timed region's own settle loop: the same on every rank
        idle = 4 + (settle.get(best) or 0)
        run_steps(best, sched, next_step, idle)
        ctx.profile_enable(n)
        run_steps(best, sched, next_step + idle, n)
        out = [ctx.profile_read(k) for k in range(3)]
        ctx.profile_enable(0)
        return out

    n_prof = min(max(steps, 20), 200)
    # CF_OPT_TRIP_HINTS: 2 = the library's default (what the timed region ran), 1 = every solver orders its batches by
    # the previous call's trip counts (round 2's default; reported beside it)
    prof = instrumented(n_prof, a.trip_hints) if a.config == "ocean" and best != "torch" else None
    prof_sorted = instrumented(n_prof, 1) if prof and not a.no_sorted_pass else None
    ctx.set_option(abi.OPT_TRIP_HINTS, a.trip_hints)

    if rank == 0:
        kw = dict(src=src, weights=w, ocean=states[0], atmos=atmos_sets[0], fluxes=fl, net=net, time_fraction=0.37)
        interp_ms = min(ctx.time_stage(abi.STAGE_INTERPOLATE, 50, **kw) for _ in range(3))
        if mode == "on":   # the kernel the pipelined step actually runs on the auxiliary stream
            ctx.set_option(abi.OPT_INTERP_TILE_CAP, 0)
            interp_bg_ms = min(ctx.time_stage(abi.STAGE_INTERPOLATE, 50, **kw) for _ in range(3))
            ctx.set_option(abi.OPT_INTERP_TILE_CAP, 128)
        net_ms_alone = min(ctx.time_stage(abi.STAGE_NET_FLUXES, 50, **kw) for _ in range(3))
        ao_ms_alone = min(ctx.time_stage(abi.STAGE_AO_FLUXES, 50, **kw) for _ in range(3))
        if prof:
            ao_ms, nrec = prof[1]
            net_ms = prof[2][0]
        else:
            ao_ms, nrec, net_ms = ao_ms_alone, 50, net_ms_alone
        copy_bytes = 256 << 20
        copy_ms = ctx.time_copy(copy_bytes, 20)

        # HBM bytes and VALU instructions per launch from committed rocprofv3 PMC passes (separate runs of this command,
        # tools/round_profile.sh; see profiles/): properties of the kernels on this workload, not re-measured in this
        # run — `traffic_source` / `instructions_source` name the file and the commit it was taken at
        lean, fused = ctx.solver_path()

        def variant_key(name):
            """profile kernel name → the key the roofline looks up: the lean solver's template variants are told apart
            (round 6 on: <COARE, FUSE, TAIL, CERT>; the round-5 counter files: <COARE, BLOCK, FUSE, FUSE_INTERP[, TAIL[, CERT]]>)"""
            base = name.split("<")[0].split("::")[-1].strip()
            if base == "ao_lean_kernel" and "<" in name:
                t = [x.strip() for x in name.split("<")[1].split(">")[0].split(",")]
                if t[1] in ("true", "false"):
                    fuse_net, piped, cert = t[1] == "true", t[2] == "true", t[3] == "true"
                else:
                    fuse_net, piped, cert = t[2] == "true", len(t) > 4 and t[4] == "true", len(t) > 5 and t[5] == "true"
                return base + (":fused" if fuse_net else ":plain") + (":piped" if piped else "") + (":certified" if cert else "")
            return base
        traffic, traffic_source, sq, sq_source = {}, None, {}, None
        canonical = (nx, ny, a.flux_configuration, world, a.config) == (1440, 560, "default", 1, "ocean")
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")))
                if canonical:
                    traffic = {variant_key(k): v["hbm_bytes_per_launch"] for k, v in pmc["kernels"].items()}
                    traffic_source = (f"committed: profiles/{tag}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes"
                                      f"{', tree ' + pmc['commit'] if pmc.get('commit') else ''})")
                break
            except Exception:
                continue
        for tag in ("r06", "r05", "r04", "r03"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.json")))
                if canonical:
                    sq = {variant_key(k): v for k, v in pmc["kernels"].items()}
                    sq_source = f"committed: profiles/{tag}_pmc_sq.json (rocprofv3 --pmc SQ_INSTS_VALU …, tree {pmc.get('commit')})"
                break
            except Exception:
                continue
        ao_kernel = "ao_lean_kernel" if lean else "ao_flux_fast_kernel"
        # the riders are priced only where the timed launch carried them: the event-bracketed pass over the schedule.  Without it
        # (--config sea_ice, the torch halo path) `ao_ms` is the solver stage alone, back to back, and so is its byte count —
        # with sea ice the ocean solve runs inside the interface solve's launch (ice_ocean_kernel), which has no roofline line here
        tail_priced = tail_mode and prof is not None
        ao_key = (ao_kernel + (":fused" if fused else ":plain") + (":piped" if tail_priced else "") +
                  (":certified" if chosen == "certified" else "")) if lean else ao_kernel
        ao_what = "compute_atmosphere_ocean_fluxes!" + (" + the cell-local part of compute_net_ocean_fluxes! in its epilogue" if fused else "")
        ao_bytes = BYTES_AO_FUSED if fused else BYTES_AO
        if tail_priced:   # the launch also interpolates the NEXT step's atmosphere state in its tail workgroups
            ao_what += " + interpolate_atmosphere_state! of the next step in its tail workgroups"
            ao_bytes += BYTES_INTERP

        def roof(name, nbytes, ncells, ms, **extra):
            achieved = nbytes * ncells / (ms * 1e-3) / 1e9
            return dict(bound="hbm", kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic.get(extra.pop("traffic_key", name.split(" ")[0])), traffic_source=traffic_source,
                        bytes_per_cell=nbytes, cells_per_launch=ncells, avg_launch_ms=ms,
                        cells_per_s=ncells / (ms * 1e-3), **extra)

        grid_name = (f"TripolarGrid surface {nx}x{a.ny} (synthetic mesh with the real fold; general weights + wind rotation)" if tripolar
                     else f"1/4-degree LatitudeLongitudeGrid surface {nx}x{a.ny}")
        workload = (f"{grid_name} per {'GPU' if a.scaling == 'weak' else 'job'}"
                    f"{' sharded into ' + str(world) + ' latitude slabs' if world > 1 and a.scaling == 'strong' else ''}, "
                    f"JRA55 640x320 f32 atmosphere ({n_levels}-snapshot window, clock advancing 20 min per step), "
                    f"SimilarityTheoryFluxes(:{a.flux_configuration}) + Radiation"
                    f"{' + sea-ice interface and partition (config 3)' if a.config == 'sea_ice' else ''}, halo {h}, ring 1")
        out = dict(metric="flux-kernel surface cells/s (update_state!: JRA55 interp + similarity-theory fluxes + net fluxes)",
                   value=value, unit="cells/s", n_gpus=world, steps=steps, warmup=warmup,
                   ms_per_step=elapsed / steps * 1e3, higher_is_better=True, scaling=a.scaling if world > 1 else "strong",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=workload, global_cells=cells_total, parallelism=f"latitude-slab x{world}",
                               rows_per_rank=ny, halo_backend=best, halo_verified=(halo_verified if world > 1 else None),
                               rccl_comm_ranks=rccl_comm_ranks, solver_path=chosen,
                               ice_free_cells=(a.ice_free_cells if a.config == "sea_ice" else None),
                               certified_budget=(a.certified_budget * 1e-9 if "certified" in path_results else None),
                               halo_rows=ring_rows if world > 1 else 0, halo_in_solver_launch=(halo_in_launch if world > 1 else None),
                               pipelined_interpolation=pipeline,
                               pipeline_mode=mode,
                               step_loop="cf_time_steps (C)" if (a.config == "ocean" and best != "torch") else "host"),
                   settle_steps=settle.get(best), untimed_steps=(settle.get(best) or 0) + warmup + (switch_steps.get(best) or 0),
                   repetitions=reps,
                   repetition_order=("interleaved: " + ", ".join(paths) + f" per repetition, {SWITCH_STEPS} untimed steps after each switch"
                                     if len(paths) > 1 else None),
                   solver_paths_ms_per_step_samples={k: [t / steps * 1e3 for t in v[1]] for k, v in path_results.items()}, ms_per_step_samples=[s / steps * 1e3 for s in samples],
                   ms_per_step_spread=[min(samples) / steps * 1e3, max(samples) / steps * 1e3],
                   ms_per_step_by_rank=([t / steps * 1e3 for t in path_per_rank[chosen]] if path_per_rank.get(chosen) else None),
                   solver_paths_ms_per_step={k: v[0] / steps * 1e3 for k, v in path_results.items()},
                   value_exact=(cells_total * steps / path_results["exact"][0] if "exact" in path_results else None),
                   value_certified=(cells_total * steps / path_results["certified"][0] if "certified" in path_results else None),
                   solver_path_parity=path_parity or None,
                   halo_paths_ms_per_step={k: v[0] / steps * 1e3 for k, v in results.items()} if world > 1 else None,
                   # dominant kernel = compute_atmosphere_ocean_fluxes! (SURVEY.md §8d contract figure 128 B/cell)
                   roofline=roof(f"{ao_kernel} ({ao_what})", ao_bytes, cells_rank, ao_ms, traffic_key=ao_key,
                                 launches_timed=nrec,
                                 bytes_per_cell_note=("80 B read + 48 B written (SURVEY §8d, the contract figure of the solver) + Qs, Ql, Mp read "
                                                      "and JT, JS, SW written by the fused net-flux epilogue" if fused else
                                                      "80 B read + 48 B written (SURVEY §8d)") +
                                                     (" + 18.3 B of the JRA55 window read and 64 B of exchange fields written by the tail "
                                                      "workgroups (interpolate_atmosphere_state! of the next step, SURVEY §8d)" if tail_priced else ""),
                                 frac_solver_and_net_only_176_B_per_cell=(BYTES_AO_FUSED * cells_rank / (ao_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                                          if tail_priced else None),
                                 frac_at_contract_128_B_per_cell=BYTES_AO * cells_rank / (ao_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 measured=("HIP events around the kernel inside a separate event-bracketed pass over the timed schedule" if prof else
                                           "the solver stage alone, launched back to back on one set of inputs (cf_time_stage)"),
                                 avg_launch_ms_batches_sorted_by_trip_hints=prof_sorted[1][0] if prof_sorted else None,
                                 avg_launch_ms_back_to_back_same_inputs=ao_ms_alone),
                   roofline_interpolate=roof("interpolate_kernel (interpolate_atmosphere_state!)", BYTES_INTERP, cells_rank, interp_ms),
                   roofline_net_fluxes=(roof("net_stress_kernel (compute_net_ocean_fluxes!: the two face stresses)", BYTES_STRESS, nx * ny, net_ms)
                                        if fused else roof("net_flux_kernel (compute_net_ocean_fluxes!)", BYTES_NET, nx * ny, net_ms)),
                   stages_ms=dict(interpolate_tiled_standalone=interp_ms,
                                  interpolate_background_standalone=interp_bg_ms if mode == "on" else None,
                                  ao_fluxes=ao_ms, net_fluxes=net_ms, ao_fluxes_standalone=ao_ms_alone,
                                  net_fluxes_standalone=net_ms_alone),
                   device_copy_GBs=2 * copy_bytes / (copy_ms * 1e-3) / 1e9,
                   parity="vs reference: unpinned (self-consistent restatements only; see DESIGN.md)")
        # the ceiling that binds the solver: FP64 VALU issue
        k_sq = sq.get(ao_key)
        if k_sq:
            insts = k_sq["SQ_INSTS_VALU"]
            rate = insts / (ao_ms * 1e-3)
            out["roofline_fp64_valu"] = dict(
                bound="fp64_valu_issue", kernel=ao_kernel, achieved=rate / 1e9, peak=FP64_ISSUE_PEAK / 1e9, unit="G wave-instr/s",
                frac=rate / FP64_ISSUE_PEAK, valu_instructions_per_launch=insts, instructions_source=sq_source,
                valu_busy_of_wave_lifetime=k_sq.get("valu_busy"),
                note=("peak = 1024 SIMDs x 2.4 GHz / 4 cycles per FP64 wave-instruction; during this loop rocm-smi reads sclk "
                      "2.36 GHz at 1160 of 1400 W (profiles/r04_clock_watch.log) — a pure v_fma_f64 stream on every SIMD is "
                      "power-limited to about 1.9 GHz = 480 G wave-instr/s (scratch/ubench_valu.hip), the solver is not; every "
                      "VALU instruction is counted at the FP64 rate (the SQ counters book an integer VALU instruction at one "
                      "quad cycle as well: 28.5 M active quads for 27.0 M instructions)"))
        # strong-scaling projection from single-GPU measurements of one rank's slab (tools/slab_curve.py): a PROJECTION,
        # labelled as such — the driver computes the real curve from its own N-GPU runs
        try:
            sc = json.load(open(os.path.join(ROOT, "profiles", next(f for f in ("r06_slab_curve.json", "r05_slab_curve.json", "r04f_slab_curve.json")
                                                                         if os.path.exists(os.path.join(ROOT, "profiles", f))))))
            if canonical and world == 1:
                out["projected_scaling"] = dict(
                    kind="projection from one GPU, not a multi-GPU measurement",
                    speedup_before_halo_rows={k: round(v, 3) for k, v in sc["projected_speedup_before_halos"].items()},
                    slab_ms_per_step={k: round(v["ms_per_step"], 5) for k, v in sc["slabs"].items()},
                    source="committed: profiles/r06_slab_curve.json, else r05 / r04f (python bench.py --ny 560/280/140/70 on one MI355X, tools/slab_curve.py)",
                    note=sc["note"])
        except Exception:
            pass
        if a.share_device:
            out["rehearsal"] = f"{world} ranks time-sharing ONE device: a test of the N-rank code path, not a scaling number"
        if not a.no_cpu_baseline and world == 1:
            case_np = dict(ocean=ocean_np[0], src=src_np, weights=w_np)
            out["cpu_baseline"], ref = cpu_baseline(case_np, params, nx, ny, h, a.cpu_seconds)
            # The reference itself, where this box can run it (never in the build image): its CPU()
            # path on the same inputs becomes the cpu_baseline (the port stays beside it), and the upstream vectors are produced so
            # that tests/test_upstream_pin.py stops skipping.  Elsewhere the line says so instead of saying nothing.
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import upstream_probe
            info = upstream_probe.probe()
            out["julia_probe"] = info["status"]
            if info["status"] == "present":
                out["julia_probe_versions"] = info["versions"]
                out["upstream_vectors"] = upstream_probe.ensure_upstream_vectors()
                ref_rec = upstream_probe.time_reference_cpu(case_np, nx, ny, h, seconds=a.cpu_seconds) if (a.config == "ocean" and not tripolar) else None
                if ref_rec is not None:
                    out["cpu_baseline_port"] = out["cpu_baseline"]
                    out["cpu_baseline"] = ref_rec
                if upstream_probe.upstream_vectors_present():
                    out["parity"] = (f"vs {upstream_probe.reference_label(info)}: vectors in tests/golden/upstream/ "
                                     f"(python -m pytest tests/test_upstream_pin.py holds the oracle and the HIP path to them at 1e-6)")
            elif info["detail"]:
                out["julia_probe_detail"] = info["detail"]
            if a.config == "ocean" and not tripolar:
                worst = measured_parity(ctx, ref, dict(src=src, weights=w, ocean=states[0]), nx, ny, h)
                out["parity_measured"] = dict(worst_scaled_error=worst, max=max(worst.values()),
                                              metric="|got - oracle| / max(|oracle|, field scale), one cf_update_state on the cpu_baseline leg's inputs",
                                              against="oracle/coflux_oracle.c (CPU restatement; parity vs the reference itself: unpinned)")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
