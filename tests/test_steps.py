"""GPU tests of what surrounds update_state! in a multi-step, multi-slab run (include/coflux.h): the C-side step
loop cf_time_steps, the pipelined interpolation (cf_prefetch_atmosphere_state), the peer-direct halo rows
(cf_peer_halo_* — two slabs on ONE device here: in one process, and in two processes through HIP IPC), and the
tripolar fold.  Everything is compared bit for bit with the host-driven three-launch path / the single-domain
state, which the parity tests in turn hold against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch

import util
from coflux import abi
from coflux import interface_computations as ic
from coflux import synthetic as syn
from coflux.distributed import fold_north_halo_torch, slab_bounds
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext

pytestmark = pytest.mark.gpu

NX, NY, H = 192, 48, 4
INC = 1.0 / 9.0


def _setup(nx=NX, ny=NY, ny_global=None, j_offset=0, n_levels=4, device=0):
    P = ic.flux_params()
    ctx = FluxContext(nx, ny, H, H, P, ring=1, device=device)
    o0 = syn.ocean_state(nx, ny, H, H, ny_global=ny_global, j_offset=j_offset)
    o1 = syn.evolved_ocean_state(o0, nx, ny, H, H, 1, ny_global=ny_global, j_offset=j_offset)
    states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in (o0, o1)]
    states[1]["mask"] = states[0]["mask"]
    src = {k: ctx.to_device(v) for k, v in syn.jra55_snapshots(n_levels).items()}
    fi, fj, phi = syn.latlon_fractional_indices(nx, ny, H, H, ny_global=ny_global, j_offset=j_offset)
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    return ctx, states, src, w, (o0, o1)


def _host_steps(ctx, states, src, w, n, n_levels=4):
    """The reference sequence: n × update_state! driven from the host, one set of exchange fields."""
    atmos, fl, net = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
    for s in range(n):
        tot = s * INC
        l1 = int(tot) % n_levels
        ctx.update_state(src, w, states[s % 2], atmos, fl, net, level1=l1, level2=(l1 + 1) % n_levels,
                         time_fraction=tot - int(tot))
    ctx.sync()
    return atmos, fl, net


@pytest.mark.parametrize("pipeline", [False, True, "merged", "tail"])
def test_time_steps_reproduces_host_loop_bitwise(pipeline):
    """cf_time_steps (C loop, advancing clock through a 4-snapshot window, alternating ocean states, optionally
    with the next step's interpolation on the auxiliary stream, or — CF_OPT_MERGED_PREFETCH — inside the current step's
    face-stress launch) == the host-driven cf_update_state loop."""
    n = 21   # crosses two snapshot boundaries (9 steps per snapshot interval)
    ctx, states, src, w, _ = _setup()
    ref_atmos, ref_fl, ref_net = _host_steps(ctx, states, src, w, n)
    if pipeline in ("merged", "tail"):
        ctx.set_option(abi.OPT_MERGED_PREFETCH, 1 if pipeline == "merged" else 2)
    sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2 if pipeline else 1)]
    fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
    sched = ctx.make_schedule(states, sets, first_level=0, time_fraction=0.0, time_fraction_increment=INC, pipeline=bool(pipeline))
    ctx.time_steps(0, 8, sched, src, w, fl, net)      # in two calls: the step counter carries the clock
    ctx.time_steps(8, n - 8, sched, src, w, fl, net)
    ctx.sync()
    last = sets[(n - 1) % len(sets)]
    for k in EXCHANGE_NAMES:
        assert torch.equal(last[k], ref_atmos[k]), k
    for k in FLUX_NAMES:
        assert torch.equal(fl[k], ref_fl[k]), k
    for k in NET_NAMES:
        assert torch.equal(net[k], ref_net[k]), k
    ctx.close()


def test_prefetch_mismatch_is_ignored_not_used():
    """A prefetched atmosphere state for ANOTHER time is never consumed by cf_update_state."""
    ctx, states, src, w, _ = _setup()
    a, fl, net = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
    ctx.update_state(src, w, states[0], a, fl, net, level1=1, level2=2, time_fraction=0.25)
    ctx.sync()
    want = {k: v.clone() for k, v in a.items()}
    ctx.prefetch_atmosphere_state(src, w, a, level1=0, level2=1, time_fraction=0.5)      # wrong time for the next call
    ctx.update_state(src, w, states[0], a, fl, net, level1=1, level2=2, time_fraction=0.25)
    ctx.sync()
    for k in EXCHANGE_NAMES:
        assert torch.equal(a[k], want[k]), k
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_request_flushed_by_sync_is_ordered_behind_the_readers_of_its_set(mode):
    """A request that waits for a solver launch (cf_prefetch_atmosphere_state) and is flushed by cf_sync instead goes to the
    auxiliary stream.  With CF_OPT_MERGED_PREFETCH its gate event is recorded at the flush, not at the request: the
    interpolation must still run behind every kernel already queued that reads the set it overwrites, and the state it
    leaves must be consumed as if it had been interpolated in its own step."""
    ctx, states, src, w, _ = _setup()
    ref = ctx.field_set(EXCHANGE_NAMES)
    ctx.interpolate_atmosphere_state(src, w, ref, level1=1, level2=2, time_fraction=0.75)
    rfl, rnet = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
    ctx.update_state(src, w, states[1], ref, rfl, rnet, level1=1, level2=2, time_fraction=0.75)
    ctx.sync()
    ctx.set_option(abi.OPT_MERGED_PREFETCH, mode)
    a, b = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(EXCHANGE_NAMES)
    fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
    for _ in range(3):   # work queued on the main stream that READS set b (the net fluxes read Qs, Ql, Mp)
        ctx.update_state(src, w, states[0], b, fl, net, level1=0, level2=1, time_fraction=0.25)
    want_fl = {k: v.clone() for k, v in fl.items()}
    ctx.prefetch_atmosphere_state(src, w, b, level1=1, level2=2, time_fraction=0.75)   # overwrites b; no solver launch follows
    ctx.sync()                                                                          # the flush
    for k in FLUX_NAMES:   # the three queued steps saw the OLD contents of b
        assert torch.equal(fl[k], want_fl[k]), k
    for k in EXCHANGE_NAMES:
        assert torch.equal(b[k], ref[k]), k
    ctx.update_state(src, w, states[1], b, fl, net, level1=1, level2=2, time_fraction=0.75)   # consumes the pending state
    ctx.sync()
    for k in FLUX_NAMES:
        assert torch.equal(fl[k], rfl[k]), k
    for k in NET_NAMES:
        assert torch.equal(net[k], rnet[k]), k
    ctx.close()


def _slab(rank, world, device=0):
    j0, j1 = slab_bounds(NY, rank, world)
    ctx, states, src, w, np_states = _setup(NX, j1 - j0, ny_global=NY, j_offset=j0, device=device)
    return ctx, states, src, w, np_states, j0, j1


def _poison_halos(fields, ny, rank, world):
    for f in fields:
        if rank > 0:
            f[:H] = float("nan")
        if rank < world - 1:
            f[H + ny:] = float("nan")


def test_peer_halo_rows_two_slabs_in_one_process():
    """Two latitude slabs as two contexts on one device: mailboxes are handed over inside the process (no IPC
    needed), rows travel by the peer kernel, and the slab solve with exchanged halos equals the single-domain solve."""
    world = 2
    slabs = [_slab(r, world) for r in range(world)]
    torch.cuda.synchronize()
    for s in slabs:   # each slab on its own stream: the two exchange kernels wait for each other and must overlap
        s[0]._check(s[0].lib.cf_set_stream(s[0]._h, None), "cf_set_stream")
    handles = [s[0].peer_halo_export(4, 2) for s in slabs]
    for r, s in enumerate(slabs):
        s[0].peer_halo_connect(handles[r - 1] if r > 0 else None, handles[r + 1] if r < world - 1 else None, r, world)
    full = syn.ocean_state(NX, NY, H, H)
    for step in range(3):          # sequence numbers and mailbox parity advance
        for r, (ctx, states, *_rest, j0, j1) in enumerate(slabs):
            fields = [states[0][k] for k in ("T", "S", "u", "v")]
            _poison_halos(fields, j1 - j0, r, world)
        torch.cuda.synchronize()
        for r, (ctx, states, *_rest) in enumerate(slabs):      # both launched before either is waited for
            ctx.halo_exchange_rows_peer([states[0][k] for k in ("T", "S", "u", "v")], rows=2)
        for r, (ctx, states, *_rest, j0, j1) in enumerate(slabs):
            ctx.sync()
            ny = j1 - j0
            lo = H - 2 if r > 0 else 0
            hi = H + ny + 2 if r < world - 1 else ny + 2 * H
            for k in ("T", "S", "u", "v"):
                got = states[0][k].cpu().numpy()
                np.testing.assert_array_equal(got[lo:hi], full[k][j0 + lo:j0 + hi], err_msg=f"step {step} rank {r} {k}")
    # slab solve == rows of the single-domain solve (ring rows included on the seam)
    gctx, gstates, gsrc, gw, _ = _setup()
    ga, gf, gn = gctx.field_set(EXCHANGE_NAMES), gctx.field_set(FLUX_NAMES), gctx.field_set(NET_NAMES)
    gctx.update_state(gsrc, gw, gstates[0], ga, gf, gn, time_fraction=0.37)
    gctx.sync()
    for r, (ctx, states, src, w, _np, j0, j1) in enumerate(slabs):
        a, f, n = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
        torch.cuda.synchronize()     # the zero fills ran on torch's stream, the context has its own
        ctx.update_state(src, w, states[0], a, f, n, time_fraction=0.37)
        ctx.sync()
        ny = j1 - j0
        for k in FLUX_NAMES:     # interior + ring rows
            assert torch.equal(f[k][H - 1:H + ny + 1, H - 1:H + NX + 1], gf[k][j0 + H - 1:j0 + H + ny + 1, H - 1:H + NX + 1]), (r, k)
        for k in ("u", "v", "T", "S"):
            assert torch.equal(n[k][H:H + ny, H:H + NX], gn[k][j0 + H:j0 + H + ny, H:H + NX]), (r, k)
    for s in slabs:
        s[0].close()
    gctx.close()


def _peer_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from coflux.distributed import SlabHaloExchanger
        ctx, states, src, w, _np, j0, j1 = _slab(rank, world, device=0)   # both ranks on the one device of the box
        ny = j1 - j0
        ex = SlabHaloExchanger(ctx, ny, H, backend="peer")                 # handles travel through torch.distributed
        full = syn.ocean_state(NX, NY, H, H)
        ok = True
        fields = [states[0][k] for k in ("T", "S", "u", "v")]
        for step in range(4):
            _poison_halos(fields, ny, rank, world)
            ex(fields)
            ctx.sync()
            lo = H - 2 if rank > 0 else 0
            hi = H + ny + 2 if rank < world - 1 else ny + 2 * H
            for k, f in zip(("T", "S", "u", "v"), fields):
                ok &= bool(np.array_equal(f.cpu().numpy()[lo:hi], full[k][j0 + lo:j0 + hi]))
        # and inside the C step loop
        sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2)]
        fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
        sched = ctx.make_schedule(states, sets, time_fraction_increment=INC, pipeline=True, halo_backend=abi.HALO_PEER, halo_rows=2)
        ctx.time_steps(0, 6, sched, src, w, fl, net)
        ctx.sync()
        ok &= bool(torch.isfinite(net["T"][H:H + ny, H:H + NX]).all())
        out[rank] = ok
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_peer_halo_rows_two_processes_hip_ipc():
    """The same exchange between two PROCESSES (one rank per process, as on a multi-GPU node), mailboxes mapped through
    hipIpcGetMemHandle / hipIpcOpenMemHandle; both ranks share the one device of the test box."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctxm = mp.get_context("spawn")
    out = ctxm.Manager().dict()
    procs = [ctxm.Process(target=_peer_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"peer worker exit code {p.exitcode}"
    assert out.get(0) is True and out.get(1) is True, dict(out)


@pytest.mark.parametrize("rows", [1, 2])
def test_fold_north_halo_matches_host_fold(rows):
    """cf_fold_north_halo vs the torch/NumPy statement of Oceananigans' zipper (centres, x-faces, y-faces; vector sign)."""
    nx, ny = 96, 20
    ctx = FluxContext(nx, ny, H, H, ic.flux_params(), ring=1)
    rng = np.random.default_rng(3)
    host = [rng.normal(size=ctx.shape) for _ in range(4)]
    dev = [ctx.to_device(a) for a in host]
    locs = [abi.FOLD_CENTER, abi.FOLD_CENTER, abi.FOLD_X_FACE, abi.FOLD_Y_FACE]
    signs = [1.0, 1.0, -1.0, -1.0]
    ctx.fold_north_halo(dev, locs, signs, rows=rows)
    ctx.sync()
    for a, d, loc, sg in zip(host, dev, ("center", "center", "x_face", "y_face"), signs):
        t = torch.from_numpy(a.copy())
        fold_north_halo_torch(t, nx, ny, H, H, rows, loc, sg)
        np.testing.assert_array_equal(d.cpu().numpy(), t.numpy())
    ctx.close()


def test_fused_net_epilogue_is_bitwise_the_three_launch_sequence():
    """CF_OPT_FUSED_NET: cell-local net fluxes in the solver's epilogue + the face-stress kernel == the separate
    compute_net_ocean_fluxes! kernel, bit for bit (shared arithmetic with contraction off), with and without sea ice."""
    ctx, states, src, w, np_states = _setup()
    ice = {k: ctx.to_device(np_states[0]["ice_" + k]) for k in ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")}
    for fluxes in (None, ic.ncar_atmosphere_ocean_fluxes()):   # the round-3 kernel; CoefficientBasedFluxes in round 2's (fused by default too)
        if fluxes is not None:
            ctx.set_flux_params(ic.flux_params(fluxes))
        for use_ice in (None, ice):
            outs = []
            for fused in (0, 1, 2):
                ctx.set_option(abi.OPT_FUSED_NET, fused)
                a, fl, net = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
                ctx.update_state(src, w, states[0], a, fl, net, ice=use_ice, time_fraction=0.37)
                ctx.sync()
                outs.append((fl, net))
            for other in outs[1:]:
                for k in FLUX_NAMES:
                    assert torch.equal(outs[0][0][k], other[0][k]), k
                for k in NET_NAMES:
                    assert torch.equal(outs[0][1][k], other[1][k]), k
    ctx.close()


def _two_device_worker(rank, world, port, backend, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from coflux.distributed import SlabHaloExchanger
        ctx, states, src, w, _np, j0, j1 = _slab(rank, world, device=rank)       # one rank per GPU
        ny = j1 - j0
        ex = SlabHaloExchanger(ctx, ny, H, backend=backend)
        full = syn.ocean_state(NX, NY, H, H)
        fields = [states[0][k] for k in ("T", "S", "u", "v")]
        ok = True
        for step in range(3):
            _poison_halos(fields, ny, rank, world)
            torch.cuda.synchronize()
            ex(fields)
            ctx.sync()
            lo = H - 2 if rank > 0 else 0
            hi = H + ny + 2 if rank < world - 1 else ny + 2 * H
            for k, f in zip(("T", "S", "u", "v"), fields):
                ok &= bool(np.array_equal(f.cpu().numpy()[lo:hi], full[k][j0 + lo:j0 + hi]))
        # the all-reduce of the salinity normaliser over the two ranks (rccl backend only: it owns the communicator)
        if backend == "rccl":
            flux = ctx.to_device(np.where(_np[0]["mask"] != 0, 1.0 + rank, 0.0))
            mean = ctx.zeros()[:1].contiguous()
            ctx.normalize_salinity_flux(flux, states[0]["mask"], mean_out=mean)
            ctx.sync()
            wet = [int((syn.ocean_state(NX, b - a, H, H, ny_global=NY, j_offset=a)["mask"][H:H + b - a, H:H + NX] != 0).sum())
                   for a, b in (slab_bounds(NY, r, world) for r in range(world))]
            want = (1.0 * wet[0] + 2.0 * wet[1]) / (wet[0] + wet[1])
            ok &= abs(float(mean.cpu()) - want) < 1e-12
        out[rank] = ok
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices (runs on the multi-GPU node, not on the 1-GPU test box)")
@pytest.mark.parametrize("backend", ["rccl", "peer"])
def test_halo_rows_between_two_devices(backend):
    """One rank per GPU: native RCCL grouped send/recv (and its all-reduce in the salinity normaliser), and the peer-direct
    mailboxes over xGMI, against the globally indexed state."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctxm = mp.get_context("spawn")
    out = ctxm.Manager().dict()
    procs = [ctxm.Process(target=_two_device_worker, args=(r, 2, port, backend, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    assert out.get(0) is True and out.get(1) is True, dict(out)


@pytest.mark.gpu
def test_bench_n_rank_code_path_rehearsal():
    """bench.py's N-rank path (rank set-up, halo backend verification, the agreed settle loop, cf_time_steps with
    peer-direct halo rows, max-over-ranks timing, one JSON line from rank 0) rehearsed with two ranks time-sharing
    this box's device — the driver's multi-GPU run is the first time it meets N devices."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--steps", "40",
                        "--warmup", "5", "--nx", "360", "--ny", "120", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 40 and "rehearsal" in line
    assert line["config"]["halo_verified"] == {"peer": True} and line["config"]["rows_per_rank"] == 60
    assert line["value"] > 0 and line["config"]["step_loop"].startswith("cf_time_steps")
    # what a first contact with N devices must show: per-rank step times, how many steps ran before the
    # timed region, the spread of the repetitions — and ONE solver path, the exact one, at N > 1 (the points
    # of a scaling series are one algorithm)
    assert len(line["ms_per_step_by_rank"]) == 2 and 3 <= line["repetitions"] <= 9 and len(line["ms_per_step_spread"]) == 2
    assert line["repetition_order"] is None and len(line["solver_paths_ms_per_step_samples"]["exact"]) == line["repetitions"]
    assert line["untimed_steps"] >= line["settle_steps"] + 5 and set(line["solver_paths_ms_per_step"]) == {"exact"}
    assert line["config"]["solver_path"] == "exact" and line["value"] == line["value_exact"] and line["value_certified"] is None
    assert "rccl_comm_ranks" in line["config"]
    # the halo rows rode in the solver launches (CF_OPT_HALO_IN_SOLVER_LAUNCH), after bench.py had proven them against the exchange kernel
    assert line["config"]["halo_in_solver_launch"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("grid", ["latlon", "tripolar"])
def test_bench_selftest_two_ranks_sharing_the_device(grid):
    """bench.py --gpus 2 --selftest: halo verification, ten steps of cf_time_steps with halo rows, the gathered surface
    against the single-domain oracle — the diagnosable first contact with a multi-GPU node, rehearsed with two processes
    on this box's one device (peer-direct rows over HIP IPC)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    size = ["--nx", "360", "--ny", "120"] if grid == "latlon" else ["--nx", "360", "--ny", "180", "--grid", "tripolar"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--selftest"] + size,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["selftest"] == "ok" and line["stages"][0] == "halo_backends" and line["stages"][-1] == "compare_with_oracle"
    assert max(line["worst_scaled_error_vs_oracle"].values()) <= 1e-9
    assert line["halo_in_solver_launch_proven"] is True      # (the riders reproduce the exchange kernel's steps here: reported, never fatal)


@pytest.mark.gpu
def test_bench_selftest_names_the_failing_stage():
    """A halo backend that cannot work here (RCCL between two processes sharing one device is filtered out, so asking for it
    leaves no backend) must end with a non-zero exit and the stage's name in the JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--selftest", "--halo-backend", "rccl",
                        "--nx", "360", "--ny", "120"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["selftest"] == "failed" and line["failed_stage"] == "halo_backends"


def test_sea_ice_step_with_tail_workgroups_is_bitwise_the_plain_step():
    """cf_update_state_sea_ice with CF_OPT_MERGED_PREFETCH = 2: this step's face stresses and the requested next-step
    interpolation ride in the tail workgroups of the sea-ice interface launch.  Same bits as the plain sequence in every
    output — exchange fields, ocean interface and net fluxes (incl. the stresses), sea-ice interface fluxes and skin
    temperature, top / bottom heat — over a host loop that crosses a snapshot boundary."""
    n, n_levels = 12, 4
    results = []
    for tail in (False, True):
        ctx, states, src, w, (o0, _) = _setup(n_levels=n_levels)
        ctx.set_sea_ice_formulation(ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes()))
        si = syn.sea_ice_state(NX, NY, H, H)
        ice = {k: ctx.to_device(o0["ice_" + k]) for k in ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")}
        ice_state = dict(concentration=ice["concentration"], **{k: ctx.to_device(si[k]) for k in ("thickness", "top_temperature", "u", "v", "albedo")})
        sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2)]
        fl, net, ai = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES), ctx.field_set(FLUX_NAMES)
        net_ice = ctx.field_set(("top_heat", "bottom_heat"))
        ai["temperature"].copy_(ice_state["top_temperature"])
        ice_state["top_temperature"] = ai["temperature"]      # the skin temperature is carried from step to step
        if tail:
            ctx.set_option(abi.OPT_MERGED_PREFETCH, 2)
        for s in range(n):
            tot = s * INC
            l1 = int(tot) % n_levels
            if tail:
                nxt = (s + 1) * INC
                l1n = int(nxt) % n_levels
                ctx.prefetch_atmosphere_state(src, w, sets[(s + 1) % 2], level1=l1n, level2=(l1n + 1) % n_levels, time_fraction=nxt - int(nxt))
            ctx.update_state_sea_ice(src, w, states[s % 2], sets[s % 2], fl, net, ice, ice_state, ai, net_ice,
                                     level1=l1, level2=(l1 + 1) % n_levels, time_fraction=tot - int(tot))
        ctx.sync()
        out = {}
        for name, d in (("atmos", sets[(n - 1) % 2]), ("fl", fl), ("net", net), ("ai", ai), ("net_ice", net_ice)):
            for k, v in d.items():
                out[f"{name}.{k}"] = v.clone()
        results.append(out)
        ctx.close()
    for k in results[0]:
        assert torch.equal(results[0][k], results[1][k]), k


@pytest.mark.gpu
def test_ice_free_cells_zero_mode_changes_nothing_where_there_is_ice():
    """CF_OPT_ICE_FREE_CELLS = CF_ICE_FREE_ZERO (opt-in): wet cells with ℵ = 0 and hᵢ = 0 get zero_interface_state instead of
    the interface iteration.  Against the default mode on the same inputs, through cf_update_state_sea_ice with every rider:
    the five net ocean fields bitwise, the atmosphere–sea-ice interface fluxes and the net sea-ice fluxes bitwise wherever
    there is ice; open water reports zero fluxes, zero iterations and its input skin temperature."""
    results = {}
    for mode in (abi.ICE_FREE_ITERATE, abi.ICE_FREE_ZERO):
        ctx, states, src, w, (o0, _) = _setup(n_levels=4)
        ctx.set_sea_ice_formulation(ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes()))
        ctx.set_option(abi.OPT_ICE_FREE_CELLS, mode)
        ctx.set_option(abi.OPT_MERGED_PREFETCH, 2)
        si = syn.sea_ice_state(NX, NY, H, H)
        conc = o0["ice_concentration"]
        assert (conc == 0).any() and (conc > 0).any()
        si["thickness"] = np.where(conc > 0, si["thickness"], 0.0)
        ice = {k: ctx.to_device(o0["ice_" + k]) for k in ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")}
        ice_state = dict(concentration=ice["concentration"], **{k: ctx.to_device(si[k]) for k in ("thickness", "top_temperature", "u", "v", "albedo")})
        atmos, fl, net = ctx.field_set(EXCHANGE_NAMES), ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
        ai = ctx.field_set(FLUX_NAMES)
        ai["iterations"] = ctx.zeros(torch.int32)
        net_ice = ctx.field_set(("top_heat", "bottom_heat"))
        for step in range(3):
            ctx.update_state_sea_ice(src, w, states[step % 2], atmos, fl, net, ice, ice_state, ai, net_ice, level1=0, level2=1,
                                     time_fraction=0.1 * step)
        ctx.sync()
        results[mode] = dict(net={k: v.cpu().numpy() for k, v in net.items()}, ai={k: v.cpu().numpy() for k, v in ai.items()},
                             net_ice={k: v.cpu().numpy() for k, v in net_ice.items()}, fl={k: v.cpu().numpy() for k, v in fl.items()})
        ctx.close()
    a, b = results[abi.ICE_FREE_ITERATE], results[abi.ICE_FREE_ZERO]
    inner = (slice(H, H + NY), slice(H, H + NX))
    icy = (conc > 0)[inner]
    water = (conc == 0)[inner] & (o0["mask"] != 0)[inner]
    for k in a["net"]:
        np.testing.assert_array_equal(a["net"][k], b["net"][k], err_msg="net ocean " + k)
    for k in a["fl"]:
        np.testing.assert_array_equal(a["fl"][k], b["fl"][k], err_msg="ocean interface " + k)
    for grp in ("ai", "net_ice"):
        for k in a[grp]:
            np.testing.assert_array_equal(a[grp][k][inner][icy], b[grp][k][inner][icy], err_msg=f"{grp}.{k} on ice")
    assert water.any() and np.all(b["ai"]["iterations"][inner][water] == 0) and np.all(a["ai"]["iterations"][inner][water] > 0)
    for k in ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum"):
        assert np.all(b["ai"][k][inner][water] == 0.0), k
    np.testing.assert_array_equal(b["ai"]["temperature"][inner][water], si["top_temperature"][inner][water])
