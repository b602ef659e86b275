"""world_size-2 `gloo` tests of the latitude-slab sharding: slab bounds, the one-row
halo exchange of the ocean surface state, and that two slabs + exchanged halos reproduce the
single-domain oracle result row for row (no data-path collective besides the neighbour rows)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as orc
import util
from coflux import interface_computations as ic
from coflux import synthetic as syn
from coflux.distributed import exchange_halo_rows_torch, halo_row_slices, slab_bounds

NX, NY, H = 48, 20, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, rows, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        j0, j1 = slab_bounds(NY, rank, world)
        ny = j1 - j0
        full = syn.ocean_state(NX, NY, H, H)  # the global field every rank can regenerate
        mine = syn.ocean_state(NX, ny, H, H, ny_global=NY, j_offset=j0)
        tensors = []
        for k in ("T", "S", "u", "v"):
            a = mine[k].copy()
            if rank > 0:
                a[:H] = np.nan  # halos owned by the neighbour are poisoned …
            if rank < world - 1:
                a[H + ny:] = np.nan
            tensors.append(torch.from_numpy(a))
        exchange_halo_rows_torch(tensors, ny, H, rows)  # … and must come back over the wire
        ok = True
        for k, t in zip(("T", "S", "u", "v"), tensors):
            a = t.numpy()
            lo = H - rows if rank > 0 else 0
            hi = H + ny + rows if rank < world - 1 else ny + 2 * H
            ok &= np.array_equal(a[lo:hi], full[k][j0 + lo:j0 + hi])
        # flux solve on the slab with exchanged halos == the same rows of the global solve
        if rows >= 2:
            g = orc.make_grid(NX, ny, H, H, 1)
            params = ic.flux_params()
            src = syn.jra55_snapshots(2)
            fi, fj, phi = syn.latlon_fractional_indices(NX, ny, H, H, ny_global=NY, j_offset=j0)
            at = orc.interpolate_atmosphere_state(g, src, dict(separable=True, fi=fi, fj=fj), 0, 1, 0.37)
            oc = {k: np.nan_to_num(t.numpy()) for k, t in zip(("T", "S", "u", "v"), tensors)}
            oc["mask"] = mine["mask"]
            fl = orc.compute_atmosphere_ocean_fluxes(g, params, oc, at)
            out[rank] = {k: v[H - 1:H + ny + 1].copy() for k, v in fl.items()}
        out[f"ok{rank}"] = bool(ok)
    finally:
        dist.destroy_process_group()


def _run(rows):
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, rows, out), nprocs=2, join=True)
    return dict(out)


def test_slab_bounds_partition_every_row_once():
    for ny in (560, 180, 7):
        for world in (1, 2, 3, 4, 8):
            b = [slab_bounds(ny, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == ny
            assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
            sizes = [j1 - j0 for j0, j1 in b]
            assert max(sizes) - min(sizes) <= 1
    assert slab_bounds(560, 3, 8) == (210, 280)  # Partition(1,8): 70 rows per GPU, pbs_launch.sh:51


def test_halo_row_slices():
    ss, rs, sn, rn = halo_row_slices(10, 3, 2)
    assert (ss, rs, sn, rn) == (slice(3, 5), slice(1, 3), slice(11, 13), slice(13, 15))


def test_two_rank_one_row_halo_exchange_gloo():
    out = _run(1)
    assert out["ok0"] and out["ok1"]


def test_two_rank_slabs_reproduce_the_global_fluxes_gloo():
    out = _run(2)  # ring = 1 needs v[j+1] of the ring row ⇒ two rows
    assert out["ok0"] and out["ok1"]
    case = util.build_case(NX, NY, H, H)
    g = orc.make_grid(NX, NY, H, H, 1)
    at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
    ref = orc.compute_atmosphere_ocean_fluxes(g, ic.flux_params(), case["ocean"], at)
    for rank in (0, 1):
        j0, j1 = slab_bounds(NY, rank, 2)
        for k, v in out[rank].items():
            lo = 1 if rank == 0 else 0  # the outermost ring rows read un-exchanged outer halos: skip
            hi = v.shape[0] - (1 if rank == 1 else 0)
            np.testing.assert_array_equal(v[lo:hi], ref[k][H - 1 + j0 + lo:H - 1 + j0 + hi], err_msg=f"{rank} {k}")


class _FakeContext:
    """What SlabHaloExchanger touches of a FluxContext; rank `fail_rank`'s device has no fine-grained memory."""

    def __init__(self, rank, fail_rank):
        from types import SimpleNamespace
        self.grid = SimpleNamespace(ring=1)
        self.rank, self.fail_rank, self.calls = rank, fail_rank, []

    def peer_halo_export(self, max_fields, max_rows):
        self.calls.append("export")
        if self.rank == self.fail_rank:
            raise RuntimeError("cf_peer_halo_export: CF_ERR_COMM (no fine-grained device memory)")
        return b"handle-%d" % self.rank

    def peer_halo_connect(self, south, north, rank, world):
        self.calls.append("connect")

    def comm_init(self, ident, rank, world):
        self.calls.append(("comm_init", ident, rank, world))


def _fallback_worker(rank, world, port, fail_rank, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from coflux import distributed, runtime
        runtime.comm_unique_id = lambda: b"unique-id"      # (the real one needs the HIP library)
        ctx = _FakeContext(rank, fail_rank)
        ex = distributed.SlabHaloExchanger(ctx, 10, 3, backend="peer")
        out[rank] = (ex.backend, list(ctx.calls), getattr(ex, "peer_fallback_reason", None))
    finally:
        dist.destroy_process_group()


def test_peer_backend_falls_back_to_rccl_on_every_rank_together_gloo():
    """a rank whose cf_peer_halo_export fails must not leave the others blocked in the handle gather — the
    ranks agree on the outcome and switch to the RCCL exchange together; without a failure they connect the mailboxes."""
    for fail_rank, want in ((1, "rccl"), (-1, "peer")):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_fallback_worker, args=(2, _free_port(), fail_rank, out), nprocs=2, join=True)
        for rank in (0, 1):
            backend, calls, reason = out[rank]
            assert backend == want, (fail_rank, rank, backend)
            if want == "rccl":
                assert ("comm_init", b"unique-id", rank, 2) in calls and "connect" not in calls
                assert "rank 1" in reason and "CF_ERR_COMM" in reason
            else:
                assert calls == ["export", "connect"] and reason is None
