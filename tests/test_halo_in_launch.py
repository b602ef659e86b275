"""CF_OPT_HALO_IN_SOLVER_LAUNCH: a step's peer-direct halo rows as rider workgroups of its solver launch,
the boundary chunks dispatched last and waiting for them.  One latitude slab per PROCESS (HIP IPC mailboxes, as on a
multi-GPU node; the ranks share the test box's one device), 2 and 4 ranks, lat-lon and tripolar (fold on the last rank):
the stepping loop with the option on must leave the same bits as with the exchange kernel of its own, the exchanges must
really have ridden, and the halo rows must really have been delivered (they are poisoned before every run)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from coflux import abi, interface_computations as ic, synthetic as syn
from coflux.distributed import slab_bounds

pytestmark = pytest.mark.gpu

H, NSTEPS = 5, 7
FIELDS = ("T", "S", "u", "v")
INC = 1200.0 / 10800.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case(grid, rank, world):
    """two ocean states of this rank's slab (one step apart), the weights, the source"""
    if grid == "tripolar":
        nx, ny_global = 360, 180
        j0, j1 = slab_bounds(ny_global, rank, world)
        tc = syn.tripolar_case(nx, ny_global, H, H, j0=j0, j1=j1)
        first = dict(tc["ocean"])
        evolved = syn.evolved_ocean_state(syn.ocean_state(nx, ny_global, H, H, latitude=(-80.0, 90.0)), nx, ny_global, H, H, 1)
        second = {}
        for k in FIELDS:
            g = evolved[k].copy()
            syn.fold_north(g, nx, ny_global, H, H, 2, syn.FOLD_LOCATION[k], syn.FOLD_SIGN[k])
            second[k] = np.ascontiguousarray(g[j0:j1 + 2 * H])
        second["mask"] = first["mask"]
        return nx, j1 - j0, [first, second], tc["weights"], tc["src"]
    nx, ny_global = 360, 120
    j0, j1 = slab_bounds(ny_global, rank, world)
    ny = j1 - j0
    first = syn.ocean_state(nx, ny, H, H, ny_global=ny_global, j_offset=j0)
    second = syn.evolved_ocean_state(first, nx, ny, H, H, 1, ny_global=ny_global, j_offset=j0)
    fi, fj, phi = syn.latlon_fractional_indices(nx, ny, H, H, ny_global=ny_global, j_offset=j0)
    return nx, ny, [first, second], dict(separable=True, fi=fi, fj=fj, latitude=phi), syn.jra55_snapshots(4)


def _worker(rank, world, port, grid, config, out):
    from coflux.distributed import SlabHaloExchanger
    from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fluxes = ic.corrected_atmosphere_ocean_fluxes() if config == "corrected" else ic.SimilarityTheoryFluxes()
        P = ic.flux_params(fluxes, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
        nx, ny, states_np, w_np, src_np = _case(grid, rank, world)
        ctx = FluxContext(nx, ny, H, H, P, ring=1)
        SlabHaloExchanger(ctx, ny, H, backend="peer")
        ctx.set_option(abi.OPT_MERGED_PREFETCH, 2)          # the stepping loop's launch: tail workgroups (and, below, the halo riders)
        states = [{k: ctx.to_device(s[k]) for k in FIELDS + ("mask",)} for s in states_np]
        states[1]["mask"] = states[0]["mask"]
        src = {k: ctx.to_device(v) for k, v in src_np.items()}
        w = {k: (ctx.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in w_np.items()}
        sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2)]
        fold = grid == "tripolar" and rank == world - 1
        results, stats = [], []
        for in_launch in (0, 1, 1):                          # (twice with the option: sequence numbers and counters carry on)
            ctx.set_option(abi.OPT_HALO_IN_SOLVER_LAUNCH, in_launch)
            for st in states:                                # halo rows a neighbour (or the fold) must deliver
                for k in FIELDS:
                    if rank > 0:
                        st[k][:H] = float("nan")
                    if rank < world - 1 or fold:
                        st[k][H + ny:] = float("nan")
            fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
            sched = ctx.make_schedule(states, sets, time_fraction_increment=INC, pipeline=True, halo_backend=abi.HALO_PEER, halo_rows=2,
                                      fold_north=fold)
            torch.cuda.synchronize()
            dist.barrier()
            before = ctx.peer_halo_stats()
            ctx.time_steps(0, NSTEPS, sched, src, w, fl, net)
            ctx.sync()
            after = ctx.peer_halo_stats()
            stats.append((after[0] - before[0], after[1] - before[1]))
            results.append({("f." + k): fl[k].cpu().numpy() for k in FLUX_NAMES} | {("n." + k): net[k].cpu().numpy() for k in ("u", "v", "T", "S")})
            dist.barrier()
        lo = H if rank == 0 else H - 1                        # (the southernmost ring row reads outer halos nobody exchanges)
        finite = all(bool(np.isfinite(v[lo:H + ny + 1, H - 1:H + nx + 1]).all()) for k, v in results[0].items() if k.startswith("f."))
        same = all(np.array_equal(results[0][k], results[n][k], equal_nan=True) for n in (1, 2) for k in results[0])
        out[rank] = dict(finite=finite, same=same, stats=stats, layout=bool(ctx.solver_latency_layout()))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grid,config", [("latlon", "default"), ("latlon", "corrected"), ("tripolar", "corrected")])
@pytest.mark.parametrize("world", [2, 4])
def test_halo_rows_in_the_solver_launch_leave_the_bits_of_the_exchange_kernel(world, grid, config):
    ctxm = mp.get_context("spawn")
    out = ctxm.Manager().dict()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, grid, config, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0, f"slab worker exit code {p.exitcode}"
    for r in range(world):
        res = out[r]
        assert res["finite"], (r, "a halo row was not delivered")
        assert res["same"], (r, "the riders left other bits than the exchange kernel")
        # every step exchanged once; with the option all but the call's last step (no request rides behind it: no tail
        # workgroups, hence the exchange kernel) rode in the solver launch
        assert res["stats"][0] == (NSTEPS, 0), res["stats"]
        assert res["stats"][1] == (NSTEPS, NSTEPS - 1) and res["stats"][2] == (NSTEPS, NSTEPS - 1), res["stats"]
    if config == "corrected":   # (the COARE profile on these small slabs runs the latency-layout kernels: their HALO variants are the ones exercised)
        assert any(out[r]["layout"] for r in range(world))
