"""BASELINE configs 4–5: the tripolar grid for real — a fold row (tracer-point pivot, Oceananigans' zipper), the rotation
of the wind into the grid frame (experiments/OMIPSimulations/scripts/visualize/cache.jl:406-427), and the 1° tripolar
surface (360×180, OceanConfigurations/one_degree_tripolar.jl:48-51) sharded 2- and 4-way by latitude slab with the fold
applied locally on the last rank.  CPU: mesh / fold properties and world-size-2/4 gloo runs of the oracle;
GPU: the HIP path on slabs (peer-direct halo rows + cf_fold_north_halo inside cf_time_steps) against the single domain,
and the single domain against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as orc
import util
from coflux import abi
from coflux import interface_computations as ic
from coflux import synthetic as syn
from coflux.distributed import exchange_halo_rows_torch, fold_north_halo_torch, slab_bounds

NX, NY, H = 360, 180, 5        # TripolarGrid(size = (360, 180, Nz), halo = (5, 5, 4)), one_degree_tripolar.jl:32,48-51
FIELDS = ("T", "S", "u", "v")


def test_mesh_has_a_fold_and_unit_rotations():
    lam, phi, c, s = syn.tripolar_mesh(NX, NY)
    assert phi.max() < 90.0 and abs(phi.min() + 80.0) < 1.0
    np.testing.assert_allclose(phi[-1], phi[-1, ::-1], atol=1e-12)                     # the last row of centres IS the fold line
    assert np.max(np.abs((lam[-1] - lam[-1, ::-1] + 180.0) % 360.0 - 180.0)) < 1e-9
    np.testing.assert_allclose(c * c + s * s, 1.0, atol=1e-14)
    sea = np.ones(NX, bool)                                                           # (away from the two grid poles, which are land)
    for pole in (0, NX // 2):
        sea[np.arange(pole - 3, pole + 3) % NX] = False
    np.testing.assert_allclose(c[-1][sea], -c[-1, ::-1][sea], atol=1e-9)              # … where the i-axis reverses
    np.testing.assert_allclose(s[-1][sea], -s[-1, ::-1][sea], atol=1e-9)
    south = phi < 50.0
    assert np.max(np.abs(s[south])) < 1e-12 and np.min(c[south]) > 1 - 1e-12            # latitude–longitude below the cap
    assert np.all(np.diff(phi[:, 7]) > 0)                                              # rows march north


def test_fold_and_rotation_conventions_are_consistent():
    """A geographic vector field rotated into the grid frame at the halo cells (whose i-axis is the mirror cell's, reversed)
    equals the fold of the grid-frame field with sign −1; scalars fold with +1; NumPy and torch folds agree; the pivot row
    is its own image."""
    case = syn.tripolar_case(NX, NY, H, H)
    w = case["weights"]
    lam = np.deg2rad(w["fi"] * (360.0 / syn.JRA55_NX))
    phi = np.deg2rad(w["latitude"])
    uE, vN = np.cos(phi) * np.sin(2 * lam), 0.3 * np.sin(phi) + np.cos(lam)           # smooth geographic (E, N) components
    ug, vg = uE * w["cos_rot"] + vN * w["sin_rot"], -uE * w["sin_rot"] + vN * w["cos_rot"]
    for name, a, sign in (("u", ug, -1.0), ("v", vg, -1.0), ("scalar", uE, 1.0)):
        folded = a.copy()
        folded[H + NY:] = np.nan
        syn.fold_north(folded, NX, NY, H, H, 2, "center", sign)
        np.testing.assert_allclose(folded[H + NY:H + NY + 2], a[H + NY:H + NY + 2], atol=1e-12, err_msg=name)
        t = torch.from_numpy(a.copy())
        fold_north_halo_torch(t, NX, NY, H, H, 2, "center", sign)
        np.testing.assert_array_equal(t.numpy()[H + NY:H + NY + 2], folded[H + NY:H + NY + 2])
    row = uE[H + NY - 1, H:H + NX]
    np.testing.assert_allclose(row, row[::-1], atol=1e-12)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_fluxes(case):
    g = orc.make_grid(case["nx"], case["ny"], H, H, 1)
    at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
    return orc.compute_atmosphere_ocean_fluxes(g, ic.flux_params(ic.corrected_atmosphere_ocean_fluxes()), case["ocean"], at)


def _gloo_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        j0, j1 = slab_bounds(NY, rank, world)
        ny = j1 - j0
        case = syn.tripolar_case(NX, NY, H, H, j0=j0, j1=j1)
        tensors = []
        for k in FIELDS:                                   # everything a neighbour or the fold owns is poisoned …
            a = case["ocean"][k]
            if rank > 0:
                a[:H] = np.nan
            a[H + ny:] = np.nan
            tensors.append(torch.from_numpy(a))
        exchange_halo_rows_torch(tensors, ny, H, 2)         # … and comes back over the wire,
        if rank == world - 1:                               # or, on the last slab, through the fold of its own rows
            for k, t in zip(FIELDS, tensors):
                fold_north_halo_torch(t, NX, ny, H, H, 2, syn.FOLD_LOCATION[k], syn.FOLD_SIGN[k])
        for k in FIELDS:
            case["ocean"][k] = np.nan_to_num(case["ocean"][k])
        fl = _oracle_fluxes(case)
        out[rank] = {k: v[H - 1:H + ny + 1].copy() for k, v in fl.items()}
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_one_degree_tripolar_slabs_with_fold_reproduce_the_global_fluxes_gloo(world):
    """BASELINE config 4: 360×180 tripolar, latitude slabs on 2 and 4 ranks, two halo rows per seam, fold on the last rank."""
    out = mp.Manager().dict()
    mp.spawn(_gloo_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ref = _oracle_fluxes(syn.tripolar_case(NX, NY, H, H))
    assert np.abs(ref["x_momentum"][H + NY]).max() > 0          # the ring row beyond the fold is really computed
    for rank in range(world):
        j0, j1 = slab_bounds(NY, rank, world)
        for k, v in out[rank].items():
            lo = 1 if rank == 0 else 0                           # the southernmost ring row reads un-exchanged outer halos
            np.testing.assert_array_equal(v[lo:], ref[k][H - 1 + j0 + lo:H + j1 + 1], err_msg=f"{world} ranks, rank {rank}, {k}")


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
def _device_case(ctx, case):
    ocean = {k: ctx.to_device(case["ocean"][k]) for k in FIELDS + ("mask",)}
    src = {k: ctx.to_device(v) for k, v in case["src"].items()}
    w = {k: (ctx.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in case["weights"].items()}
    return ocean, src, w


def _hip_worker(rank, world, port, out):
    """One latitude slab of the 1° tripolar surface per PROCESS (all on the one device of the test box): peer-direct halo
    rows through HIP IPC on the seams, the fold on the last slab, one step of cf_time_steps."""
    from coflux.distributed import SlabHaloExchanger
    from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes())
        j0, j1 = slab_bounds(NY, rank, world)
        ny = j1 - j0
        case = syn.tripolar_case(NX, NY, H, H, j0=j0, j1=j1)
        for k in FIELDS:                                 # halos a neighbour or the fold must deliver
            if rank > 0:
                case["ocean"][k][:H] = np.nan
            case["ocean"][k][H + ny:] = np.nan
        ctx = FluxContext(NX, ny, H, H, P, ring=1)
        if world > 1:
            SlabHaloExchanger(ctx, ny, H, backend="peer")
        ocean, src, w = _device_case(ctx, case)
        sets = [ctx.field_set(EXCHANGE_NAMES)]
        fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
        sched = ctx.make_schedule([ocean], sets, halo_backend=abi.HALO_PEER if world > 1 else abi.HALO_NONE,
                                  halo_rows=2 if world > 1 else 0, fold_north=(rank == world - 1))
        if world > 1:
            dist.barrier()
        ctx.time_steps(0, 1, sched, src, w, fl, net)
        ctx.sync()
        out[rank] = dict(fluxes={k: fl[k].cpu().numpy() for k in FLUX_NAMES}, net={k: net[k].cpu().numpy() for k in ("u", "v", "T", "S")})
        if world > 1:
            dist.barrier()
        ctx.close()
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4])
def test_one_degree_tripolar_hip_path_slabs_with_fold(world):
    """BASELINE config 4 on the HIP path: the 1° tripolar surface as 1, 2 and 4 latitude slabs (one process per slab, as on a
    multi-GPU node; they share the test box's one device), against the oracle on the single domain."""
    from coflux.runtime import FLUX_NAMES
    P = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes())
    ref_case = syn.tripolar_case(NX, NY, H, H)
    g = orc.make_grid(NX, NY, H, H, 1)
    at = orc.interpolate_atmosphere_state(g, ref_case["src"], ref_case["weights"], 0, 1, 0.0)
    ref = orc.compute_atmosphere_ocean_fluxes(g, P, ref_case["ocean"], at)
    ref_net = orc.compute_net_ocean_fluxes(g, P, ref_case["ocean"], at, ref, weights=ref_case["weights"])
    ctxm = mp.get_context("spawn")
    out = ctxm.Manager().dict()
    port = _free_port()
    procs = [ctxm.Process(target=_hip_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"slab worker exit code {p.exitcode}"
    for r in range(world):
        j0, j1 = slab_bounds(NY, r, world)
        ny = j1 - j0
        lo = 1 if r == 0 else 0
        for k in FLUX_NAMES:                             # interior + ring rows, the row beyond the fold included
            got = out[r]["fluxes"][k][H - 1 + lo:H + ny + 1, H - 1:H + NX + 1]
            want = ref[k][H - 1 + j0 + lo:H + j1 + 1, H - 1:H + NX + 1]
            assert util.rel_err(got, want, util.FIELD_SCALE[k]) < 1e-9, (world, r, k)
        for k in ("u", "v", "T", "S"):
            got = out[r]["net"][k][H:H + ny, H:H + NX]
            assert util.rel_err(got, ref_net[k][H + j0:H + j1, H:H + NX], util.FIELD_SCALE[k]) < 1e-9, (world, r, "net." + k)
