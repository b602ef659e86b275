#!/usr/bin/env python
"""bench.py — surface-flux hot path throughput on MI355X (BASELINE.json metric).

A "step" is one update_state! of the coupled model's flux path over one synthetic surface:
(N>1: one-row halo exchange of the ocean surface state over RCCL) → JRA55 interpolation →
Monin–Obukhov solve → net ocean fluxes, all through the C ABI (libcoflux.so).
Workload at N = 1 is BASELINE.json configs[1]: the 1/4° 1440×560 surface, JRA55 atmosphere,
SimilarityTheory fluxes + Radiation, Float64, inputs resident in HBM before the timed region.
At N > 1 every rank owns one 1440×560 latitude slab of a 1440×(560·N) surface (weak scaling);
`--scaling strong` shards the fixed 1440×560 surface instead.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from coflux import abi, synthetic as syn  # noqa: E402
from coflux import interface_computations as ic  # noqa: E402
from coflux.distributed import SlabHaloExchanger, slab_bounds  # noqa: E402
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Algorithmic bytes per surface cell (derivation in DESIGN.md §4)
BYTES_AO = 128.0                       # compute_atmosphere_ocean_fluxes!: 80 read + 48 written
BYTES_INTERP = 18.3 + 64.0             # JRA55 window amortised + 8 exchange fields written
BYTES_NET = 88.0 + 40.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=1440)
    ap.add_argument("--ny", type=int, default=560)
    ap.add_argument("--halo", type=int, default=7)  # README.md:58 halo=(7,7,7)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--halo-backend", choices=("rccl", "torch"), default="rccl")
    ap.add_argument("--flux-configuration", choices=("default", "corrected", "ncar"), default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-repeats", type=int, default=0)  # 0: as many passes as fit ≈ 6 s (3…30)
    return ap.parse_args()


def cpu_baseline(case_np, params, nx, ny, h, repeats):
    """The CPU oracle (C restatement, OpenMP over rows) timed on this box's host cores on the same
    workload: full passes of interpolate + solver + net fluxes over the rank-0 slab.  `repeats` = 0 sizes the
    sample itself: as many passes as fit ≈ 6 s of wall time (at least 3, at most 30), best pass reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    g = orc.make_grid(nx, ny, h, h, 1)
    cores = orc.max_threads()

    def one_pass():
        t0 = time.perf_counter()
        atmos = orc.interpolate_atmosphere_state(g, case_np["src"], case_np["weights"], 0, 1, 0.37)
        fl = orc.compute_atmosphere_ocean_fluxes(g, params, case_np["ocean"], atmos, nthreads=0, scales=False)
        orc.compute_net_ocean_fluxes(g, params, case_np["ocean"], atmos, fl, weights=case_np["weights"])
        return time.perf_counter() - t0

    t_best = one_pass()
    if repeats <= 0:
        repeats = min(30, max(3, int(6.0 / max(t_best, 1e-3))))
    for _ in range(repeats - 1):
        t_best = min(t_best, one_pass())
    return dict(value=nx * ny / t_best, unit="cells/s", cores=cores, kind="port",
                sample=f"{repeats} full update_state passes over the {nx}x{ny} surface (best of {repeats}); "
                       "oracle/coflux_oracle.c, OpenMP over rows")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the flux path has no CPU backend")
    torch.cuda.set_device(local_rank)

    h = a.halo
    if a.scaling == "weak":
        ny_global, (j0, j1) = a.ny * world, (rank * a.ny, (rank + 1) * a.ny)
    else:
        ny_global = a.ny
        j0, j1 = slab_bounds(a.ny, rank, world)
    nx, ny = a.nx, j1 - j0

    fluxes_cfg = {"default": ic.SimilarityTheoryFluxes, "corrected": ic.corrected_atmosphere_ocean_fluxes,
                  "ncar": ic.ncar_atmosphere_ocean_fluxes}[a.flux_configuration]()
    params = ic.flux_params(fluxes_cfg, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))

    # ---- synthetic inputs, resident in HBM before the timed region ---------------------------------
    ocean_np = syn.ocean_state(nx, ny, h, h, ny_global=ny_global, j_offset=j0)
    src_np = syn.jra55_snapshots(2)
    fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h, ny_global=ny_global, j_offset=j0)
    w_np = dict(separable=True, fi=fi, fj=fj, latitude=phi)

    ctx = FluxContext(nx, ny, h, h, params, ring=1, device=local_rank)
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES)
    fl = ctx.field_set(FLUX_NAMES)
    net = ctx.field_set(NET_NAMES)
    try:
        halo = SlabHaloExchanger(ctx, ny, h, rows=1, backend=a.halo_backend)
    except Exception as exc:  # native RCCL init failed: same exchange through torch.distributed (also RCCL)
        print(f"[bench] native RCCL halo path unavailable ({exc}); using torch.distributed P2P", file=sys.stderr)
        halo = SlabHaloExchanger(ctx, ny, h, rows=1, backend="torch")
    halo_fields = [ocean[k] for k in ("T", "S", "u", "v")]

    # Prove the exchange on this machine before timing it: the synthetic state is a function of the GLOBAL cell
    # index, so every rank knows what its neighbours' boundary rows must be.  Wipe the halo rows, exchange, compare.
    halo_verified = None
    if world > 1:
        def check(exchanger):
            rows = [r for r, has in ((h - 1, rank > 0), (h + ny, rank < world - 1)) if has]
            want = [[f[r].clone() for r in rows] for f in halo_fields]
            for f in halo_fields:
                for r in rows:
                    f[r].fill_(float("nan"))
            exchanger(halo_fields)
            ctx.sync()
            torch.cuda.synchronize()
            good = all(torch.equal(f[r], w) for f, ws in zip(halo_fields, want) for r, w in zip(rows, ws))
            for f, ws in zip(halo_fields, want):      # restore either way
                for r, w in zip(rows, ws):
                    f[r].copy_(w)
            flag = torch.tensor([1 if good else 0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())
        halo_verified = check(halo)
        if not halo_verified and halo.backend == "rccl":
            print("[bench] native RCCL halo rows did not match the neighbours' rows; using torch.distributed P2P", file=sys.stderr)
            halo = SlabHaloExchanger(ctx, ny, h, rows=1, backend="torch")
            halo_verified = check(halo)
        if not halo_verified:
            raise SystemExit("bench.py: halo exchange does not reproduce the neighbours' boundary rows")

    def step():
        halo(halo_fields)
        ctx.update_state(src, w, ocean, atmos, fl, net, time_fraction=0.37)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    # per-kernel HIP events on the launch stream inside the timed region; every event record between two kernels
    # costs ≈ 4 µs of stream time (124 vs 143 µs per step measured with and without them), so only every
    # `stride`-th step is bracketed: ≈ 25 sampled launches of each kernel
    stride = max(1, a.steps // 25)
    ctx.set_option(abi.OPT_PROFILE_STRIDE, stride)
    ctx.profile_enable((a.steps + stride - 1) // stride)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cells_total = nx * (ny_global if a.scaling == "weak" else a.ny) if world > 1 else nx * ny
    cells_rank = (nx + 2) * (ny + 2)  # launch covers the ring as the reference does
    value = cells_total * a.steps / elapsed

    if rank == 0:
        interp_ms, nrec = ctx.profile_read(0)   # per-kernel HIP-event averages over the timed region
        ao_ms, _ = ctx.profile_read(1)
        net_ms, _ = ctx.profile_read(2)
        copy_bytes = 256 << 20
        copy_ms = ctx.time_copy(copy_bytes, 20)

        # HBM bytes per launch from the committed rocprofv3 PMC passes (separate runs, see profiles/);
        # only valid for the workload they were collected on
        traffic = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            if (nx, ny, a.flux_configuration, world) == (1440, 560, "default", 1):
                traffic = {k.split("<")[0]: v["hbm_bytes_per_launch"] for k, v in pmc.items()}
        except Exception:
            pass

        # Second view of the dominant kernel: it is FP64-VALU-issue bound, not HBM bound.  VALU
        # wave-instructions per launch come from the committed PMC pass (profiles/r01_pmc_ao.json, same workload);
        # peak = one VALU instruction per SIMD per 4 cycles × 1024 SIMDs at the 2.4 GHz engine clock.
        valu = None
        try:
            kname = {"default": "ao_flux_fast_kernel<false, 0>", "corrected": "ao_flux_fast_kernel<true, 0>"}.get(a.flux_configuration)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_ao.json")))["kernels"]
            if kname in pm and (nx, ny, world) == (1440, 560, 1):
                n_inst = pm[kname]["SQ_INSTS_VALU"]
                peak = 1024 * 2.4e9 / 4
                ach = n_inst / (ao_ms * 1e-3)
                valu = dict(bound="fp64-valu-issue", kernel="ao_flux_fast_kernel", achieved=ach / 1e9, peak=peak / 1e9,
                            unit="G wave-instructions/s", frac=ach / peak, valu_instructions_per_launch=n_inst,
                            lane_utilisation=pm[kname]["SQ_THREAD_CYCLES_VALU"] / (64.0 * pm[kname]["SQ_ACTIVE_INST_VALU"]),
                            source="profiles/r01_pmc_ao.json (rocprofv3 --pmc SQ_INSTS_VALU)")
        except Exception:
            pass

        def roof(name, nbytes, ncells, ms):
            achieved = nbytes * ncells / (ms * 1e-3) / 1e9
            return dict(bound="hbm", kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic.get(name.split(" ")[0]),
                        bytes_per_cell=nbytes, cells_per_launch=ncells,
                        avg_launch_ms=ms, launches_timed=nrec, cells_per_s=ncells / (ms * 1e-3))

        out = dict(metric="flux-kernel surface cells/s (update_state!: JRA55 interp + similarity-theory fluxes + net fluxes)",
                   value=value, unit="cells/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=elapsed / a.steps * 1e3, higher_is_better=True, scaling=a.scaling,
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=f"1/4-degree LatitudeLongitudeGrid surface {nx}x{a.ny} per "
                                        f"{'GPU' if a.scaling == 'weak' else 'job'}, JRA55 640x320 f32 atmosphere, "
                                        f"SimilarityTheoryFluxes(:{a.flux_configuration}) + Radiation, halo {h}, ring 1",
                               global_cells=cells_total, parallelism=f"latitude-slab x{world}",
                               halo_backend=halo.backend, halo_verified=halo_verified),
                   # dominant kernel = compute_atmosphere_ocean_fluxes! (SURVEY.md §8d contract figure 128 B/cell)
                   roofline=roof("ao_flux_fast_kernel (compute_atmosphere_ocean_fluxes!)", BYTES_AO, cells_rank, ao_ms),
                   roofline_interpolate=roof("interpolate_kernel (interpolate_atmosphere_state!)", BYTES_INTERP, cells_rank, interp_ms),
                   roofline_net_fluxes=roof("net_flux_kernel (compute_net_ocean_fluxes!)", BYTES_NET, nx * ny, net_ms),
                   roofline_fp64_valu=valu,
                   stages_ms=dict(interpolate=interp_ms, ao_fluxes=ao_ms, net_fluxes=net_ms),
                   device_copy_GBs=2 * copy_bytes / (copy_ms * 1e-3) / 1e9,
                   parity="vs reference: unpinned (self-consistent restatements only; see DESIGN.md)")
        if not a.no_cpu_baseline:
            case_np = dict(ocean=ocean_np, src=src_np, weights=w_np)
            out["cpu_baseline"] = cpu_baseline(case_np, params, nx, ny, h, a.cpu_repeats)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
