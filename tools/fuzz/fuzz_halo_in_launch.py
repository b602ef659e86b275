"""GPU fuzz, round 6: CF_OPT_HALO_IN_SOLVER_LAUNCH against the exchange kernel — bitwise — on random slab decompositions (2–5 slabs
as contexts of one process, each on its own stream, mailboxes handed over in-process), sizes down to three rows per slab (boundary
chunk sets that overlap), flux configurations, step counts, with and without land, halo rows poisoned before every run.
(Two or three slabs only: the contexts of ONE process share the runtime's four hardware queues, and an exchange that waits for a
neighbour whose kernels sit behind it in the same queue times out — a limit of this harness, with either form of the exchange; one
process per rank, as on a node and in tests/test_halo_in_launch.py, has no such coupling: 4 ranks pass there.)
usage: fuzz_halo_in_launch.py [seed] [cases]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ.setdefault("COFLUX_EXPERIMENTS", "1")
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.distributed import slab_bounds
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
FLAGS = tuple(int(x) for x in os.environ.get("FUZZ_FLAGS", "0,1").split(","))   # (0,0: the exchange kernel against itself)
for n in range(ncases):
    world = int(os.environ.get("FUZZ_WORLD", 0)) or int(rng.integers(2, 4)); h = int(rng.integers(2, 6))
    nx = int(rng.choice([24, 96, 360, 1440, int(rng.integers(8, 700))]))
    rows = [int(rng.choice([3, 4, 9, 35, 70, int(rng.integers(3, 120))])) for _ in range(world)]
    NY = sum(rows); bounds = np.concatenate([[0], np.cumsum(rows)])
    nsteps = int(rng.integers(2, 9)); land = bool(rng.integers(0, 2)); chunk = int(rng.choice([0, 0, 256, 512]))
    cfg = [ic.SimilarityTheoryFluxes, ic.corrected_atmosphere_ocean_fluxes][int(rng.integers(0, 2))]
    desc = dict(world=world, nx=nx, rows=rows, h=h, nsteps=nsteps, land=land, chunk=chunk, cfg=cfg.__name__)
    try:
        P = ic.flux_params(cfg(), ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
        slabs = []
        for r in range(world):
            j0, j1 = int(bounds[r]), int(bounds[r + 1]); ny = j1 - j0
            ctx = FluxContext(nx, ny, h, h, P, ring=1)
            ctx._check(ctx.lib.cf_set_stream(ctx._h, None), "cf_set_stream")      # its own stream: the slabs' kernels must overlap
            if chunk: ctx.set_option(abi.OPT_AO_CHUNK, chunk)
            ctx.set_option(abi.OPT_MERGED_PREFETCH, 2)
            o0 = syn.ocean_state(nx, ny, h, h, ny_global=NY, j_offset=j0, land_fraction=land)
            o1 = syn.evolved_ocean_state(o0, nx, ny, h, h, 1, ny_global=NY, j_offset=j0)
            states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in (o0, o1)]
            states[1]["mask"] = states[0]["mask"]
            fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h, ny_global=NY, j_offset=j0)
            w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
            src = {k: ctx.to_device(v) for k, v in syn.jra55_snapshots(3).items()}
            slabs.append(dict(ctx=ctx, states=states, w=w, src=src, ny=ny, sets=[ctx.field_set(EXCHANGE_NAMES) for _ in range(2)]))
        torch.cuda.synchronize()
        handles = [s["ctx"].peer_halo_export(4, 2) for s in slabs]
        for r, s in enumerate(slabs):
            s["ctx"].peer_halo_connect(handles[r - 1] if r > 0 else None, handles[r + 1] if r < world - 1 else None, r, world)
        runs = []
        for flag in FLAGS:
            outs = []
            for r, s in enumerate(slabs):
                s["ctx"].set_option(abi.OPT_HALO_IN_SOLVER_LAUNCH, flag)
                for st in s["states"]:
                    for k in ("T", "S", "u", "v"):
                        if r > 0: st[k][:h] = float("nan")
                        if r < world - 1: st[k][h + s["ny"]:] = float("nan")
                s["fl"], s["net"] = s["ctx"].field_set(FLUX_NAMES), s["ctx"].field_set(NET_NAMES)
                s["sched"] = s["ctx"].make_schedule(s["states"], s["sets"], time_fraction_increment=1 / 9, pipeline=True, halo_backend=abi.HALO_PEER, halo_rows=2)
                s["before"] = s["ctx"].peer_halo_stats()
            torch.cuda.synchronize()
            for s in slabs:   # every slab's steps are queued before any is waited for
                s["ctx"].time_steps(0, nsteps, s["sched"], s["src"], s["w"], s["fl"], s["net"])
            for s in slabs:
                s["ctx"].sync()
                st = s["ctx"].peer_halo_stats()
                outs.append(([s["fl"][k].clone() for k in FLUX_NAMES] + [s["net"][k].clone() for k in ("u", "v", "T", "S")], st[1] - s["before"][1]))
            runs.append(outs)
        for r in range(world):
            ny = slabs[r]["ny"]
            assert runs[0][r][1] == (nsteps - 1) * FLAGS[0] and runs[1][r][1] == (nsteps - 1) * FLAGS[1], ("riders used", runs[0][r][1], runs[1][r][1])
            for a, b in zip(runs[0][r][0], runs[1][r][0]):
                assert torch.equal(a, b) or (torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))), ("bits", r)
            lo = h if r == 0 else h - 1
            for a in runs[1][r][0][:5]:
                assert bool(torch.isfinite(a[lo:h + ny + (0 if r == world - 1 else 1), h - 1:h + nx + 1]).all()), ("a halo row was not delivered", r)
        for s in slabs: s["ctx"].close()
    except Exception as exc:
        bad += 1
        print("FAIL", n, desc, repr(exc)[:300], flush=True)
        for s in slabs:
            try: s["ctx"].close()
            except Exception: pass
print(f"{ncases - bad} of {ncases} cases passed", flush=True)
