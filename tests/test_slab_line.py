"""CF_OPT_LATENCY_LAYOUT (include/coflux.h; csrc/coflux_solver_slab.hip, csrc/tools/gcn_sched.py): the exact path of
compute_atmosphere_ocean_fluxes! (omip_simulation.jl:40-49) on launches that leave a SIMD with one or two waves — a
latitude slab of launch.sh:165 / pbs_launch.sh:51's Partition(1, N) — runs kernels whose iteration is laid out in big basic
blocks and re-scheduled for latency AFTER register allocation.  Same instructions on the same operand values: every output
must be the same BITS as the production kernels', on every flux preset, with and without the fused net fluxes, with tail
workgroups, and at full size where the layout is forced."""
import numpy as np
import pytest
import torch

import util
from coflux import abi
from coflux import interface_computations as ic
from coflux import synthetic as syn
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
from test_gpu_parity import run_gpu, run_oracle, compare

pytestmark = pytest.mark.gpu

NEVER, AUTO, ALWAYS = 0, 1, 2


def _same_bits(a, b, label):
    for group in ("atmos", "fluxes", "net"):
        for k in a[group]:
            assert np.array_equal(a[group][k].view(np.uint8), b[group][k].view(np.uint8)), (label, group, k)


@pytest.mark.parametrize("config", ["default", "corrected", "corrected_wind"])
@pytest.mark.parametrize("fused", [True, False])
def test_eighth_slab_same_bits_as_the_production_kernels(config, fused):
    fluxes, vd = util.CONFIGS[config]()
    params = ic.flux_params(fluxes, velocity_difference=vd, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    case = util.build_case(1440, 70, 7, 7, ny_global=560, j_offset=140)
    ref = run_gpu(case, params, fused=fused, ice=True, options=((abi.OPT_LATENCY_LAYOUT, NEVER),))
    got = run_gpu(case, params, fused=fused, ice=True, options=((abi.OPT_LATENCY_LAYOUT, ALWAYS),))
    _same_bits(got, ref, f"{config} fused={fused}")
    compare(case, got, run_oracle(case, params, ice=True), 1)


def test_automatic_mode_follows_the_chunk_plan():
    """1440×70 (394 workgroups on 256 CUs) takes the layout, the full surface (three waves per SIMD) does not; a parameter set
    with β_gust = 0 never does; CF_OPT_SOLVER_PATH = certified keeps its own kernels."""
    def layout(nx, ny, params, options=()):
        ctx = FluxContext(nx, ny, 7, 7, params)
        for o, v in options:
            ctx.set_option(o, v)
        mask = ctx.to_device(syn.ocean_state(nx, ny, 7, 7)["mask"])
        ctx.ensure_chunk_table(mask)
        out = ctx.solver_latency_layout()
        ctx.close()
        return out
    P = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes())
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert layout(1440, 70, P) is (394 <= 2 * cus)
    assert layout(1440, 560, P) is False
    assert layout(1440, 560, P, ((abi.OPT_LATENCY_LAYOUT, ALWAYS),)) is True
    assert layout(1440, 70, P, ((abi.OPT_LATENCY_LAYOUT, NEVER),)) is False
    assert layout(1440, 70, P, ((abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED),)) is False
    # the plain logarithmic profile: only when forced (measured: no gain)
    assert layout(1440, 70, ic.flux_params()) is False
    assert layout(1440, 70, ic.flux_params(), ((abi.OPT_LATENCY_LAYOUT, ALWAYS),)) is True
    calm = ic.corrected_atmosphere_ocean_fluxes()
    calm.gustiness_parameter = 0.0
    assert layout(1440, 70, ic.flux_params(calm), ((abi.OPT_LATENCY_LAYOUT, ALWAYS),)) is False


@pytest.mark.parametrize("config", ["default", "corrected"])
def test_forced_on_the_quarter_degree_surface_same_bits_and_trip_counts(config):
    fluxes, vd = util.CONFIGS[config]()
    params = ic.flux_params(fluxes, velocity_difference=vd, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    case = util.build_case(1440, 560, 7, 7)
    ref = run_gpu(case, params, fused=True, ice=True, options=((abi.OPT_LATENCY_LAYOUT, NEVER),))
    got = run_gpu(case, params, fused=True, ice=True, options=((abi.OPT_LATENCY_LAYOUT, ALWAYS),))
    _same_bits(got, ref, config)


@pytest.mark.parametrize("stop", ["fixed0", "fixed1", "fixed5", "maxiter3"])
def test_stop_rules_at_the_edges(stop):
    """FixedIterations(0) (no trip: the 1e-4 first guess is returned), one trip (only the peeled one), a cap below convergence."""
    fluxes = ic.SimilarityTheoryFluxes()
    if stop.startswith("fixed"):
        fluxes.solver_stop_criteria = ic.FixedIterations(int(stop[5:]))
    else:
        fluxes.solver_stop_criteria = ic.ConvergenceStopCriteria(tolerance=1e-8, maxiter=3)
    params = ic.flux_params(fluxes)
    case = util.build_case(360, 48, 4, 4)
    ref = run_gpu(case, params, fused=True, options=((abi.OPT_LATENCY_LAYOUT, NEVER),))
    got = run_gpu(case, params, fused=True, options=((abi.OPT_LATENCY_LAYOUT, ALWAYS),))
    _same_bits(got, ref, stop)


@pytest.mark.parametrize("layout,config", [(AUTO, "corrected"), (ALWAYS, "default"), (ALWAYS, "corrected")])
def test_time_steps_with_tail_workgroups_same_bits(layout, config):
    """cf_time_steps with the next step's interpolation in the solver launch's tail workgroups (the bench's schedule)."""
    nx, ny, H, inc, n = 192, 48, 4, 1.0 / 9.0, 21
    outs = []
    for mode in (NEVER, layout):
        fluxes, vd = util.CONFIGS[config]()
        ctx = FluxContext(nx, ny, H, H, ic.flux_params(fluxes, velocity_difference=vd), ring=1)
        ctx.set_option(abi.OPT_LATENCY_LAYOUT, mode)
        ctx.set_option(abi.OPT_MERGED_PREFETCH, 2)
        o0 = syn.ocean_state(nx, ny, H, H)
        o1 = syn.evolved_ocean_state(o0, nx, ny, H, H, 1)
        states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in (o0, o1)]
        states[1]["mask"] = states[0]["mask"]
        src = {k: ctx.to_device(v) for k, v in syn.jra55_snapshots(4).items()}
        fi, fj, phi = syn.latlon_fractional_indices(nx, ny, H, H)
        w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
        sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2)]
        fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
        sched = ctx.make_schedule(states, sets, first_level=0, time_fraction=0.0, time_fraction_increment=inc, pipeline=True)
        ctx.time_steps(0, n, sched, src, w, fl, net)
        ctx.sync()
        if mode != NEVER:
            assert ctx.solver_latency_layout()
        outs.append({**{k: v.cpu() for k, v in fl.items()}, **{"net." + k: v.cpu() for k, v in net.items()},
                     **{"a." + k: v.cpu() for k, v in sets[(n - 1) % 2].items()}})
        ctx.close()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("ny,config", [(70, "default"), (140, "default"), (140, "corrected")])
def test_piece_layer_plan_same_bits_as_uniform_chunks(ny, config):
    """plan_chunk_rounds cuts the LAST dispatch layer of a surface of one to three 256-cell chunks per CU into pieces of 64 /
    128 / 192 wet cells (one piece per CU at most: a wave there shares its SIMD with older ones and runs its chain 1.4 × slower).
    A plan only steers scheduling: forced uniform 256-cell chunks (CF_OPT_AO_CHUNK) must give the same bits."""
    fluxes, vd = util.CONFIGS[config]()
    params = ic.flux_params(fluxes, velocity_difference=vd, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    case = util.build_case(1440, ny, 7, 7, ny_global=560, j_offset=210)
    auto = run_gpu(case, params, fused=True, ice=True)
    uniform = run_gpu(case, params, fused=True, ice=True, options=((abi.OPT_AO_CHUNK, 256),))
    _same_bits(auto, uniform, f"1440x{ny} {config}")
