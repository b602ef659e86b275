"""JRA55 snapshot window in HBM (cf_window_*): the sliding-window backend of
JRA55PrescribedAtmosphere must give exactly what the all-in-memory backend gives."""
import time

import numpy as np
import pytest
import torch

import oracle as orc
import util
from coflux import abi
from coflux import interface_computations as ic
from coflux import models as cm
from coflux import synthetic as syn
from coflux.runtime import EXCHANGE_NAMES, CofluxError, FluxContext, SnapshotWindow

pytestmark = pytest.mark.gpu


def test_window_source_matches_oracle_and_reuses_slots():
    case = util.build_case(90, 40, n_levels=7)
    snaps = case["src"]
    ctx = FluxContext(90, 40, 3, 3, ic.flux_params())
    w = {k: (ctx.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in case["weights"].items()}
    win = SnapshotWindow(ctx, syn.JRA55_NX, syn.JRA55_NY, 3)
    g = orc.make_grid(90, 40, 3, 3, 1)
    atmos = ctx.field_set(EXCHANGE_NAMES)
    win.upload(0, {k: v[0] for k, v in snaps.items()})
    for n in range(6):                                   # slots 0,1,2 are reused twice
        win.upload(n + 1, {k: v[n + 1] for k, v in snaps.items()})
        assert win.find(n) == n % 3 and win.find(n + 1) == (n + 1) % 3
        src = win.source(n, n + 1, 0.25 + 0.1 * n)
        assert (src.level1, src.level2, src.n_levels) == (n % 3, (n + 1) % 3, 3)
        ctx.interpolate_atmosphere_state(src, w, atmos)   # queued behind the uploads by events, no host sync
        ref = orc.interpolate_atmosphere_state(g, snaps, case["weights"], n, n + 1, 0.25 + 0.1 * n)
        for k in EXCHANGE_NAMES:
            e = util.rel_err(util.window(atmos[k].cpu().numpy(), 3, 3, 90, 40, 1), util.window(ref[k], 3, 3, 90, 40, 1),
                             util.ATMOS_SCALE[k])
            assert e <= 1e-12, (n, k, e)
    assert win.find(2) == -1                              # overwritten by snapshot 5
    with pytest.raises(CofluxError, match="not in the window"):
        win.source(2, 3, 0.5)
    with pytest.raises(CofluxError, match="time fraction"):
        win.source(5, 6, 1.5)
    # staging buffers are host memory the reader fills in place
    win.wait_slot(1)
    view = win.host_view(1, "tas")
    assert view.shape == (syn.JRA55_NY, syn.JRA55_NX) and view.dtype == np.float32
    view[...] = 300.0
    for v in abi.JRA55_VARIABLES[1:]:
        win.host_view(1, v)[...] = 0.0
    win.commit(1, 7)
    ctx.interpolate_atmosphere_state(win.source(7, 7, 0.0), w, atmos)
    assert np.max(np.abs(util.window(atmos["T"].cpu().numpy(), 3, 3, 90, 40, 1) - 300.0)) < 1e-10
    # the compute stream waits for an upload ONCE (the first descriptor after the commit); a second commit of the same
    # slot must be waited for again, however many descriptors of the old content were handed out in between
    for value in (310.0, 320.0, 330.0):
        for _ in range(3):
            ctx.interpolate_atmosphere_state(win.source(7, 7, 0.0), w, atmos)
        win.wait_slot(1)
        view[...] = value
        win.commit(1, 7)
        ctx.interpolate_atmosphere_state(win.source(7, 7, 0.0), w, atmos)
        assert np.max(np.abs(util.window(atmos["T"].cpu().numpy(), 3, 3, 90, 40, 1) - value)) < 1e-10
    win.close()
    ctx.close()


@pytest.mark.parametrize("prefetch", [False, True])
def test_sliding_window_atmosphere_equals_in_memory_atmosphere(prefetch):
    """README recipe with a provider-backed window of 3 snapshots vs all 6 snapshots resident: every step's
    boundary conditions are bitwise identical (same kernels, same data, different residency)."""
    nx, ny, nz, h = 90, 40, 10, 3
    snaps = syn.jra55_snapshots(7)          # 7 is no multiple of the 3 slots: the wrap 6 → 0 must not collide
    state = syn.ocean_state(nx, ny, h, h)
    reads = []

    def provider(n):
        reads.append(n)
        time.sleep(0.002)                                  # a slow file system
        return {k: v[n] for k, v in snaps.items()}

    def build(atmosphere):
        grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h), latitude=(-70, 70), z=(-3000, 0))
        ocean = cm.ocean_simulation(grid)
        cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
        return ocean, cm.OceanSeaIceModel(ocean, atmosphere=atmosphere)

    ocean_a, model_a = build(cm.JRA55PrescribedAtmosphere(snaps))
    windowed = cm.JRA55PrescribedAtmosphere(provider=provider, total_snapshots=7, time_indices_in_memory=3,
                                            prefetch=prefetch)
    ocean_b, model_b = build(windowed)
    for _ in range(72):                                    # 72 × 20 min = 24 h: 8 snapshot intervals → wraps the repeat "year"
        cm.time_step(model_a, 20 * cm.minutes)
        cm.time_step(model_b, 20 * cm.minutes)
        for name in ("u", "v", "T", "S"):
            a = getattr(ocean_a.model.top_boundary_conditions, name)
            b = getattr(ocean_b.model.top_boundary_conditions, name)
            assert torch.equal(a, b), (model_a.clock.iteration, name)
    assert set(reads) == set(range(7)) and len(reads) >= 9   # wrapped: 0 and 1 were read again
    if prefetch:
        assert windowed._reader is not None
    windowed.close()
