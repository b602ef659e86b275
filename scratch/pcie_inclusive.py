#!/usr/bin/env python
"""PCIe-inclusive rate of the stepping loop: the JRA55 snapshots arrive from HOST memory while the model steps.

bench.py's `value` is quoted with the 4-snapshot JRA55 window resident in HBM (the bench contract).  The one place the
C ABI takes host buffers is that window (cf_window_upload / cf_window_commit, include/coflux.h): a coupled run hands
over one new 3-hourly snapshot (nine 640×320 Float32 planes = 7.37 MB) every nine 20-minute steps.  This script
times the same 1440×560 step with that traffic inside the timed region:

  * `resident`  : the host-driven loop on a window that already holds every snapshot it needs (no upload);
  * `pageable`  : cf_window_upload from ordinary NumPy arrays (a host memcpy into the pinned staging buffer on the
                  launching thread, then the asynchronous copy on the window's copy stream);
  * `in_place`  : the reader fills the pinned staging buffer itself beforehand (cf_window_host_buffer) and only
                  cf_window_commit runs inside the loop — what jra55.py's provider does with memory-mapped files;
  * `upload_only`: the uploads alone, back to back (GB/s over the link).

    python scratch/pcie_inclusive.py [--steps 900]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from coflux import abi, synthetic as syn  # noqa: E402
from coflux import interface_computations as ic  # noqa: E402
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext, SnapshotWindow  # noqa: E402

SNAPSHOT_INTERVAL, DT = 3 * 3600.0, 20 * 60.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--slots", type=int, default=4)
    a = ap.parse_args()
    nx, ny, h = 1440, 560, 7
    params = ic.flux_params(ic.SimilarityTheoryFluxes(), ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    ctx = FluxContext(nx, ny, h, h, params, ring=1, device=0)
    ctx.set_option(abi.OPT_MERGED_PREFETCH, 2)
    o0 = syn.ocean_state(nx, ny, h, h)
    o1 = syn.evolved_ocean_state(o0, nx, ny, h, h, 1)
    states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in (o0, o1)]
    states[1]["mask"] = states[0]["mask"]
    fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2)]
    fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES[:5])

    n_host = a.slots   # distinct host snapshots, cycled (snapshot counter t ↦ host record t mod n_host = its slot)
    host = syn.jra55_snapshots(n_host, temporal_correlation=0.95)
    snap = [{v: np.ascontiguousarray(host[v][n]) for v in abi.JRA55_VARIABLES} for n in range(n_host)]
    snap_bytes = sum(x.nbytes for x in snap[0].values())
    win = SnapshotWindow(ctx, syn.JRA55_NX, syn.JRA55_NY, a.slots)
    inc = DT / SNAPSHOT_INTERVAL

    def prime(t0):
        for t in range(t0, t0 + a.slots - 1):
            win.upload(t, snap[t % n_host])
        ctx.sync()

    resident = {v: ctx.to_device(np.ascontiguousarray(host[v][:a.slots])) for v in abi.JRA55_VARIABLES}

    def loop(first, n, mode):
        """n steps from step `first`; snapshot counter of step s is ⌊s·inc⌋; the window is kept a.slots − 1 ahead."""
        have = int(first * inc) + a.slots - 2   # newest snapshot counter already in the window
        if mode == "resident":   # bench.py's arrangement: every level in HBM, memory index = counter mod n_levels
            def source(c, f):
                return dict(src=resident, level1=c % a.slots, level2=(c + 1) % a.slots, time_fraction=f)
        else:
            def source(c, f):
                return dict(src=win.source(c, c + 1, f))
        c0 = int(first * inc)
        ctx.interpolate_atmosphere_state(weights=w, atmos=sets[first % 2], **source(c0, first * inc - c0))
        for s in range(first, first + n):
            n1 = int(s * inc)
            # a new interval has begun: the oldest slot is free once the interpolations queued on it have run
            while mode != "resident" and have < n1 + a.slots - 2:
                have += 1
                if mode == "pageable":
                    win.upload(have, snap[have % n_host])
                else:   # in_place: the staging buffer already holds the record (filled before the timer started)
                    win.commit(have % a.slots, have)
            nxt = (s + 1) * inc
            ctx.prefetch_atmosphere_state(weights=w, atmos_next=sets[(s + 1) % 2], **source(int(nxt), nxt - int(nxt)))
            ctx.update_state(weights=w, ocean=states[s % 2], atmos=sets[s % 2], fluxes=fl, net=net, **source(n1, s * inc - n1))

    out = dict(steps=a.steps, slots=a.slots, snapshot_MB=snap_bytes / 1e6, snapshots_uploaded=int(a.steps * inc))
    for mode in ("resident", "pageable", "in_place"):
        n = a.steps
        if mode == "in_place":   # n_host must map onto the slots consistently: record t sits in staging buffer t mod slots
            for slot in range(a.slots):
                win.wait_slot(slot)
                for k, v in enumerate(abi.JRA55_VARIABLES):
                    win.host_view(slot, k)[...] = snap[slot % n_host][v]
        best = None
        for _ in range(3):
            prime(0)
            loop(0, min(n, 50), mode if mode != "in_place" else "pageable")   # warm
            prime(0)
            if mode == "in_place":
                for slot in range(a.slots):
                    win.wait_slot(slot)
                    for k, v in enumerate(abi.JRA55_VARIABLES):
                        win.host_view(slot, k)[...] = snap[slot % n_host][v]
                for t in range(a.slots - 1):
                    win.commit(t % a.slots, t)
            ctx.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(0, n, mode)
            ctx.sync()
            dt = (time.perf_counter() - t0) / n
            best = dt if best is None else min(best, dt)
        out[mode] = dict(ms_per_step=best * 1e3, cells_per_s=nx * ny / best, steps_timed=n)
    # the uploads alone
    ctx.sync()
    t0 = time.perf_counter()
    m = 40
    for t in range(1000, 1000 + m):
        win.upload(t, snap[t % n_host])
    ctx.sync()
    for slot in range(a.slots):
        win.wait_slot(slot)
    dt = time.perf_counter() - t0
    out["upload_only"] = dict(ms_per_snapshot=dt / m * 1e3, GB_per_s=snap_bytes * m / dt / 1e9)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
