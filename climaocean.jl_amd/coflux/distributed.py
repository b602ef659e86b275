"""Latitude-slab sharding of the surface grid (the reference's own production
shapes are Partition(1,4) launch.sh:165 and Partition(1,8) pbs_launch.sh:51: contiguous j-slabs,
one rank per GPU).

The flux kernels need one halo row of the ocean surface state from each neighbour
(v[j+1] north; the ring row south/north when ring = 1).  On GPUs the rows travel over
RCCL/xGMI through libcoflux's cf_halo_exchange_rows; the torch.distributed point-to-point form
below is the same exchange for CPU tensors (gloo) and is what the world_size-2 CPU tests run.
"""
import torch
import torch.distributed as dist


def slab_bounds(ny_global, rank, world_size):
    """Rows [j0, j1) owned by `rank`: as equal as possible, remainder to the southern ranks."""
    base, rem = divmod(ny_global, world_size)
    j0 = rank * base + min(rank, rem)
    return j0, j0 + base + (1 if rank < rem else 0)


def halo_row_slices(ny, hy, rows):
    """(send_south, recv_south, send_north, recv_north) row slices of a (ny+2hy, ·) slab array."""
    return (slice(hy, hy + rows), slice(hy - rows, hy),
            slice(hy + ny - rows, hy + ny), slice(hy + ny, hy + ny + rows))


def exchange_halo_rows_torch(tensors, ny, hy, rows=1, group=None):
    """Neighbour exchange of `rows` boundary rows for every tensor in `tensors` using
    torch.distributed P2P (gloo on CPU, RCCL on GPU).  Non-periodic in j: the first and last
    ranks keep their outer halos."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return
    ss, rs, sn, rn = halo_row_slices(ny, hy, rows)
    ops, recvs = [], []
    for t in tensors:
        if rank > 0:
            buf = torch.empty_like(t[rs])
            ops.append(dist.P2POp(dist.isend, t[ss].contiguous(), rank - 1, group))
            ops.append(dist.P2POp(dist.irecv, buf, rank - 1, group))
            recvs.append((t, rs, buf))
        if rank < world - 1:
            buf = torch.empty_like(t[rn])
            ops.append(dist.P2POp(dist.isend, t[sn].contiguous(), rank + 1, group))
            ops.append(dist.P2POp(dist.irecv, buf, rank + 1, group))
            recvs.append((t, rn, buf))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for t, sl, buf in recvs:
        t[sl].copy_(buf)


def fold_north_halo_torch(t, nx, ny, hx, hy, rows, location="center", sign=1.0):
    """Tripolar fold of one (ny+2hy, nx+2hx) array held by the LAST slab (include/coflux.h: cf_fold_north_halo;
    Oceananigans' zipper boundary condition): north halo rows from the slab's own mirrored interior rows.
    The torch form of the fold for CPU slabs (gloo tests); libcoflux's kernel is the GPU form."""
    i = torch.arange(-hx, nx + hx) % nx
    for r in range(1, rows + 1):
        if location == "x_face":
            src_i, src_j, sg = (nx - i) % nx, ny - 1 - r, torch.where(i == 0, abs(sign), sign).to(t.dtype)
        elif location == "y_face":
            src_i, src_j, sg = nx - 1 - i, ny - r, sign
        else:
            src_i, src_j, sg = nx - 1 - i, ny - 1 - r, sign
        t[hy + ny - 1 + r, :] = sg * t[hy + src_j, hx + src_i]


class SlabHaloExchanger:
    """Per-step halo exchange of the ocean surface fields of one slab.

    backend "rccl": libcoflux's grouped ncclSend/ncclRecv on its communication stream (the unique id is
    created on rank 0 and broadcast with torch.distributed); "peer": libcoflux's peer-direct mailboxes (HIP IPC
    handles gathered with torch.distributed); "torch": torch.distributed P2P.
    `rows` defaults to ring + 1: with ring = 1 the kernels also compute row j = ny, whose cell-centre v reads the
    y-face at j = ny + 1.
    """

    def __init__(self, ctx, ny, hy, rows=None, backend="rccl"):
        self.ctx, self.ny, self.hy = ctx, ny, hy
        self.rows = rows if rows is not None else ctx.grid.ring + 1
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.backend = backend if self.world > 1 else "none"
        if self.backend == "peer":
            # cf_peer_halo_export fails (CF_ERR_COMM) on a rank whose device cannot give it fine-grained memory.  Every
            # rank reaches the collectives below whatever happened locally: the ranks agree on the outcome first, and a
            # failure anywhere moves ALL of them to the RCCL exchange together (a rank that raised before the gather
            # would leave the others blocked in it — ADVICE r3).
            mine, failure = None, None
            try:
                mine = ctx.peer_halo_export(max_fields=4, max_rows=self.rows)
            except Exception as exc:  # noqa: BLE001 — reported below, on every rank
                failure = f"rank {self.rank}: {exc}"
            handles = [None] * self.world
            dist.all_gather_object(handles, (mine, failure))
            failures = [f for _, f in handles if f is not None]
            if failures:
                self.peer_fallback_reason = "; ".join(failures)
                self.backend = "rccl"
            else:
                handles = [h for h, _ in handles]
                ctx.peer_halo_connect(handles[self.rank - 1] if self.rank > 0 else None,
                                      handles[self.rank + 1] if self.rank < self.world - 1 else None, self.rank, self.world)
        if self.backend == "rccl":
            from . import runtime
            # rank 0 always reaches the broadcast — with the id or with the reason it has none — so that no rank is left
            # waiting in it; every rank then raises the same error
            ident = [None]
            if self.rank == 0:
                try:
                    ident = [(runtime.comm_unique_id(), None)]
                except Exception as exc:  # noqa: BLE001
                    ident = [(None, f"rank 0: {exc}")]
            dist.broadcast_object_list(ident, src=0)
            uid, failure = ident[0]
            if failure is not None:
                raise RuntimeError(f"no RCCL unique id ({failure})")
            ctx.comm_init(uid, self.rank, self.world)

    def __call__(self, tensors):
        if self.backend == "rccl":
            self.ctx.halo_exchange_rows(tensors, self.rows)
        elif self.backend == "peer":
            self.ctx.halo_exchange_rows_peer(tensors, self.rows)
        elif self.backend == "torch":
            exchange_halo_rows_torch(tensors, self.ny, self.hy, self.rows)
