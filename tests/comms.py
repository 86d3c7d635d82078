"""Communicators for the Z-slab tests: in-process threads (one GPU, several contexts) and gloo (CPU, 2 processes)."""
import threading

import numpy as np


class ThreadGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm:
    """Host-mediated exchange between contexts living in one process (nl_planes_get / nl_planes_put)."""

    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world

    def _all(self, value):
        self.g.slots[self.rank] = value
        self.g.barrier.wait()
        vals = list(self.g.slots)
        self.g.barrier.wait()
        return vals

    def exchange_halo(self, ctx, field, depth, offset=0, run_async=False):
        lo, hi = ctx.own
        down = ctx.planes_get(field, lo + offset, lo + offset + depth) if self.rank > 0 else None
        up = ctx.planes_get(field, hi - offset - depth, hi - offset) if self.rank + 1 < self.world else None
        vals = self._all((down, up))
        if self.rank > 0:
            ctx.planes_put(field, lo - offset - depth, lo - offset, vals[self.rank - 1][1])
        if self.rank + 1 < self.world:
            ctx.planes_put(field, hi + offset, hi + offset + depth, vals[self.rank + 1][0])

    def exchange_bits(self, ctx, which):
        lo, hi = ctx.own
        down = ctx.slab_bits_get(which, lo) if self.rank > 0 else None
        up = ctx.slab_bits_get(which, hi - 1) if self.rank + 1 < self.world else None
        vals = self._all((down, up))
        if self.rank > 0:
            ctx.slab_bits_put(which, lo - 1, vals[self.rank - 1][1])
        if self.rank + 1 < self.world:
            ctx.slab_bits_put(which, hi, vals[self.rank + 1][0])

    def allgather_list(self, arr):
        return self._all(np.array(arr, copy=True))

    def allreduce(self, arr, op):
        vals = self._all(np.array(arr, copy=True))
        st = np.stack(vals)
        return {"sum": st.sum(0), "min": st.min(0), "max": st.max(0)}[op].astype(arr.dtype)

    def allgather(self, arr):
        return np.concatenate(self._all(np.array(arr, copy=True)))

    def allgather_mask_bits(self, ctx, slab_plane0):
        ny = ctx.shape[1]
        p0, p1 = slab_plane0[self.rank], slab_plane0[self.rank + 1]
        mine = ctx.label_bits_get(p0 * ny, (p1 - p0) * ny)
        for r, words in enumerate(self._all(mine)):
            if r != self.rank:
                ctx.label_bits_put(slab_plane0[r] * ny, words)


class GlooComm:
    """torch.distributed (gloo) version of the same, for the world_size-2 CPU tests."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def exchange_halo(self, ctx, field, depth, offset=0, run_async=False):
        import torch
        lo, hi = ctx.own
        reqs, recv_lo, recv_hi = [], None, None
        if self.rank > 0:
            reqs.append(self.dist.isend(torch.from_numpy(ctx.planes_get(field, lo + offset, lo + offset + depth)), self.rank - 1))
            recv_lo = torch.empty((depth,) + tuple(ctx.shape[1:]), dtype=torch.float32)
            reqs.append(self.dist.irecv(recv_lo, self.rank - 1))
        if self.rank + 1 < self.world:
            reqs.append(self.dist.isend(torch.from_numpy(ctx.planes_get(field, hi - offset - depth, hi - offset)), self.rank + 1))
            recv_hi = torch.empty((depth,) + tuple(ctx.shape[1:]), dtype=torch.float32)
            reqs.append(self.dist.irecv(recv_hi, self.rank + 1))
        for r in reqs:
            r.wait()
        if recv_lo is not None:
            ctx.planes_put(field, lo - offset - depth, lo - offset, recv_lo.numpy())
        if recv_hi is not None:
            ctx.planes_put(field, hi + offset, hi + offset + depth, recv_hi.numpy())

    def exchange_bits(self, ctx, which):
        lo, hi = ctx.own
        out = [None] * self.world
        mine = (ctx.slab_bits_get(which, lo) if self.rank > 0 else None,
                ctx.slab_bits_get(which, hi - 1) if self.rank + 1 < self.world else None)
        self.dist.all_gather_object(out, mine)
        if self.rank > 0:
            ctx.slab_bits_put(which, lo - 1, out[self.rank - 1][1])
        if self.rank + 1 < self.world:
            ctx.slab_bits_put(which, hi, out[self.rank + 1][0])

    def allgather_list(self, arr):
        out = [None] * self.world
        self.dist.all_gather_object(out, np.array(arr, copy=True))
        return out

    def allreduce(self, arr, op):
        import torch
        t = torch.from_numpy(np.array(arr, copy=True))
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN,
                                    "max": self.dist.ReduceOp.MAX}[op])
        return t.numpy()

    def allgather(self, arr):
        out = [None] * self.world
        self.dist.all_gather_object(out, np.array(arr, copy=True))
        return np.concatenate(out)
