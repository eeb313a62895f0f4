"""Tile partition + gather of one frame across the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  The tree is replicated, the frame is cut into interleaved macro tiles
(rank = tile % world, optionally relieving rank 0; see ``mnv_partition`` in include/mnv.h), each rank renders its tiles into a
compact local-tile-major buffer with one kernel launch, and the buffers are gathered to rank 0
and un-permuted into the frame there.  The march itself needs no exchange: this gather is the only
collective of the path.  The reference has no multi-GPU code; this is new.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class TilePartition:
    """Index math of the interleaved macro-tile partition (pure host code, mirrors part_tile_of / part_owner_of in
    csrc/mnv_accel.h).  Tiles are dealt in rounds of `world`; with ``root_period`` M >= 2 every M-th round leaves rank 0 out, so
    that the rank which also receives the gather and un-permutes the frames renders (M - 1) / M of a plain share."""

    def __init__(self, width: int, height: int, world: int, tile_w: int = 64, tile_h: int = 24, root_period: int = 0):
        if tile_w % 8 or tile_h % 8:
            raise ValueError("macro tiles must be multiples of 8 pixels")
        if root_period < 0 or root_period == 1:
            raise ValueError("root_period must be 0 or >= 2")
        self.width, self.height, self.world = width, height, world
        self.tile_w, self.tile_h = tile_w, tile_h
        self.root_period = root_period if world > 1 else 0
        self.macros_x = -(-width // tile_w)
        self.macros_y = -(-height // tile_h)
        self.n_macro = self.macros_x * self.macros_y
        self.j_max = max(self.local_tiles(r) for r in range(world))

    def owner(self, m: int):
        """(rank, local index) of macro tile m."""
        M, w = self.root_period, self.world
        if M < 2:
            return m % w, m // w
        L = w * M - 1
        p, o = divmod(m, L)
        if o < (M - 1) * w:
            k, r = divmod(o, w)
        else:
            k, r = M - 1, o - (M - 1) * w + 1
        return r, p * (M - 1 if r == 0 else M) + k

    def tiles_of(self, rank: int) -> List[int]:
        return [m for m in range(self.n_macro) if self.owner(m)[0] == rank]

    def local_tiles(self, rank: int) -> int:
        if self.root_period < 2:
            return len(range(rank, self.n_macro, self.world))
        return sum(1 for m in range(self.n_macro) if self.owner(m)[0] == rank)

    def tile_rect(self, m: int):
        """(x0, y0, w, h) of macro tile m clipped to the frame."""
        mx, my = m % self.macros_x, m // self.macros_x
        x0, y0 = mx * self.tile_w, my * self.tile_h
        return x0, y0, min(self.tile_w, self.width - x0), min(self.tile_h, self.height - y0)

    def part(self, rank: int):
        """The tuple the binding's `part=` argument takes."""
        return (rank, self.world, self.tile_w, self.tile_h, self.root_period)

    def source_index(self, device) -> torch.Tensor:
        """For macro tile m: row of the [world * j_max] gathered tile table that holds it."""
        if self.root_period < 2:
            m = torch.arange(self.n_macro, device=device)
            return (m % self.world) * self.j_max + m // self.world
        rows = [r * self.j_max + j for r, j in (self.owner(m) for m in range(self.n_macro))]
        return torch.tensor(rows, dtype=torch.int64, device=device)

    def unpermute(self, gathered: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """gathered [world, j_max, tile_h, tile_w, C] -> frame [height, width, C]; with a frame dimension,
        gathered [world, F, j_max, tile_h, tile_w, C] -> frames [F, height, width, C]."""
        c = gathered.shape[-1]
        batched = gathered.dim() == 6
        if (gathered.is_cuda and out is not None and out.is_cuda and c == 4 and gathered.is_contiguous() and out.is_contiguous()
                and gathered.dtype == out.dtype and gathered.dtype in (torch.uint8, torch.float32)):
            # one pass over the pixels in libmnv (mnv_assemble_tiles) instead of index_select + permute + copy
            from . import assemble_tiles
            assemble_tiles(gathered, out, self.width, self.height, self.world, self.tile_w, self.tile_h,
                           n_frames=gathered.shape[1] if batched else 1, stream=torch.cuda.current_stream(gathered.device).cuda_stream,
                           root_period=self.root_period)
            return out
        g = gathered if batched else gathered.unsqueeze(1)
        f = g.shape[1]
        t = g.permute(1, 0, 2, 3, 4, 5).reshape(f, self.world * self.j_max, self.tile_h, self.tile_w, c)
        t = t[:, self.source_index(gathered.device)]
        t = t.view(f, self.macros_y, self.macros_x, self.tile_h, self.tile_w, c).permute(0, 1, 3, 2, 4, 5)
        frames = t.reshape(f, self.macros_y * self.tile_h, self.macros_x * self.tile_w, c)[:, : self.height, : self.width]
        if not batched:
            frames = frames[0]
        if out is None:
            return frames.contiguous()
        out.copy_(frames)
        return out


class TileGatherer:
    """Ring of `depth` in-flight steps: render into ``local(slot)`` on the current stream, then ``submit(slot)`` enqueues the
    gather to rank 0 and (on rank 0) the un-permute into ``frame(slot)`` on the SLOT'S OWN side stream; ``finish(slot)`` orders the
    slot's next render (current stream) after that gather.  Nothing here blocks the host, and nothing but the reuse of a slot's own
    buffer makes a render stream wait: with depth >= 2 the gather and assembly of step k overlap the render of step k+1 (on rank 0
    too, which would otherwise pay the un-permute between two renders).  One side stream per slot: with a shared one, the wait for
    march k+1 that precedes gather k+1 would also delay the completion record of gather k, and with it march k+2.

    The gather itself is ``comm.gather_tiles`` (libmnv's RCCL path, mnv_gather_tiles) when an ``mnv.Comm`` is given, otherwise
    ``torch.distributed.gather`` on `group` (gloo in the CPU tests and the one-GPU rehearsal)."""

    def __init__(self, part: TilePartition, rank: int, device, dtype=torch.float32, channels: int = 4, depth: int = 3, group=None,
                 frames: int = 0, stage_on_host: bool = False, comm=None, timing: bool = False):
        """`frames` > 0: every slot holds a batch of that many frames (one launch + one gather per batch).
        `stage_on_host`: gather through host copies (for process groups without device collectives, e.g. gloo
        in the single-GPU rehearsal of the N > 1 path); the default hands device buffers to RCCL."""
        self.part, self.rank, self.group, self.depth, self.comm = part, rank, group, depth, comm
        self.stage_on_host = stage_on_host
        lead = (frames,) if frames > 0 else ()
        shape = lead + (part.j_max, part.tile_h, part.tile_w, channels)
        self._local = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(depth)]
        self._pending = [None] * depth
        on_gpu = torch.device(device).type == "cuda"
        if comm is not None and (not on_gpu or stage_on_host):
            raise ValueError("the RCCL communicator gathers device buffers")
        self._side = [torch.cuda.Stream(device=device) for _ in range(depth)] if on_gpu and not stage_on_host else None
        self._sent = [torch.cuda.Event() for _ in range(depth)] if self._side is not None else None
        # `timing`: device time of every gather and (rank 0) un-permute on the side streams, for the per-rank report of bench.py --gpus N
        self._timing = bool(timing) and self._side is not None
        # three events per slot, made once and reused (gather begin, gather end, un-permute end); a slot's previous measurement is read
        # when the slot comes round again (its work finished `depth` steps ago) or by take_timings()
        self._t_slot = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(depth)] if self._timing else None
        self._t_armed = [False] * depth
        self._t_values = []   # (gather ms, un-permute ms or None)
        if rank == 0:
            self._gathered = [torch.empty((part.world,) + shape, dtype=dtype, device=device) for _ in range(depth)]
            self._frames = [torch.empty(lead + (part.height, part.width, channels), dtype=dtype, device=device) for _ in range(depth)]

    def local(self, slot: int) -> torch.Tensor:
        return self._local[slot]

    def frame(self, slot: int) -> torch.Tensor:
        return self._frames[slot]

    def submit(self, slot: int) -> None:
        if self.stage_on_host:
            src = self._local[slot].cpu()  # synchronises with the render on the current stream
            host = [torch.empty_like(src) for _ in range(self.part.world)] if self.rank == 0 else None
            self._pending[slot] = (dist.gather(src, host, dst=0, group=self.group, async_op=True), host, src)
            return
        glist = [self._gathered[slot][r] for r in range(self.part.world)] if self.rank == 0 else None
        if self._side is None:
            self._pending[slot] = dist.gather(self._local[slot], glist, dst=0, group=self.group, async_op=True)
            return
        side = self._side[slot]
        # the slot's side stream already holds the un-permute that last read this slot's gather table; the collective is ordered
        # after it and after the render just enqueued on the current stream
        side.wait_stream(torch.cuda.current_stream(side.device))
        with torch.cuda.stream(side):
            ev = None
            if self._timing:
                self._harvest(slot)
                ev = self._t_slot[slot]
                ev[0].record()
            if self.comm is not None:
                self.comm.gather_tiles(self._local[slot], self._gathered[slot] if self.rank == 0 else None, root=0, stream=side.cuda_stream)
            else:
                dist.gather(self._local[slot], glist, dst=0, group=self.group, async_op=True).wait()  # stream-level wait on `side`
            self._sent[slot].record()      # local(slot) may be overwritten from here on
            if ev:
                ev[1].record()
            if self.rank == 0:
                self.part.unpermute(self._gathered[slot], out=self._frames[slot])
                if ev:
                    ev[2].record()
            if ev:
                self._t_armed[slot] = True
        self._pending[slot] = True

    def _harvest(self, slot: int) -> None:
        if self._timing and self._t_armed[slot]:
            ev = self._t_slot[slot]
            ev[2 if self.rank == 0 else 1].synchronize()
            self._t_values.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]) if self.rank == 0 else None))
            self._t_armed[slot] = False

    def finish(self, slot: int) -> None:
        w = self._pending[slot]
        if w is None:
            return
        self._pending[slot] = None
        if self.stage_on_host:
            w[0].wait()
            if self.rank == 0:
                for r, h in enumerate(w[1]):
                    self._gathered[slot][r].copy_(h)
                self.part.unpermute(self._gathered[slot], out=self._frames[slot])
            return
        if self._side is None:
            w.wait()
            if self.rank == 0:
                self.part.unpermute(self._gathered[slot], out=self._frames[slot])
            return
        torch.cuda.current_stream(self._side[slot].device).wait_event(self._sent[slot])

    def wait_frame(self, slot: int) -> None:
        """Rank 0: order the current stream after the un-permute that fills ``frame(slot)`` (``finish`` only waits for the gather, which
        is what the next render into the slot needs).  A no-op for the host-staged and blocking variants, whose ``finish`` assembles."""
        if self._side is not None and self.rank == 0:
            torch.cuda.current_stream(self._side[slot].device).wait_stream(self._side[slot])

    def take_timings(self) -> dict:
        """Average device milliseconds per step of the gather (from its enqueue on the side stream, so waiting for the peers counts) and of
        rank 0's un-permute since the last call (call it once after the warm-up to discard the channel set-up of the first gather);
        empty without `timing`.  Waits for the slots' events."""
        if not self._timing:
            return {}
        for slot in range(self.depth):
            self._harvest(slot)
        if not self._t_values:
            return {}
        g = [v[0] for v in self._t_values]
        u = [v[1] for v in self._t_values if v[1] is not None]
        self._t_values = []
        out = {"gather_ms": sum(g) / len(g), "gathers": len(g)}
        if u:
            out["unpermute_ms"] = sum(u) / len(u)
        return out

    def finish_all(self) -> None:
        for s in range(self.depth):
            self.finish(s)
        if self._side is not None:
            # frames are read on the current stream (or the host) next
            for side in self._side:
                torch.cuda.current_stream(side.device).wait_stream(side)
