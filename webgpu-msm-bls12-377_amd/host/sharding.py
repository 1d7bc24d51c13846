"""Window sharding across the GPUs of one node.

The 16 window subtasks are independent after decomposition -- the reference already loops
over them in groups of four (src/submission/submission.ts:199-224) -- so rank g computes a
contiguous block of windows on its own GPU with no data-path collective, and ONE exchange
follows: an all-gather of the per-window partial records (16 x 208 bytes per window,
include/msm377.h) over RCCL/xGMI, a few KB in total, latency-bound.  Point addition is not an
RCCL reduction op, so it is gather-then-add: the Horner combine runs on the host of every
rank (msm377_g1_combine_partials).
"""
from typing import Callable, Optional, Tuple

from .engine import EEXCEPTIONAL, GLV_WINDOWS, NUM_WINDOWS, WINDOW_PARTIAL_BYTES, MsmError, combine_partials_bytes, fold_partials_bytes


def windows_for_rank(rank: int, world_size: int, num_windows: int = NUM_WINDOWS) -> Tuple[int, int]:
    """(win_begin, win_count) of rank's contiguous window block; counts differ by at most one
    and ranks beyond num_windows get (num_windows, 0)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(num_windows, world_size)
    count = base + (1 if rank < rem else 0)
    begin = rank * base + min(rank, rem)
    return begin, count


def combine_partials(partials: bytes) -> bytes:
    return combine_partials_bytes(partials)


class ShardedMsm:
    """sharded_msm with the exchange buffers allocated once (bench.py, services): a pinned host
    staging tensor, a send tensor and a world_size-slot receive tensor on ``device``."""

    def __init__(self, rank: int, world_size: int, group=None, device=None, num_windows: int = NUM_WINDOWS, force_collective: bool = False):
        """``force_collective``: allocate the exchange buffers and issue the collective even for one rank (a rehearsal of
        the RCCL path on a one-GPU box; a real single-rank run calls the engine's full MSM instead)."""
        import torch

        self.rank, self.world, self.group = rank, world_size, group
        self.num_windows = num_windows  # 16 on the plain path, GLV_WINDOWS = 8 behind the GLV front end
        self.begin, self.count = windows_for_rank(rank, world_size, num_windows)
        self.max_count = (num_windows + world_size - 1) // world_size
        self.slot = self.max_count * WINDOW_PARTIAL_BYTES
        self.counts = [windows_for_rank(r, world_size, num_windows)[1] for r in range(world_size)]
        self.force_collective = force_collective
        if world_size > 1 or force_collective:
            pin = device is not None and str(device).startswith("cuda")
            self.send_host = torch.zeros(self.slot, dtype=torch.uint8, pin_memory=pin)
            self.recv_host = torch.zeros(self.slot * world_size, dtype=torch.uint8, pin_memory=pin)
            dev = device if device is not None else "cpu"
            self.send_dev = torch.zeros(self.slot, dtype=torch.uint8, device=dev)
            self.recv_dev = torch.zeros(self.slot * world_size, dtype=torch.uint8, device=dev)

    def run(self, partials_fn: Callable[[int, int], bytes]) -> bytes:
        mine = partials_fn(self.begin, self.count) if self.count else b""
        if len(mine) != self.count * WINDOW_PARTIAL_BYTES:
            raise ValueError("partials_fn returned %d bytes for %d windows" % (len(mine), self.count))
        if self.world == 1:
            return combine_partials_bytes(mine, self.num_windows)
        import torch.distributed as dist

        # This rank's share of the host tail: fold its own windows into one point before the exchange, so that the
        # combine every rank runs afterwards is doublings plus one addition per rank (the tail does not shrink
        # with the number of GPUs otherwise).
        mine = fold_partials_bytes(mine)
        if self.count:
            self.send_host.numpy()[: len(mine)] = memoryview(mine)
        self.send_dev.copy_(self.send_host, non_blocking=True)
        dist.all_gather_into_tensor(self.recv_dev, self.send_dev, group=self.group)
        self.recv_host.copy_(self.recv_dev)  # synchronising D2H (53 KB at most)
        flat = self.recv_host.numpy()
        parts = [flat[r * self.slot : r * self.slot + c * WINDOW_PARTIAL_BYTES].tobytes() for r, c in enumerate(self.counts)]
        return combine_partials_bytes(b"".join(parts), self.num_windows)

    def run_resident(self, write_records: Callable[[int, int, int], None], combine: Optional[Callable[[bytes], bytes]] = None,
                     rerun_weierstrass: Optional[Callable[[int, int, int], None]] = None) -> bytes:
        """The RCCL path: the records never visit the host before the exchange.

        ``write_records(win_begin, win_count, d_out)`` leaves this rank's window records in device memory at
        ``d_out`` (MsmEngine.window_partials_resident bound to the resident inputs; it returns with the engine's
        stream idle).  ONE all-gather straight from that buffer over RCCL/xGMI follows, then one D2H of the
        gathered records (49 KB at most) and the host combine on every rank -- ``combine`` is normally
        MsmEngine.combine_partials (the context's tail threads).  Records that add up to an exceptional case of
        the twisted Edwards law (MsmError EEXCEPTIONAL; every rank sees the same records, so every rank gets the
        same verdict) are recomputed through ``rerun_weierstrass`` (same signature; the engine in form 0) and
        exchanged again.
        """
        if self.world == 1 and not self.force_collective:
            raise ValueError("run_resident is the multi-rank path; a single rank calls the engine's full MSM")
        import torch.distributed as dist

        for attempt in (0, 1):
            fn = write_records if attempt == 0 else rerun_weierstrass
            if self.count:
                fn(self.begin, self.count, self.send_dev.data_ptr())
            dist.all_gather_into_tensor(self.recv_dev, self.send_dev, group=self.group)
            self.recv_host.copy_(self.recv_dev)  # synchronising D2H
            if combine is not None and self.num_windows == NUM_WINDOWS and all(c == self.max_count for c in self.counts):
                parts = self.recv_host.data_ptr()  # every slot is full: the gathered buffer IS the 16 records, combined in place
            else:
                flat = self.recv_host.numpy()
                parts = b"".join(flat[r * self.slot : r * self.slot + c * WINDOW_PARTIAL_BYTES].tobytes() for r, c in enumerate(self.counts))
            try:
                return combine(parts) if combine is not None else combine_partials_bytes(parts, self.num_windows)
            except MsmError as e:
                if e.code != EEXCEPTIONAL or attempt == 1 or rerun_weierstrass is None:
                    raise
        raise AssertionError("unreachable")


def sharded_msm(
    partials_fn: Callable[[int, int], bytes],
    rank: int,
    world_size: int,
    group=None,
    device=None,
) -> bytes:
    """Run this rank's windows through ``partials_fn(win_begin, win_count)`` (normally
    MsmEngine.window_partials_device bound to the resident inputs), all-gather the records and
    return the affine result (96 bytes) on every rank.

    With world_size == 1 no collective is issued.  ``device`` is the torch device of the
    exchange buffer ("cuda:<local_rank>" under RCCL, "cpu" under gloo).
    """
    begin, count = windows_for_rank(rank, world_size)
    mine = partials_fn(begin, count) if count else b""
    if len(mine) != count * WINDOW_PARTIAL_BYTES:
        raise ValueError("partials_fn returned %d bytes for %d windows" % (len(mine), count))
    if world_size == 1:
        return combine_partials_bytes(mine)

    import torch
    import torch.distributed as dist

    mine = fold_partials_bytes(mine)  # this rank's share of the host tail, see ShardedMsm.run
    max_count = (NUM_WINDOWS + world_size - 1) // world_size
    slot = max_count * WINDOW_PARTIAL_BYTES
    send = torch.zeros(slot, dtype=torch.uint8)
    if count:
        send[: len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    if device is not None:
        send = send.to(device)
    recv = torch.empty(slot * world_size, dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    flat = recv.cpu().numpy().tobytes()
    parts = []
    for r in range(world_size):
        _, c = windows_for_rank(r, world_size)
        parts.append(flat[r * slot : r * slot + c * WINDOW_PARTIAL_BYTES])
    return combine_partials_bytes(b"".join(parts))
