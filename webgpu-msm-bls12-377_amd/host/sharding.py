"""Sharding one MSM across the GPUs of one node: by windows, or by points.

POINT sharding (``ShardedMsm.run_points``; SURVEY.md section 8e's fallback): rank g runs a complete MSM over its slice
of the points and scalars -- nothing is replicated -- and the exchange is one all-gather of 96-byte results, added up on
every rank (msm377_g1_add_points).  WINDOW sharding (below) is what the reference's structure suggests and what
BASELINE.json's config 4 names; it replicates the base conversion and keeps the per-window fixed costs on every rank.
``choose_partition`` picks between them from the measured per-rank times.

Window sharding:

The 16 window subtasks are independent after decomposition -- the reference already loops
over them in groups of four (src/submission/submission.ts:199-224) -- so rank g computes a
contiguous block of windows on its own GPU with no data-path collective, and ONE exchange
follows: an all-gather of the per-window partial records (16 x 208 bytes per window,
include/msm377.h) over RCCL/xGMI, a few KB in total, latency-bound.  Point addition is not an
RCCL reduction op, so it is gather-then-add: the Horner combine runs on the host of every
rank (msm377_g1_combine_partials).
"""
from typing import Callable, Optional, Tuple

from .engine import EEXCEPTIONAL, EHIP, GLV_WINDOWS, NUM_WINDOWS, WINDOW_PARTIAL_BYTES, MsmError, add_points_bytes, combine_partials_bytes, fold_partials_bytes


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


def points_for_rank(rank: int, world_size: int, n: int) -> Tuple[int, int]:
    """(first, count) of rank's contiguous slice of the n points / scalars; counts differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(n, world_size)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def choose_partition(n: int, world_size: int) -> str:
    """"points" or "windows" for an n-point MSM on world_size GPUs, from the per-rank times measured on one MI355X
    (tools/time_shard.py, profiles/r03_final/time_shard_2e20.txt and _2e22.txt): a rank's share
    under window sharding keeps the whole base conversion, the sort set-up, a 15-level reduction and the combine,
    under point sharding everything shrinks with n / G -- so points win as soon as there is more than one rank,
    narrowly at 2 ranks (1.82 vs 1.76 ms per rank at 2^20) and by 2x at 8 (0.95 vs 0.47)."""
    return "points" if world_size > 1 else "windows"


# Status word every rank appends to its slot of an exchange, so that a rank whose local work failed does not leave the
# others waiting in the collective for the RCCL timeout: all ranks enter the all-gather, read every status and raise
# together.  0 = fine; otherwise the negative MSM377_E* code (as u32) or STATUS_PYERR for any other exception.
STATUS_BYTES = 16  # keeps the slots 16-byte aligned
STATUS_PYERR = 0x7FFFFFFF


def _status_word(exc) -> int:
    if exc is None:
        return 0
    return (exc.code & 0xFFFFFFFF) if isinstance(exc, MsmError) else STATUS_PYERR


def _raise_agreed(statuses, local_exc, what):
    """Called on every rank with every rank's status: raises the local exception where there is one, and an MsmError
    naming the first failed rank everywhere else."""
    bad = [(r, s) for r, s in enumerate(statuses) if s]
    if not bad:
        return
    if local_exc is not None:
        raise local_exc
    r, s = bad[0]
    code = s - (1 << 32) if s & 0x80000000 else EHIP
    raise MsmError(code, "%s: rank %d failed (status 0x%x); every rank abandons the exchange" % (what, r, s))


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
        self.rec_bytes = self.max_count * WINDOW_PARTIAL_BYTES
        self.slot = self.rec_bytes + STATUS_BYTES  # records, then this rank's status word
        self.counts = [windows_for_rank(r, world_size, num_windows)[1] for r in range(world_size)]
        self.force_collective = force_collective
        if world_size > 1 or force_collective:
            pin = device is not None and str(device).startswith("cuda")
            self.send_host = torch.zeros(self.slot, dtype=torch.uint8, pin_memory=pin)
            self.recv_host = torch.zeros(self.slot * world_size, dtype=torch.uint8, pin_memory=pin)
            dev = device if device is not None else "cpu"
            self.send_dev = torch.zeros(self.slot, dtype=torch.uint8, device=dev)
            self.recv_dev = torch.zeros(self.slot * world_size, dtype=torch.uint8, device=dev)
            # point sharding: one 96-byte result + status per rank
            self.pslot = 96 + STATUS_BYTES
            self.psend_host = torch.zeros(self.pslot, dtype=torch.uint8, pin_memory=pin)
            self.precv_host = torch.zeros(self.pslot * world_size, dtype=torch.uint8, pin_memory=pin)
            self.psend_dev = torch.zeros(self.pslot, dtype=torch.uint8, device=dev)
            self.precv_dev = torch.zeros(self.pslot * world_size, dtype=torch.uint8, device=dev)

    def _set_status(self, host_tensor, offset: int, exc):
        host_tensor.numpy()[offset : offset + 4] = memoryview(_status_word(exc).to_bytes(4, "little"))

    @staticmethod
    def _statuses(flat, slot: int, offset: int, world: int):
        return [int.from_bytes(flat[r * slot + offset : r * slot + offset + 4].tobytes(), "little") for r in range(world)]

    def run_points(self, msm_fn: Callable[[int, int], bytes], n: int) -> bytes:
        """Point sharding: ``msm_fn(first, count)`` returns the 96-byte MSM of this rank's slice of the points and scalars
        (MsmEngine.msm_device on offset device pointers: a complete MSM with its own fallbacks and host tail); ONE
        all-gather of the results (112 bytes per rank) follows and every rank adds them up.  A rank whose MSM raised
        still enters the collective, with its status word set, and then every rank raises."""
        first, count = points_for_rank(self.rank, self.world, n)
        if self.world == 1 and not self.force_collective:
            return msm_fn(first, count)
        import torch.distributed as dist

        exc, mine = None, b"\x00" * 96
        try:
            mine = msm_fn(first, count)
            if len(mine) != 96:
                raise ValueError("msm_fn returned %d bytes" % len(mine))
        except Exception as e:  # noqa: BLE001 -- agreed on below, re-raised on this rank
            exc = e
        self.psend_host.numpy()[:96] = memoryview(mine if exc is None else b"\x00" * 96)
        self._set_status(self.psend_host, 96, exc)
        self.psend_dev.copy_(self.psend_host, non_blocking=True)
        dist.all_gather_into_tensor(self.precv_dev, self.psend_dev, group=self.group)
        self.precv_host.copy_(self.precv_dev)  # synchronising D2H
        flat = self.precv_host.numpy()
        _raise_agreed(self._statuses(flat, self.pslot, 96, self.world), exc, "point-sharded MSM")
        return add_points_bytes(b"".join(flat[r * self.pslot : r * self.pslot + 96].tobytes() for r in range(self.world)))

    def run(self, partials_fn: Callable[[int, int], bytes]) -> bytes:
        if self.world == 1:
            mine = partials_fn(self.begin, self.count) if self.count else b""
            if len(mine) != self.count * WINDOW_PARTIAL_BYTES:
                raise ValueError("partials_fn returned %d bytes for %d windows" % (len(mine), self.count))
            return combine_partials_bytes(mine, self.num_windows)
        import torch.distributed as dist

        exc, mine = None, b""
        try:
            mine = partials_fn(self.begin, self.count) if self.count else b""
            if len(mine) != self.count * WINDOW_PARTIAL_BYTES:
                raise ValueError("partials_fn returned %d bytes for %d windows" % (len(mine), self.count))
            # This rank's share of the host tail: fold its own windows into one point before the exchange, so that the
            # combine every rank runs afterwards is doublings plus one addition per rank (the tail does not shrink
            # with the number of GPUs otherwise).
            mine = fold_partials_bytes(mine)
        except Exception as e:  # noqa: BLE001 -- every rank still enters the collective; agreed on below
            exc, mine = e, b""
        if mine:
            self.send_host.numpy()[: len(mine)] = memoryview(mine)
        self._set_status(self.send_host, self.rec_bytes, exc)
        self.send_dev.copy_(self.send_host, non_blocking=True)
        dist.all_gather_into_tensor(self.recv_dev, self.send_dev, group=self.group)
        self.recv_host.copy_(self.recv_dev)  # synchronising D2H (53 KB at most)
        flat = self.recv_host.numpy()
        _raise_agreed(self._statuses(flat, self.slot, self.rec_bytes, self.world), exc, "window-sharded MSM")
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
            exc = None
            try:
                if self.count:
                    fn(self.begin, self.count, self.send_dev.data_ptr())
            except Exception as e:  # noqa: BLE001 -- a rank that failed (EHIP, ENOMEM, ...) still enters the collective
                exc = e
            # the status word rides behind the records in the same slot (a 16-byte H2D into the send buffer)
            self._set_status(self.send_host, self.rec_bytes, exc)
            self.send_dev[self.rec_bytes :].copy_(self.send_host[self.rec_bytes :], non_blocking=True)
            dist.all_gather_into_tensor(self.recv_dev, self.send_dev, group=self.group)
            self.recv_host.copy_(self.recv_dev)  # synchronising D2H
            flat = self.recv_host.numpy()
            _raise_agreed(self._statuses(flat, self.slot, self.rec_bytes, self.world), exc, "window-sharded MSM")
            # the gathered slots carry a status word each, so the 16 records are re-packed for the combine (48 KB)
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
