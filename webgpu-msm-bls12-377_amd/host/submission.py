"""compute_msm: the drop-in entry point, same name, arguments and result as the reference.

Reference: src/submission/submission.ts:85-327
    export const compute_msm = async (baseAffinePoints: BigIntPoint[] | U32ArrayPoint[] | Buffer,
                                      scalars: bigint[] | Uint32Array[] | Buffer,
                                      log_result = true, force_recompile = false)
                                      : Promise<{x: bigint, y: bigint}>
Callers: src/ui/Benchmark.tsx:32, src/submission/miscellaneous/full_benchmarks.ts:62,99.
The TypeScript twin for src/submission lives in ../node/compute_msm.ts; this module is the
Python mirror used by the parity tests and bench.py.  Both call the C ABI (include/msm377.h).
"""
from typing import Dict, Optional, Sequence, Union

from .codecs import bigIntsToBufferLE, u32ArrayToBigInts
from .engine import MsmEngine

_ENGINE: Optional[MsmEngine] = None

BufferLike = Union[bytes, bytearray, memoryview]


def _is_buffer(obj) -> bool:
    return isinstance(obj, (bytes, bytearray, memoryview))


def _coord(pt, name):
    return pt[name] if isinstance(pt, dict) else getattr(pt, name)


def _to_int(v) -> int:
    """bigint, or a Uint32Array of most-significant-first words (src/reference/webgpu/utils.ts:49-61)."""
    if isinstance(v, int):
        return v
    return u32ArrayToBigInts(list(v), 32 * len(v))[0]


def points_to_buffer(baseAffinePoints) -> bytes:
    """BigIntPoint[] / U32ArrayPoint[] -> the harness's points Buffer
    (x || y, 384-bit little-endian each: src/ui/AllBenchmarks.tsx:57-63)."""
    if _is_buffer(baseAffinePoints):
        return bytes(baseAffinePoints)
    xy = []
    for pt in baseAffinePoints:
        if isinstance(pt, (tuple, list)):
            x, y = pt[0], pt[1]
        else:
            x, y = _coord(pt, "x"), _coord(pt, "y")
        xy.append(_to_int(x))
        xy.append(_to_int(y))
    return bigIntsToBufferLE(xy, 384)


def scalars_to_buffer(scalars) -> bytes:
    """bigint[] / Uint32Array[] -> the harness's scalars Buffer (256-bit LE: AllBenchmarks.tsx:67)."""
    if _is_buffer(scalars):
        return bytes(scalars)
    return bigIntsToBufferLE([_to_int(s) for s in scalars], 256)


def _engine_for(n: int) -> MsmEngine:
    global _ENGINE
    if _ENGINE is None or _ENGINE.max_points < n:
        if _ENGINE is not None:
            _ENGINE.close()
        cap = 1 << 16
        while cap < n:
            cap <<= 1
        _ENGINE = MsmEngine(cap)
    return _ENGINE


def compute_msm(baseAffinePoints, scalars, log_result: bool = True, force_recompile: bool = False) -> Dict[str, int]:
    """Q = sum k_i P_i over BLS12-377 G1; returns affine {"x": int, "y": int}.

    Same contract as the reference: input_size = len(scalars buffer) / 32, empty input gives
    {x: 0, y: 1} (submission.ts:91-95); errors raise (MsmError) instead of rejecting a Promise.
    ``force_recompile`` exists for signature parity only: kernels are compiled ahead of time
    for gfx950, there is no runtime shader cache to defeat (shader_manager.ts:71-77).
    """
    del force_recompile
    sbuf = scalars_to_buffer(scalars)
    pbuf = points_to_buffer(baseAffinePoints)
    if len(sbuf) % 32:
        raise ValueError("scalars buffer length must be a multiple of 32")
    n = len(sbuf) // 32
    if n == 0:
        result = {"x": 0, "y": 1}
    else:
        out = _engine_for(n).msm(pbuf, sbuf)
        result = {"x": int.from_bytes(out[:48], "little"), "y": int.from_bytes(out[48:], "little")}
    if log_result:
        print(result)
    return result
