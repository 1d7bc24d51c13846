"""Wire codecs of the test harness, mirrored name for name.

Reference: src/reference/webgpu/utils.ts:41-99 (bigIntsToU32Array, bigIntToU32Array,
bigIntBufferLE, bigIntsToBufferLE, readBigIntsFromBufferLE, u32ArrayToBigInts); vectors in
src/reference/webgpu/utils.test.ts:4-14.  These define the buffers compute_msm receives
(src/ui/AllBenchmarks.tsx:57-68): points as x||y 384-bit little-endian, scalars 256-bit
little-endian.
"""
from typing import Iterable, List


def bigIntToU32Array(beBigInt: int, bigIntSize: int = 256) -> List[int]:
    """Most-significant-first u32 words (utils.ts:49-61)."""
    num = bigIntSize // 32
    return [(beBigInt >> (32 * (num - 1 - i))) & 0xFFFFFFFF for i in range(num)]


def bigIntsToU32Array(beBigInts: Iterable[int], bigIntSize: int = 256) -> List[int]:
    """Concatenation of bigIntToU32Array (utils.ts:41-46)."""
    out: List[int] = []
    for v in beBigInts:
        out.extend(bigIntToU32Array(v, bigIntSize))
    return out


def u32ArrayToBigInts(u32Array: Iterable[int], bigIntSize: int = 256) -> List[int]:
    """Inverse of bigIntsToU32Array (utils.ts:87-103)."""
    words = list(u32Array)
    chunk = bigIntSize // 32
    out = []
    for i in range(0, len(words), chunk):
        v = 0
        for w in words[i : i + chunk]:
            v = (v << 32) | (int(w) & 0xFFFFFFFF)
        if len(words[i : i + chunk]) < chunk:  # the reference tolerates a short tail
            v <<= 32 * (chunk - len(words[i : i + chunk]))
        out.append(v)
    return out


def bigIntBufferLE(bigInt: int, bigIntSize: int = 256) -> bytes:
    """Little-endian bytes of one integer (utils.ts:63-67)."""
    return int(bigInt).to_bytes(bigIntSize // 8, "little")


def bigIntsToBufferLE(bigInts: Iterable[int], bigIntSize: int = 256) -> bytes:
    """Concatenated little-endian integers (utils.ts:69-72)."""
    return b"".join(bigIntBufferLE(v, bigIntSize) for v in bigInts)


def readBigIntsFromBufferLE(buffer: bytes, bigIntSize: int = 256) -> List[int]:
    """Inverse of bigIntsToBufferLE (utils.ts:74-85).  Unlike the reference (whose
    Buffer.reverse() on a slice view reverses the caller's buffer in place, utils.ts:78-79)
    the input is left untouched."""
    step = bigIntSize // 8
    buf = bytes(buffer)
    return [int.from_bytes(buf[i : i + step], "little") for i in range(0, len(buf) - step + 1, step)]
