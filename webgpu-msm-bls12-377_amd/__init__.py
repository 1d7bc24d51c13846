"""MI355X-native BLS12-377 MSM engine: Python host side.

The directory name carries hyphens (it is fixed by the project layout), so import it through
the repo-root shim ``webgpu_msm_bls12_377_amd`` or via importlib.  Everything here is a thin
mirror of the reference's operator interface over the C ABI in include/msm377.h; the
arithmetic lives in csrc/ (hand-written HIP for gfx950).  There is no CPU fallback: loading
fails loudly when csrc/libmsm377.so is missing, and every MSM call fails when no HIP device
is usable.
"""
from .host.codecs import (  # noqa: F401
    bigIntBufferLE,
    bigIntsToBufferLE,
    bigIntsToU32Array,
    bigIntToU32Array,
    readBigIntsFromBufferLE,
    u32ArrayToBigInts,
)
from .host.engine import MsmEngine, MsmError, library_path, load_library  # noqa: F401
from .host.submission import compute_msm, points_to_buffer, scalars_to_buffer  # noqa: F401
from .host.sharding import combine_partials, windows_for_rank  # noqa: F401

__all__ = [
    "compute_msm",
    "MsmEngine",
    "MsmError",
    "load_library",
    "library_path",
    "windows_for_rank",
    "combine_partials",
    "points_to_buffer",
    "scalars_to_buffer",
    "bigIntBufferLE",
    "bigIntsToBufferLE",
    "readBigIntsFromBufferLE",
    "bigIntToU32Array",
    "bigIntsToU32Array",
    "u32ArrayToBigInts",
]
