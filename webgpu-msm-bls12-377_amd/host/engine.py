"""ctypes binding of the C ABI in include/msm377.h (csrc/libmsm377.so).

Replaces the reference's device runtime wrappers (src/submission/implementation/cuzk/gpu.ts:2-170:
get_device, create_and_write_sb, create_compute_pipeline, execute_pipeline, read_from_gpu):
a context owns the HIP stream and every HBM buffer, and one call runs the whole pipeline.
No CPU fallback exists: a missing library raises at import-use time, a missing GPU raises
MsmError(MSM377_EHIP) at context creation.
"""
import ctypes
import os
import sys
from typing import List, Optional, Sequence, Tuple

NUM_WINDOWS = 16
WINDOW_BITS = 16
PARTIAL_POINTS = 16
POINT_WORDS = 52  # a bucket point in the device format (stage read-backs)
RECORD_POINT_WORDS = 48  # a point of a window partial record (host-tail format)
WINDOW_PARTIAL_BYTES = PARTIAL_POINTS * RECORD_POINT_WORDS * 4
NUM_BUCKETS = 32768
STAGE_NAMES = ("convert", "decompose", "sort", "accumulate", "reduce", "tail", "accumulate_kernel")

OK, EINVAL, EHIP, ESCALAR, ENOMEM, ESTATE, EGLVRANGE, EEXCEPTIONAL = 0, -1, -2, -3, -4, -5, -6, -7
# msm377_ctx_get_fallback_info: where an exceptional case of the twisted Edwards law surfaced (include/msm377.h)
FB_ACCUMULATE, FB_MERGE, FB_TREE, FB_TAIL, FB_CONVERT = 4, 8, 16, 32, 64
GLV_WINDOWS = 8

_LIB = None


class MsmError(RuntimeError):
    """Raised for any non-zero return of the C ABI (the reference throws Error /
    AssertionError, src/submission/implementation/cuzk/gpu.ts:7-10, submission.ts:405)."""

    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        msg = "%s failed: %s (%d)" % (what, _strerror(code), code)
        if detail:
            msg += ": " + detail
        super().__init__(msg)


def library_path() -> str:
    """csrc/libmsm377.so next to this package.  MSM377_LIB points tools/ab_libs.sh at another BUILD of the same
    library (an A/B of two engine versions on one box); it is never a different implementation."""
    override = os.environ.get("MSM377_LIB")
    if override:
        return os.path.abspath(override)
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.normpath(os.path.join(here, "..", "csrc", "libmsm377.so"))


def load_library():
    """Load csrc/libmsm377.so (built by __graft_entry__.build() / make -C csrc)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "msm377: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C webgpu-msm-bls12-377_amd/csrc`; there is no CPU fallback" % path
        )
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 with the same
    # SONAME as /opt/rocm's.  If this library pulled in the system copy first, a later `import torch`
    # would bind to it and fail ("No HIP GPUs are available"); loading torch's first makes both share
    # one runtime (measured working on the MI355X box).  Hosts without torch (node, C) use /opt/rocm's.
    if "torch" not in sys.modules and os.environ.get("MSM377_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing
            pass
    lib = ctypes.CDLL(path)
    u8p, vp, u64, u32, i32 = ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    sigs = {
        "msm377_version": (ctypes.c_char_p, []),
        "msm377_strerror": (ctypes.c_char_p, [i32]),
        "msm377_ctx_create": (i32, [i32, u64, ctypes.POINTER(vp)]),
        "msm377_ctx_destroy": (None, [vp]),
        "msm377_last_error": (ctypes.c_char_p, [vp]),
        "msm377_g1_msm": (i32, [vp, u8p, u8p, u64, vp]),
        "msm377_g1_msm_device": (i32, [vp, vp, vp, u64, vp]),
        "msm377_g1_set_bases": (i32, [vp, u8p, u64]),
        "msm377_g1_set_bases_device": (i32, [vp, vp, u64]),
        "msm377_g1_set_bases_precomputed": (i32, [vp, u8p, u64]),
        "msm377_g1_set_bases_precomputed_device": (i32, [vp, vp, u64]),
        "msm377_g1_msm_fixed_base": (i32, [vp, u8p, u64, vp]),
        "msm377_g1_msm_fixed_base_device": (i32, [vp, vp, u64, vp]),
        "msm377_g1_msm_fixed_base_batch_device": (i32, [vp, vp, u64, u32, vp]),
        "msm377_g1_window_partials_device": (i32, [vp, vp, vp, u64, u32, u32, vp]),
        "msm377_g1_window_partials_resident": (i32, [vp, vp, vp, u64, u32, u32, vp]),
        "msm377_g1_combine_partials": (i32, [vp, vp]),
        "msm377_g1_combine_partials_split": (i32, [vp, ctypes.c_uint32, vp]),
        "msm377_g1_combine_partials_ctx": (i32, [vp, vp, vp]),
        "msm377_ctx_get_fallback_info": (i32, [vp, ctypes.POINTER(u64), ctypes.POINTER(u32)]),
        "msm377_g1_glv_window_partials_device": (i32, [vp, vp, vp, u64, u32, u32, vp]),
        "msm377_g1_combine_window_partials": (i32, [vp, u32, vp]),
        "msm377_g1_generate_bases_device": (i32, [vp, u64, u64, vp]),
        "msm377_ed_msm": (i32, [vp, u8p, u8p, u64, vp]),
        "msm377_ed_msm_device": (i32, [vp, vp, vp, u64, vp]),
        "msm377_ed_generate_bases_device": (i32, [vp, u64, u64, vp]),
        "msm377_ctx_set_stage_capture": (i32, [vp, i32]),
        "msm377_g1_read_stage": (i32, [vp, u32, vp, vp, vp, vp]),
        "msm377_g1_xyzz_to_affine": (i32, [vp, vp]),
        "msm377_g1_fold_window_partials": (i32, [vp, ctypes.c_uint32]),
        "msm377_ctx_set_glv": (i32, [vp, i32]),
        "msm377_ctx_set_g1_form": (i32, [vp, i32]),
        "msm377_ctx_set_timing": (i32, [vp, i32]),
        "msm377_ctx_get_stage_ms": (i32, [vp, vp]),
        "msm377_ctx_get_products_per_addition": (i32, [vp]),
        "msm377_ctx_set_precompute_window": (i32, [vp, i32]),
        "msm377_g1_add_points": (i32, [vp, u32, vp]),
        "msm377_ctx_reserve_host_staging": (i32, [vp]),
        "msm377_ctx_get_stage_form": (i32, [vp]),
        "msm377_ctx_set_narrow_max": (i32, [vp, u64]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _strerror(code: int) -> str:
    try:
        return load_library().msm377_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "error"


def combine_partials_bytes(partials: bytes, num_windows: int = NUM_WINDOWS) -> bytes:
    """Host-only Horner + inversion over the windows' partial records (16 plain, 8 behind the GLV front
    end; msm377_g1_combine_window_partials; replaces submission.ts:290-321)."""
    if len(partials) != num_windows * WINDOW_PARTIAL_BYTES:
        raise ValueError("expected %d bytes of partials" % (num_windows * WINDOW_PARTIAL_BYTES))
    lib = load_library()
    src = (ctypes.c_uint32 * (len(partials) // 4)).from_buffer_copy(partials)
    out = ctypes.create_string_buffer(96)
    rc = lib.msm377_g1_combine_window_partials(ctypes.addressof(src), int(num_windows), ctypes.addressof(out))
    if rc:
        raise MsmError(rc, "msm377_g1_combine_window_partials")
    return out.raw


def add_points_bytes(points: bytes) -> bytes:
    """Sum of affine wire points (96 bytes each; the identity as x = 0, y = 1): the last step of a points-partitioned
    multi-GPU MSM (msm377_g1_add_points, host-only)."""
    if len(points) % 96:
        raise ValueError("points buffer length must be a multiple of 96")
    lib = load_library()
    out = ctypes.create_string_buffer(96)
    rc = lib.msm377_g1_add_points(bytes(points), len(points) // 96, ctypes.addressof(out))
    if rc:
        raise MsmError(rc, "msm377_g1_add_points")
    return out.raw


def combine_partials_split_bytes(partials: bytes, pieces: int) -> bytes:
    """The combine of 16 Edwards-form window records computed as the threaded host tail computes it -- ``pieces``
    balanced pieces of the Horner chain, added up -- on the calling thread (msm377_g1_combine_partials_split)."""
    if len(partials) != NUM_WINDOWS * WINDOW_PARTIAL_BYTES:
        raise ValueError("expected %d bytes of partials" % (NUM_WINDOWS * WINDOW_PARTIAL_BYTES))
    lib = load_library()
    src = (ctypes.c_uint32 * (len(partials) // 4)).from_buffer_copy(partials)
    out = ctypes.create_string_buffer(96)
    rc = lib.msm377_g1_combine_partials_split(ctypes.addressof(src), int(pieces), ctypes.addressof(out))
    if rc:
        raise MsmError(rc, "msm377_g1_combine_partials_split")
    return out.raw


def fold_partials_bytes(partials: bytes) -> bytes:
    """A rank's share of the host tail before the exchange (msm377_g1_fold_window_partials): the records of its
    consecutive windows are replaced by records of the same size and total, all identity but one point."""
    if len(partials) % WINDOW_PARTIAL_BYTES:
        raise ValueError("partials must be whole window records")
    count = len(partials) // WINDOW_PARTIAL_BYTES
    if count == 0:
        return partials
    lib = load_library()
    buf = (ctypes.c_uint32 * (len(partials) // 4)).from_buffer_copy(partials)
    rc = lib.msm377_g1_fold_window_partials(ctypes.addressof(buf), count)
    if rc:
        raise MsmError(rc, "msm377_g1_fold_window_partials")
    return bytes(buf)


def xyzz_to_affine(words: Sequence[int]) -> bytes:
    """One device-format XYZZ point (52 u32) -> 96-byte affine wire format."""
    lib = load_library()
    src = (ctypes.c_uint32 * POINT_WORDS)(*[int(w) for w in words])
    out = ctypes.create_string_buffer(96)
    rc = lib.msm377_g1_xyzz_to_affine(ctypes.addressof(src), ctypes.addressof(out))
    if rc:
        raise MsmError(rc, "msm377_g1_xyzz_to_affine")
    return out.raw


class MsmEngine:
    """One HIP device context: workspace for up to ``max_points`` inputs, reusable across
    calls (the reference re-acquires and destroys the device every call,
    submission.ts:113,288)."""

    def __init__(self, max_points: int, device: int = 0):
        self._lib = load_library()
        self._ctx = ctypes.c_void_p()
        rc = self._lib.msm377_ctx_create(int(device), int(max_points), ctypes.byref(self._ctx))
        if rc:
            self._ctx = ctypes.c_void_p()
            raise MsmError(rc, "msm377_ctx_create", "device %d, max_points %d (is a HIP device visible?)" % (device, max_points))
        self.max_points = int(max_points)
        self.device = int(device)

    # -- lifetime --
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.msm377_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc:
            raise MsmError(rc, what, self._lib.msm377_last_error(self._ctx).decode())

    # -- G1 MSM --
    def msm(self, points: bytes, scalars: bytes) -> bytes:
        """compute_msm on host buffers; returns x||y (96 bytes)."""
        n = _check_lengths(points, scalars)
        out = ctypes.create_string_buffer(96)
        self._check(self._lib.msm377_g1_msm(self._ctx, bytes(points), bytes(scalars), n, ctypes.addressof(out)), "msm377_g1_msm")
        return out.raw

    def msm_device(self, d_points: int, d_scalars: int, n: int) -> bytes:
        """Inputs already in HBM (raw device pointers, wire format)."""
        out = ctypes.create_string_buffer(96)
        self._check(self._lib.msm377_g1_msm_device(self._ctx, d_points, d_scalars, int(n), ctypes.addressof(out)), "msm377_g1_msm_device")
        return out.raw

    def set_bases(self, points: bytes):
        if len(points) % 96:
            raise ValueError("points buffer length must be a multiple of 96")
        self._check(self._lib.msm377_g1_set_bases(self._ctx, bytes(points), len(points) // 96), "msm377_g1_set_bases")

    def set_bases_device(self, d_points: int, n: int):
        self._check(self._lib.msm377_g1_set_bases_device(self._ctx, d_points, int(n)), "msm377_g1_set_bases_device")

    def set_bases_precomputed(self, points: bytes):
        """Resident bases WITH precomputed window multiples [2^(16 w)] P_i (msm377_g1_set_bases_precomputed): one bucket
        reduction and a 16-step tail per fixed-base MSM."""
        if len(points) % 96:
            raise ValueError("points buffer length must be a multiple of 96")
        self._check(self._lib.msm377_g1_set_bases_precomputed(self._ctx, bytes(points), len(points) // 96), "msm377_g1_set_bases_precomputed")

    def reserve_host_staging(self):
        """Allocate the pinned staging of the host-buffer entry points now instead of inside the first call
        (msm377_ctx_reserve_host_staging)."""
        self._check(self._lib.msm377_ctx_reserve_host_staging(self._ctx), "msm377_ctx_reserve_host_staging")

    def set_precompute_window(self, window_bits: int):
        """Window width of the next precomputed table: 16 (16 windows) or 20 (13 windows over one set of 2^19 buckets,
        msm377_ctx_set_precompute_window)."""
        self._check(self._lib.msm377_ctx_set_precompute_window(self._ctx, int(window_bits)), "msm377_ctx_set_precompute_window")

    def set_bases_precomputed_device(self, d_points: int, n: int):
        self._check(self._lib.msm377_g1_set_bases_precomputed_device(self._ctx, d_points, int(n)), "msm377_g1_set_bases_precomputed_device")

    def msm_fixed_base(self, scalars: bytes) -> bytes:
        if len(scalars) % 32:
            raise ValueError("scalars buffer length must be a multiple of 32")
        out = ctypes.create_string_buffer(96)
        self._check(self._lib.msm377_g1_msm_fixed_base(self._ctx, bytes(scalars), len(scalars) // 32, ctypes.addressof(out)), "msm377_g1_msm_fixed_base")
        return out.raw

    def msm_fixed_base_device(self, d_scalars: int, n: int) -> bytes:
        out = ctypes.create_string_buffer(96)
        self._check(self._lib.msm377_g1_msm_fixed_base_device(self._ctx, d_scalars, int(n), ctypes.addressof(out)), "msm377_g1_msm_fixed_base_device")
        return out.raw

    def msm_fixed_base_batch_device(self, d_scalars: int, n: int, batch: int) -> List[bytes]:
        """`batch` MSMs of n scalars each (contiguous in HBM) against the resident bases; the host tail
        of one MSM overlaps the GPU work of the next."""
        out = ctypes.create_string_buffer(96 * max(1, batch))
        self._check(
            self._lib.msm377_g1_msm_fixed_base_batch_device(self._ctx, d_scalars, int(n), int(batch), ctypes.addressof(out)),
            "msm377_g1_msm_fixed_base_batch_device",
        )
        return [out.raw[96 * b : 96 * b + 96] for b in range(batch)]

    def window_partials_device(self, d_points: int, d_scalars: int, n: int, win_begin: int, win_count: int) -> bytes:
        """Partial records of windows [win_begin, win_begin + win_count) (multi-GPU sharding)."""
        out = ctypes.create_string_buffer(max(1, win_count) * WINDOW_PARTIAL_BYTES)
        self._check(
            self._lib.msm377_g1_window_partials_device(self._ctx, d_points, d_scalars, int(n), int(win_begin), int(win_count), ctypes.addressof(out)),
            "msm377_g1_window_partials_device",
        )
        return out.raw[: win_count * WINDOW_PARTIAL_BYTES]

    def window_partials_resident(self, d_points: int, d_scalars: int, n: int, win_begin: int, win_count: int, d_partials_out: int):
        """The same records left in DEVICE memory at d_partials_out (win_count x WINDOW_PARTIAL_BYTES): the multi-GPU
        exchange reads them there (msm377_g1_window_partials_resident)."""
        self._check(
            self._lib.msm377_g1_window_partials_resident(self._ctx, d_points, d_scalars, int(n), int(win_begin), int(win_count), d_partials_out),
            "msm377_g1_window_partials_resident",
        )

    def combine_partials(self, partials) -> bytes:
        """Host tail over all 16 windows' records on the context's tail threads (msm377_g1_combine_partials_ctx);
        ``partials`` is bytes or anything with a buffer address of 16 x WINDOW_PARTIAL_BYTES bytes.  Raises
        MsmError(EEXCEPTIONAL) when Edwards records add up to an exceptional case (recompute in form 0)."""
        if isinstance(partials, (bytes, bytearray)):
            if len(partials) != NUM_WINDOWS * WINDOW_PARTIAL_BYTES:
                raise ValueError("expected %d bytes of partials" % (NUM_WINDOWS * WINDOW_PARTIAL_BYTES))
            src = (ctypes.c_uint32 * (len(partials) // 4)).from_buffer_copy(partials)
            addr = ctypes.addressof(src)
        else:
            addr = int(partials)
        out = ctypes.create_string_buffer(96)
        self._check(self._lib.msm377_g1_combine_partials_ctx(self._ctx, addr, ctypes.addressof(out)), "msm377_g1_combine_partials_ctx")
        return out.raw

    def fallback_info(self) -> Tuple[int, int]:
        """(count, last_mask): reruns on the Weierstrass path after an exceptional case of the Edwards law, and the
        FB_* bits of the last one."""
        count, mask = ctypes.c_uint64(), ctypes.c_uint32()
        self._check(self._lib.msm377_ctx_get_fallback_info(self._ctx, ctypes.byref(count), ctypes.byref(mask)), "msm377_ctx_get_fallback_info")
        return int(count.value), int(mask.value)

    def glv_window_partials_device(self, d_points: int, d_scalars: int, n: int, win_begin: int, win_count: int) -> bytes:
        """Same behind the GLV front end (8 windows); raises MsmError(EGLVRANGE) for out-of-range scalars."""
        out = ctypes.create_string_buffer(max(1, win_count) * WINDOW_PARTIAL_BYTES)
        self._check(
            self._lib.msm377_g1_glv_window_partials_device(self._ctx, d_points, d_scalars, int(n), int(win_begin), int(win_count), ctypes.addressof(out)),
            "msm377_g1_glv_window_partials_device",
        )
        return out.raw[: win_count * WINDOW_PARTIAL_BYTES]

    def generate_bases_device(self, seed: int, n: int, d_points_out: int):
        self._check(self._lib.msm377_g1_generate_bases_device(self._ctx, int(seed) & (2**64 - 1), int(n), d_points_out), "msm377_g1_generate_bases_device")

    # -- Twisted-Edwards BLS12 (BASELINE.json config 3): 64-byte points, 64-byte result --
    def ed_msm(self, points: bytes, scalars: bytes) -> bytes:
        if len(scalars) % 32 or len(points) != 2 * len(scalars):
            raise ValueError("Edwards points buffer must hold 64 bytes and scalars 32 bytes per input")
        out = ctypes.create_string_buffer(64)
        self._check(self._lib.msm377_ed_msm(self._ctx, bytes(points), bytes(scalars), len(scalars) // 32, ctypes.addressof(out)), "msm377_ed_msm")
        return out.raw

    def ed_msm_device(self, d_points: int, d_scalars: int, n: int) -> bytes:
        out = ctypes.create_string_buffer(64)
        self._check(self._lib.msm377_ed_msm_device(self._ctx, d_points, d_scalars, int(n), ctypes.addressof(out)), "msm377_ed_msm_device")
        return out.raw

    def ed_generate_bases_device(self, seed: int, n: int, d_points_out: int):
        self._check(self._lib.msm377_ed_generate_bases_device(self._ctx, int(seed) & (2**64 - 1), int(n), d_points_out), "msm377_ed_generate_bases_device")

    # -- stage access (the reference's debug=true read-backs) --
    def set_stage_capture(self, enabled: bool = True):
        self._check(self._lib.msm377_ctx_set_stage_capture(self._ctx, int(bool(enabled))), "msm377_ctx_set_stage_capture")

    def read_stage(self, slot: int, n: int, want=("digits", "row_ptr", "val_idx", "buckets")):
        """Returns a dict of numpy arrays for window slot ``slot`` of the last call."""
        import numpy as np

        res = {}
        digits = np.empty(n, dtype=np.uint16) if "digits" in want else None
        row_ptr = np.empty(NUM_BUCKETS + 2, dtype=np.uint32) if "row_ptr" in want else None
        val_idx = np.empty(n, dtype=np.uint32) if "val_idx" in want else None
        buckets = np.empty((NUM_BUCKETS, POINT_WORDS), dtype=np.uint32) if "buckets" in want else None

        def ptr(a):
            return a.ctypes.data if a is not None else None

        self._check(self._lib.msm377_g1_read_stage(self._ctx, int(slot), ptr(digits), ptr(row_ptr), ptr(val_idx), ptr(buckets)), "msm377_g1_read_stage")
        for k, v in (("digits", digits), ("row_ptr", row_ptr), ("val_idx", val_idx), ("buckets", buckets)):
            if v is not None:
                res[k] = v
        return res

    def stage_form(self) -> int:
        """Coordinate system of the captured buckets: 0 = Weierstrass XYZZ, 1 = twisted Edwards (X, Y, T, Z), -1 = none."""
        return int(self._lib.msm377_ctx_get_stage_form(self._ctx))

    def set_narrow_max(self, max_points: int = 1 << 16):
        """Inputs of at most this many points run with narrow windows of 2^11 buckets (msm377_ctx_set_narrow_max; 0 = never)."""
        self._check(self._lib.msm377_ctx_set_narrow_max(self._ctx, int(max_points)), "msm377_ctx_set_narrow_max")

    def set_g1_form(self, form="edwards"):
        """Internal coordinates of the G1 full-MSM entry points: "edwards" / 1 (default, csrc/te377.hpp) or
        "weierstrass" / 0 (XYZZ behind the GLV front end)."""
        f = {"edwards": 1, "weierstrass": 0}.get(form, form)
        self._check(self._lib.msm377_ctx_set_g1_form(self._ctx, int(f)), "msm377_ctx_set_g1_form")

    def set_glv(self, mode="auto"):
        """GLV front end of the Weierstrass form: True/1 = on (the caller vouches that every point lies in the
        prime-order subgroup), False/0 or "auto"/2 = off (the default)."""
        m = 2 if mode == "auto" else int(mode)
        self._check(self._lib.msm377_ctx_set_glv(self._ctx, m), "msm377_ctx_set_glv")

    # -- measurement --
    def set_timing(self, enabled=True):
        """True / 1: HIP events around every stage; 2: around the accumulation kernel only (the other stages read 0);
        False / 0: off.  The events cost GPU idle time between the launches (~50 us per MSM with every stage on)."""
        self._check(self._lib.msm377_ctx_set_timing(self._ctx, 2 if enabled == 2 else int(bool(enabled))), "msm377_ctx_set_timing")

    def stage_ms(self) -> dict:
        arr = (ctypes.c_double * len(STAGE_NAMES))()
        self._check(self._lib.msm377_ctx_get_stage_ms(self._ctx, ctypes.addressof(arr)), "msm377_ctx_get_stage_ms")
        return dict(zip(STAGE_NAMES, list(arr)))


    def accumulate_products(self) -> int:
        """Field products per bucket addition of the last accumulation launch (10 / 8 / 7)."""
        return int(self._lib.msm377_ctx_get_products_per_addition(self._ctx))


def _check_lengths(points: bytes, scalars: bytes) -> int:
    """input_size = scalars.length / 32 (submission.ts:91)."""
    if len(scalars) % 32:
        raise ValueError("scalars buffer length must be a multiple of 32")
    n = len(scalars) // 32
    if len(points) != 96 * n:
        raise ValueError("points buffer must hold %d bytes (96 per scalar), got %d" % (96 * n, len(points)))
    return n
