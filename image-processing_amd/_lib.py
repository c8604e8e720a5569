"""ctypes binding of libfx.so (the C ABI declared in include/fx.h).

There is no CPU fallback: if the shared library is missing or a call fails,
an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("FX_LIB", _HERE / "libfx.so"))

FX_OK = 0

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64

# name -> (restype, argtypes); every symbol include/fx.h declares.
SIGNATURES = {
    "fx_init": (c_int, [c_int]),
    "fx_shutdown": (c_int, []),
    "fx_last_error": (ctypes.c_char_p, []),
    "fx_abi_version": (c_int, []),
    "fx_hu_batch_u8": (c_int, [c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_int, c_int,
                               c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_hu_batch_u8_noalloc": (c_int, [c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_release_stream": (c_int, [c_void_p]),
    "fx_hu_partial_u8": (c_int, [c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_int, c_int, c_i64, c_void_p, c_void_p]),
    "fx_hu_from_partial": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_hu_groupstat": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "fx_pca_batch_f32": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "fx_pca_batch_f64": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "fx_silhouette_batch_u8": (c_int, [c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_int, ctypes.c_double, c_int,
                                       c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_silhouette_moments_batch_u8": (c_int, [c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_int, ctypes.c_double,
                                               c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_gray_fit_batch_u8": (c_int, [c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_void_p, c_int, c_int, c_void_p,
                                     c_void_p]),
    "fx_center_scale": (c_int, [c_void_p, c_int, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_fps_f32": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_allgather_features": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_read_npy_points_f32": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_int]),
    "fx_write_npy_rows_f32": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int]),
    "fx_decode_png_gray_u8": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "fx_png_stage_bytes": (c_i64, [c_void_p, c_i64, c_int]),
    "fx_png_stage": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_int]),
    "fx_png_inflate_gray_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p]),
    "fx_column_thresholds_f64": (c_int, [c_void_p, c_i64, c_int, c_i64, c_void_p, c_int, c_int, ctypes.c_double, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "fx_binarize_ge_f64": (c_int, [c_void_p, c_i64, c_int, c_i64, c_void_p, c_void_p, c_void_p]),
    "fx_cloud_bounds_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "fx_camera_table_f64": (c_int, [c_void_p, c_i64, c_int, c_void_p, ctypes.c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "fx_render_views_u8": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p]),
    "fx_synth_shapes_u8": (c_int, [c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_u64, c_i64, c_void_p]),
    "fx_synth_points_f32": (c_int, [c_void_p, c_i64, c_i64, c_u64, c_i64, c_void_p]),
    "fx_profile": (c_int, [c_int]),
    "fx_profile_last_ms": (ctypes.c_double, []),
    "fx_profile_ms_back": (ctypes.c_double, [c_int]),
    "fx_profile_last_units": (ctypes.c_longlong, []),
}


class FxError(RuntimeError):
    """A libfx call returned a non-zero status."""

    def __init__(self, fn: str, code: int, message: str):
        super().__init__(f"{fn} failed with {code}: {message}")
        self.fn, self.code, self.message = fn, code, message


_lib = None


def load() -> ctypes.CDLL:
    """Loads libfx.so once and attaches the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C {_HERE / 'csrc'}` "
            "(or __graft_entry__.build()); this package has no CPU fallback")
    # torch brings its own HIP runtime: it has to be in the process before libfx.so is, so that both
    # resolve to the same one (with the system's runtime loaded first, libfx then finds no device)
    import torch  # noqa: F401
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name: str, *args):
    """Calls an int-returning entry point and raises FxError on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != FX_OK:
        msg = lib.fx_last_error()
        raise FxError(name, rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
