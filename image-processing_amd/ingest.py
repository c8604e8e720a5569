"""File ingest for the extractors: threaded decode, pinned staging, overlapped
host-to-device copies, one batch of lookahead.

The reference reads one file per pool task (load_image, reference
src/features/extract_2d_hu.py:84-98; load_sample,
src/features/extract_3d_pca_features.py:48-56) and computes on the same core.
Here the device finishes a batch in microseconds, so the host side is what
sets files per second: decoding runs on a thread pool (PIL and NumPy release
the GIL while they inflate / read), decoded arrays are packed into page-locked
memory and copied on a side stream, and the next batch is decoded while the
current one is on the device.
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, Iterable, Iterator, List, Sequence, Tuple

import numpy as np


def default_workers() -> int:
    """Decode threads: the host cores, at most 16 (one GPU's share of a node)."""
    env = os.environ.get("FX_DECODE_WORKERS")
    if env:
        return max(1, int(env))
    return max(1, min(16, os.cpu_count() or 1))


def map_ordered(fn: Callable[[Any], Any], items: Sequence[Any], workers: int) -> List[Tuple[bool, Any]]:
    """fn over items on a thread pool; one (True, value) or (False, exception)
    per item, in item order.  An exception in one item never hides another."""
    def guarded(x):
        try:
            return True, fn(x)
        except Exception as e:  # the caller logs it per file
            return False, e

    items = list(items)
    if workers <= 1 or len(items) <= 1:
        return [guarded(x) for x in items]
    with ThreadPoolExecutor(max_workers=min(workers, len(items))) as pool:
        return list(pool.map(guarded, items))


def as_device_points(arr: np.ndarray, dims: int = 3) -> np.ndarray:
    """The first `dims` columns of a point array in the narrowest type that holds
    them exactly: float32 if every value survives the round trip (then the
    float32 kernels compute what the reference computes from the widened
    values), float64 otherwise."""
    a = np.asarray(arr)[:, :dims]
    if a.dtype == np.float32:
        return np.ascontiguousarray(a)
    wide = np.ascontiguousarray(a, dtype=np.float64)
    narrow = wide.astype(np.float32)
    same = (narrow.astype(np.float64) == wide) | np.isnan(wide)
    return narrow if bool(same.all()) else wide


def read_npy_points(path) -> np.ndarray:
    """First three columns of an [N, >=3] .npy file as [N, 3] float32, or
    float64 where float32 would round (load_sample, reference :48-56, hands
    over the float64 values); ValueError on any other shape."""
    arr = np.load(path, allow_pickle=False, mmap_mode=None)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError(f"invalid shape {arr.shape} (expected Nx3) for {path}")
    return as_device_points(arr, 3)


def _path_array(paths: Sequence[str]):
    import ctypes
    arr = (ctypes.c_char_p * len(paths))()
    arr[:] = [os.fsencode(p) for p in paths]
    return arr


def host_buffer(shape, dtype=np.float32):
    """A host array for staging: page-locked when a HIP device is present (so
    that the later copy is asynchronous), ordinary memory otherwise.  Returns
    (numpy view, owner to keep alive / to copy from)."""
    import torch
    tdtype = torch.from_numpy(np.empty(0, dtype=dtype)).dtype
    t = torch.empty(tuple(shape), dtype=tdtype, pin_memory=torch.cuda.is_available())
    return t.numpy(), t


def read_points_batch(paths: Sequence[str], rows: int, out: np.ndarray, workers: int):
    """Native batch read of [rows, >=3] float32/float64 .npy files into
    out[i] (float32 [n, rows, 3]).  Returns (status int32 [n], rows_found int64
    [n]); status 0 = read, anything else = left to the caller (float64 files
    whose values float32 would round among them)."""
    from . import _lib
    n = len(paths)
    status = np.zeros(n, dtype=np.int32)
    found = np.zeros(n, dtype=np.int64)
    if n == 0:
        return status, found
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (n, rows, 3)
    _lib.call("fx_read_npy_points_f32", _path_array(paths), n, int(rows), out.ctypes.data, status.ctypes.data,
              found.ctypes.data, int(workers))
    return status, found


def write_rows_batch(paths: Sequence[str], data: np.ndarray, workers: int) -> np.ndarray:
    """Native batch np.save of the float32 rows of `data` to `paths`; returns
    the per-file status (0 = written)."""
    from . import _lib
    n = len(paths)
    status = np.zeros(n, dtype=np.int32)
    if n == 0:
        return status
    data = np.ascontiguousarray(data, dtype=np.float32)
    assert data.ndim == 2 and data.shape[0] == n
    _lib.call("fx_write_npy_rows_f32", _path_array(paths), n, data.ctypes.data, int(data.shape[1]),
              status.ctypes.data, int(workers))
    return status


def decode_png_batch(paths: Sequence[str], H: int, W: int, out: np.ndarray, workers: int):
    """Native batch decode of H x W PNG files to gray into out[i] (uint8
    [n, H, W]).  Returns (status int32 [n], found int32 [n, 2] = height, width
    as far as the header could be read); status 0 = decoded, anything else =
    left to PIL."""
    from . import _lib
    n = len(paths)
    status = np.zeros(n, dtype=np.int32)
    found = np.full((n, 2), -1, dtype=np.int32)
    if n == 0:
        return status, found
    assert out.dtype == np.uint8 and out.flags.c_contiguous and out.shape == (n, H, W)
    _lib.call("fx_decode_png_gray_u8", _path_array(paths), n, int(H), int(W), out.ctypes.data, status.ctypes.data,
              found.ctypes.data, int(workers))
    return status, found


# Page-locked staging blocks are kept and reused: locking host memory costs about as much per megabyte as reading
# the files does.  A block is taken for a batch and given back with an event that tells when the copy out of it is done.
_staging_blocks: list = []
_staging_lock = threading.Lock()


def take_staging(nbytes: int) -> dict:
    import torch
    with _staging_lock:
        for e in _staging_blocks:
            if not e["busy"] and e["t"].numel() >= nbytes and (e["event"] is None or e["event"].query()):
                e["busy"] = True
                return e
        cap = max(int(nbytes * 1.25), 32 << 20)
        e = {"t": torch.empty(cap, dtype=torch.uint8, pin_memory=torch.cuda.is_available()), "busy": True, "event": None}
        if len(_staging_blocks) < 4:
            _staging_blocks.append(e)
        return e


def release_staging(entry: dict, event=None) -> None:
    with _staging_lock:
        entry["event"] = event
        entry["busy"] = False


def stage_png_batch(paths: Sequence[str], H: int, W: int, workers: int) -> dict:
    """First half of the device decode of H x W PNG files: host threads read the files, check the container (as
    decode_png_batch does) and copy every file's zlib stream into one page-locked block.  Returns {"staging"
    (uint8 tensor, host), "used" (bytes of it written), "offset" int64 [n], "length" int32 [n], "channels" int32 [n],
    "status" int32 [n] (0 = staged; anything else = left to the host decoders), "found" int32 [n, 2], "block" (what
    release_staging takes once the bytes have been copied out)}; the second half is engine.png_inflate_batch."""
    import ctypes
    import torch
    from . import _lib
    n = len(paths)
    out = {"offset": np.zeros(n, dtype=np.int64), "length": np.zeros(n, dtype=np.int32),
           "channels": np.zeros(n, dtype=np.int32), "status": np.zeros(n, dtype=np.int32),
           "found": np.full((n, 2), -1, dtype=np.int32), "used": 0,
           "staging": torch.empty(0, dtype=torch.uint8), "block": None}
    if n == 0:
        return out
    arr = _path_array(paths)
    cap = int(_lib.load().fx_png_stage_bytes(arr, n, int(workers)))
    if cap < 0:
        raise ValueError("fx_png_stage_bytes: bad arguments")
    block = take_staging(cap)
    staging = block["t"]
    used = ctypes.c_int64(0)
    _lib.call("fx_png_stage", arr, n, int(H), int(W), staging.data_ptr(), cap, out["offset"].ctypes.data,
              out["length"].ctypes.data, out["channels"].ctypes.data, out["status"].ctypes.data, out["found"].ctypes.data,
              ctypes.addressof(used), int(workers))
    out["staging"] = staging
    out["used"] = int(used.value)
    out["block"] = block
    return out


def image_size(path):
    """(height, width) from the file header, or None."""
    try:
        from PIL import Image
        with Image.open(str(path)) as im:
            return int(im.height), int(im.width)
    except Exception:
        return None


def npy_rows(path) -> int:
    """Row count from the header of a 2-D .npy file (no data read); -1 if it cannot be told."""
    try:
        arr = np.load(path, allow_pickle=False, mmap_mode="r")
        return int(arr.shape[0]) if arr.ndim == 2 else -1
    except Exception:
        return -1


def stack_to_device(arrays: Sequence[np.ndarray], device, workers: int = 1):
    """Equally shaped arrays -> one device tensor [n, ...] through page-locked
    memory and an asynchronous copy on a side stream; the current stream is
    made to wait for the copy, the host is not."""
    import torch

    first = np.asarray(arrays[0])
    staging = torch.empty((len(arrays),) + first.shape, dtype=torch.from_numpy(np.empty(0, dtype=first.dtype)).dtype,
                          pin_memory=True)
    view = staging.numpy()

    def put(i):
        view[i] = arrays[i]

    if workers > 1 and len(arrays) * first.nbytes > (8 << 20):
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(put, range(len(arrays))))
    else:
        for i in range(len(arrays)):
            put(i)
    side = _copy_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        out = staging.to(device, non_blocking=True)
    torch.cuda.current_stream(device).wait_stream(side)
    # both buffers must outlive the copy: tie them to the consuming stream
    out.record_stream(torch.cuda.current_stream(device))
    _keep_alive(staging, side)
    return out


_streams = {}
_pending: List[Tuple[Any, Any]] = []
_lock = threading.Lock()


def _copy_stream(device):
    import torch
    key = (device.index if device.index is not None else torch.cuda.current_device())
    with _lock:
        if key not in _streams:
            _streams[key] = torch.cuda.Stream(device=device)
        return _streams[key]


def _keep_alive(staging, stream) -> None:
    """Holds the pinned buffer until the copy that reads it has finished."""
    import torch
    ev = torch.cuda.Event()
    ev.record(stream)
    with _lock:
        _pending.append((ev, staging))
        _pending[:] = [(e, s) for e, s in _pending if not e.query()]


def prefetch(produce: Callable[[Any], Any], batches: Iterable[Any]) -> Iterator[Any]:
    """Yields produce(batch) for every batch in order while produce() of the
    next batch already runs on a background thread (one batch of lookahead).
    An exception raised by produce() is re-raised at the position of its batch."""
    it = iter(batches)
    with ThreadPoolExecutor(max_workers=1) as pool:
        try:
            fut = pool.submit(produce, next(it))
        except StopIteration:
            return
        for nxt in it:
            nfut = pool.submit(produce, nxt)
            yield fut.result()
            fut = nfut
        yield fut.result()
