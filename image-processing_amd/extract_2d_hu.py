"""Hu-moment feature extraction on MI355X behind the reference's function names.

Mirrors reference src/features/extract_2d_hu.py: same function names, argument
meaning, return shapes, CSV/.npy layout and error policy.  The arithmetic of
compute_hu_from_image (:443-478: cv2.moments, cv2.HuMoments, log transform) and
of the per-sample mean/std (:641-644) runs in the HIP kernels of libfx
(csrc/fx_hu.hip); there is no CPU fallback for it.

The silhouette preprocessing in front of it (:101-440: blur, Otsu, morphology,
bounding-box crop, nearest resize, mask cleaning) runs on the device as well
(csrc/fx_sil.hip).  The reference has two code paths there, chosen by whether
OpenCV is installed; both are built.  SILHOUETTE_BACKEND = "opencv" (the
default: the path its production run took, cv2.GaussianBlur / THRESH_OTSU /
5x5 ellipse morphology / largest external contour / 8-connected component
filter) or "pil" (PIL + scipy.ndimage); the `silhouette_backend` key of a
task's config, the `backend` argument of the functions below and the
--silhouette_backend flag override it.  Only the image decoding is a PIL call
on the host.

Added next to the reference API: `compute_hu_batch` and `process_samples`,
which push many views through one kernel launch.
"""
from __future__ import annotations

import argparse
import datetime
import logging
import os
import sys
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image, ImageOps

logger = logging.getLogger("extract_2d_hu")

SILHOUETTE_BACKEND = "opencv"   # which branch of the reference's preprocessing is reproduced: "opencv" or "pil"

IMAGE_SUFFIXES = (".png", ".jpg", ".jpeg")
HU_COLS = ["id"] + [f"hu{i + 1}_mean" for i in range(7)] + [f"hu{i + 1}_std" for i in range(7)] + ["note"]


def ensure_dir(p: Path) -> None:
    p.mkdir(parents=True, exist_ok=True)


def list_sample_ids(views_dir: Path) -> List[Tuple[str, Path]]:
    """(id, dir) of every sub-directory holding images, sorted by path (reference :66-81)."""
    found = []
    if not views_dir.exists():
        return found
    root = str(views_dir)
    with os.scandir(root) as it:
        dirs = sorted(os.path.join(root, e.name) for e in it if e.is_dir())   # (the order of str(path), as the reference's)
    for d in dirs:
        with os.scandir(d) as files:
            if any(os.path.splitext(f.name)[1].lower() in IMAGE_SUFFIXES for f in files):
                found.append((os.path.basename(d), Path(d)))
    return found


def list_view_files(dir_path: Path) -> List[Path]:
    return sorted((p for p in dir_path.glob("*") if p.suffix.lower() in IMAGE_SUFFIXES), key=lambda p: str(p))


def load_image(path: Path) -> np.ndarray:
    """Grayscale uint8 array of an image file (reference :84-98, PIL branch)."""
    return np.asarray(Image.open(str(path)).convert("L"), dtype=np.uint8)


# --------------------------------------------------------------------------
# device path
# --------------------------------------------------------------------------
def _device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the Hu extractor runs on the GPU only")
    return torch.device("cuda", torch.cuda.current_device())


def pad_and_resize(img: np.ndarray, target: int, bg_color: str) -> np.ndarray:
    """Aspect-preserving bicubic fit of the gray image into target x target,
    centred on the background colour (reference :224-240, PIL
    ImageOps.contain + paste), computed on the device."""
    import torch
    from . import engine

    arr = np.ascontiguousarray(img, dtype=np.uint8)
    out = engine.gray_fit_batch(torch.from_numpy(arr[None]).to(_device()), None, target,
                                255 if bg_color == "white" else 0)
    return out[0].cpu().numpy()


def compute_silhouettes(images: np.ndarray, resolution: int, crop_margin: float, preprocess_crop: bool,
                        backend: Optional[str] = None):
    """Cleaned 0/255 silhouettes of equally sized gray views [B, H, W] in one
    device call: (uint8 [B, res, res], int32 [B, 8] = Otsu level, inverted
    flag, crop x0 y0 x1 y1, resized width and height).  Covers compute_otsu,
    find_largest_bbox, the mask half of preprocess_image and clean_mask_np of
    the reference (:129-440) on the chosen branch."""
    import torch
    from . import engine

    arr = np.ascontiguousarray(images, dtype=np.uint8)
    if arr.ndim == 2:
        arr = arr[None]
    out = engine.silhouette_batch(torch.from_numpy(arr).to(_device()), resolution, crop_margin, preprocess_crop,
                                  info=True, backend=backend or SILHOUETTE_BACKEND)
    return out["mask"].cpu().numpy(), out["info"].cpu().numpy()


def preprocess_image(img: np.ndarray, resolution: int, crop_margin: float, bg_color: str,
                     preprocess_crop: bool, backend: Optional[str] = None) -> Tuple[np.ndarray, np.ndarray]:
    """(preprocessed gray image, cleaned 0/255 silhouette), both
    resolution x resolution uint8 (reference :352-440), both from the device."""
    import torch
    from . import engine

    assert img.ndim == 2
    d = torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint8)[None]).to(_device())
    sil = engine.silhouette_batch(d, resolution, crop_margin, preprocess_crop, info=True,
                                  backend=backend or SILHOUETTE_BACKEND)
    pre = engine.gray_fit_batch(d, sil["info"], resolution, 255 if bg_color == "white" else 0)
    return pre[0].cpu().numpy(), sil["mask"][0].cpu().numpy()


def compute_hu_batch(images: Sequence[np.ndarray] | np.ndarray, thresh: int = -1, on_value: int = 255,
                     group_sizes: Optional[Sequence[int]] = None):
    """float32 [B, 7] log-Hu vectors of equally sized uint8 images in one
    launch.  With group_sizes (views per sample, summing to B) also returns the
    float32 [G, 14] mean || std table computed on the device."""
    import torch
    from . import engine

    arr = np.ascontiguousarray(np.stack([np.asarray(i).astype(np.uint8) for i in images])
                               if not isinstance(images, np.ndarray) else images.astype(np.uint8))
    if arr.ndim == 2:
        arr = arr[None]
    dev = _device()
    d_img = torch.from_numpy(arr).to(dev)
    res = engine.hu_batch(d_img, thresh=thresh, on_value=on_value)
    if group_sizes is None:
        return res["hulog7"].cpu().numpy()
    off = np.zeros(len(group_sizes) + 1, dtype=np.int64)
    np.cumsum(np.asarray(group_sizes, dtype=np.int64), out=off[1:])
    if off[-1] != arr.shape[0]:
        raise ValueError("group_sizes must sum to the number of images")
    stats = engine.hu_groupstat(res["hulog7"], torch.from_numpy(off).to(dev))
    return res["hulog7"].cpu().numpy(), stats.cpu().numpy()


def compute_hu_from_image(img_or_mask: np.ndarray) -> np.ndarray:
    """float32 [7]: -sign(h) * log10(|h| + 1e-300) of the seven Hu invariants
    of a grayscale image or 0/255 mask, pixel value = weight (reference :443-478)."""
    return compute_hu_batch(np.asarray(img_or_mask).astype(np.uint8)[None])[0]


def hu_from_clouds_device(points, offsets, views: int = 12, resolution: int = 224, point_size: float = 2.0,
                          crop_margin: float = 0.05, preprocess_crop: bool = False, use_silhouette: bool = False,
                          bg_color: str = "white", camera_radius_factor: float = 2.0, seed: int = 42,
                          return_views: bool = False, backend: Optional[str] = None):
    """hu_from_clouds for clouds already in HBM (float32 [total, 3], int64
    offsets [B + 1]); returns a float32 device tensor [B, 14]."""
    import torch
    from . import engine, render_views
    rendered = render_views.render_views_device(points, offsets, views, resolution, point_size, bg_color,
                                                camera_radius_factor, seed)
    B = rendered.shape[0]
    raw = rendered.reshape(B * views, resolution, resolution)
    if use_silhouette:
        # Hu of the silhouettes: moments summed while a silhouette is on chip, the silhouettes are not written
        rows = engine.silhouette_hu_batch(raw, resolution, crop_margin, preprocess_crop,
                                          backend=backend or SILHOUETTE_BACKEND)["hulog7"]
    else:
        if preprocess_crop:
            sil = engine.silhouette_batch(raw, resolution, crop_margin, preprocess_crop, info=True,
                                          backend=backend or SILHOUETTE_BACKEND)
            target = engine.gray_fit_batch(raw, sil["info"], resolution, 255 if bg_color == "white" else 0)
        else:
            # the gray view itself is measured: the reference computes a silhouette here too and drops it (:372-397, :626)
            target = engine.gray_fit_batch(raw, None, resolution, 255 if bg_color == "white" else 0)
        rows = engine.hu_batch(target)["hulog7"]
    off = torch.arange(B + 1, dtype=torch.int64, device=points.device) * views
    out = engine.hu_groupstat(rows, off)
    return (out, rendered) if return_views else out


def hu_from_clouds(clouds: Sequence[np.ndarray], views: int = 12, resolution: int = 224, point_size: float = 2.0,
                   crop_margin: float = 0.05, preprocess_crop: bool = False, use_silhouette: bool = False,
                   bg_color: str = "white", camera_radius_factor: float = 2.0, seed: int = 42,
                   return_views: bool = False, backend: Optional[str] = None):
    """Sampled clouds -> rendered views -> preprocessing -> log-Hu -> per-sample
    mean / std, float32 [B, 14], without the views leaving the device: the
    chain of the reference's render_views.py (its matplotlib renderer) and
    extract_2d_hu.py (:602-667) with the PNG files in between left out (a PNG
    of a gray view decodes to the same bytes)."""
    from . import render_views
    for c in clouds:
        if c.shape[0] < 1:
            raise ValueError("empty point cloud")
    pts, off = render_views.pack_clouds(clouds, _device())
    res = hu_from_clouds_device(pts, off, views, resolution, point_size, crop_margin, preprocess_crop, use_silhouette,
                                bg_color, camera_radius_factor, seed, return_views, backend)
    if return_views:
        return res[0].cpu().numpy().astype(np.float32), res[1]
    return res.cpu().numpy().astype(np.float32)


def save_descriptor_npy(sample_id: str, vec: np.ndarray, out_dir: Path) -> Path:
    """<id>_hu.npy, float32 (reference :481-488)."""
    ensure_dir(out_dir)
    p = out_dir / f"{sample_id}_hu.npy"
    np.save(str(p), vec.astype(np.float32), allow_pickle=False)
    return p


def append_to_csv(records: List[Dict[str, Any]], out_csv_path: Path, overwrite: bool = False) -> None:
    """Writes records with the fixed column order id, hu1..7_mean, hu1..7_std,
    note; when the file exists and overwrite is False the new rows are appended
    and the last row per id wins (reference :491-507)."""
    import pandas as pd
    df_new = pd.DataFrame(records, columns=HU_COLS)
    if out_csv_path.exists() and not overwrite:
        try:
            both = pd.concat([pd.read_csv(out_csv_path), df_new], ignore_index=True)
            both.drop_duplicates(subset=["id"], keep="last").to_csv(out_csv_path, index=False)
            return
        except Exception as e:
            logger.debug(f"Failed to append to existing CSV; will overwrite. Error: {e}")
    df_new.to_csv(out_csv_path, index=False)


def _fit_panel(im, resolution: int, bg_rgb, resample):
    """`im` as an RGB panel of resolution x resolution."""
    if im.size != (resolution, resolution):
        im = im.resize((resolution, resolution), resample=resample)
    return im.convert("RGB")


def save_preview(id: str, chosen_view_paths: List[Path], preproc_img: np.ndarray, mask: np.ndarray,
                 out_preview_path: Path, bg_color: str, resolution: int) -> None:
    """The three-part collage of a sample (reference :510-596): up to three raw views, each fitted into a
    resolution x resolution square on the background colour, then the preprocessed image (bicubic if it has
    another size), then the silhouette (nearest neighbour, re-thresholded at 128), side by side as one RGB PNG.
    A part that cannot be produced is a blank square."""
    bg_rgb = (255, 255, 255) if bg_color == "white" else (0, 0, 0)
    side = (resolution, resolution)

    def blank():
        return Image.new("RGB", side, color=bg_rgb)

    panels = []
    for path in chosen_view_paths[:3]:
        try:
            thumb = ImageOps.contain(Image.open(path).convert("RGB"), side)
            square = blank()
            square.paste(thumb, ((resolution - thumb.width) // 2, (resolution - thumb.height) // 2))
            panels.append(square)
        except Exception:
            panels.append(blank())
    if not panels:
        panels.append(blank())
    try:
        panels.append(_fit_panel(Image.fromarray(preproc_img).convert("L"), resolution, bg_rgb, Image.BICUBIC))
    except Exception:
        panels.append(blank())
    try:
        sil = Image.fromarray(mask).convert("L")
        if sil.size != side:
            sil = sil.resize(side, resample=Image.NEAREST)
        panels.append(sil.point(lambda v: 255 if v > 128 else 0).convert("L").convert("RGB"))
    except Exception:
        panels.append(blank())
    # every panel is resolution high, so the row needs no vertical padding
    collage = Image.new("RGB", (sum(im.width for im in panels), resolution), color=bg_rgb)
    x = 0
    for im in panels:
        collage.paste(im, (x, 0))
        x += im.width
    ensure_dir(out_preview_path.parent)
    collage.save(str(out_preview_path), format="PNG")


def write_previews(results: List[Dict[str, Any]], views_dir: Path, reports_dir: Path, n_preview: int, resolution: int,
                   crop_margin: float, bg_color: str, preprocess_crop: bool, use_silhouette: bool,
                   backend: Optional[str] = None) -> int:
    """Collages of the first n_preview successful samples under reports_dir/sample_images (reference
    :827-866): views 01, 05 and 09 (else the first three PNG files) on the left, then the preprocessed first of
    them and its silhouette (the preprocessed image again without --use_silhouette).  The first views of all
    previews go through the device in one batch per view size.  Prints the silhouette's white-pixel count for
    the first five (the reference also prints the count of the Otsu mask before the component filter, an
    intermediate the device does not write out).  Returns the number of collages written."""
    import torch
    from . import engine

    picks = []
    for r in [r for r in results if r.get("success")][:max(0, n_preview)]:
        d = views_dir / r["id"]
        chosen = [d / n for n in ("view_01.png", "view_05.png", "view_09.png") if (d / n).exists()]
        if len(chosen) < 3:
            chosen = sorted(d.glob("*.png"))[:3]
        if chosen:
            picks.append((r["id"], d, chosen))
    groups: Dict[tuple, List[int]] = {}
    first: List[Optional[np.ndarray]] = []
    for k, (sid, d, chosen) in enumerate(picks):
        try:
            first.append(load_image(chosen[0]))
            groups.setdefault(first[-1].shape, []).append(k)
        except Exception as e:
            first.append(None)
            _log_error(reports_dir, d, "preview_failed", e)
    done: Dict[int, tuple] = {}
    fill = 255 if bg_color == "white" else 0
    for shape, ks in groups.items():
        try:
            dev = torch.from_numpy(np.stack([first[k] for k in ks])).to(_device())
            sil = engine.silhouette_batch(dev, resolution, crop_margin, preprocess_crop, info=True,
                                          backend=backend or SILHOUETTE_BACKEND)
            pre = engine.gray_fit_batch(dev, sil["info"], resolution, fill).cpu().numpy()
            masks = sil["mask"].cpu().numpy()
            for j, k in enumerate(ks):
                done[k] = (pre[j], masks[j])
        except Exception as e:
            for k in ks:
                _log_error(reports_dir, picks[k][1], "preview_failed", e)
    written = 0
    for k, (sid, d, chosen) in enumerate(picks):
        if k not in done:
            continue
        pre, mask = done[k]
        try:
            save_preview(sid, chosen, pre, mask if use_silhouette else pre,
                         reports_dir / "sample_images" / f"{sid}_hu_preview.png", bg_color, resolution)
            written += 1
            if written <= 5:
                print(f"[mask-clean] {sid}: white-pixels after={int(np.count_nonzero(mask))}")
        except Exception as e:
            _log_error(reports_dir, d, "preview_failed", e)
    return written


def _log_error(reports_dir: Path, where, kind: str, msg) -> None:
    try:
        ensure_dir(reports_dir)
        with open(reports_dir / "read_errors.log", "a", encoding="utf-8") as fh:
            fh.write(f"{where}\t{kind}\t{msg}\n")
    except Exception:
        pass


def png_on_device() -> bool:
    """PNG views are inflated on the device (FX_PNG_DEVICE=0 in the environment keeps the decode on the host's
    threads, as for every other file type)."""
    if os.environ.get("FX_PNG_DEVICE", "1") == "0":
        return False
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def decode_samples(tasks: Iterable[Tuple[str, str, Dict[str, Any]]], workers: Optional[int] = None) -> Dict[str, Any]:
    """Host half of process_samples: lists every task's views and decodes
    them.  PNG files of the batch's common size go through the native batch
    path: their zlib streams are staged in one page-locked block for the
    device to inflate (png_on_device), or they are decoded by native threads
    straight into such a block; every other file (another size or format,
    anything the native path does not take) is opened with PIL on a thread
    pool.  Views that fail to load are logged to
    <reports_dir>/read_errors.log and skipped; a task whose directory cannot be
    listed is marked failed."""
    from . import ingest

    tasks = list(tasks)
    workers = ingest.default_workers() if workers is None else int(workers)
    results: List[Dict[str, Any]] = []
    jobs: List[Tuple[int, Path]] = []
    for k, (sample_id, dir_path_str, config) in enumerate(tasks):
        res = {"success": False, "id": sample_id, "vec": None, "note": ""}
        results.append(res)
        dir_path = Path(dir_path_str)
        try:
            view_files = list_view_files(dir_path)
            if not view_files:
                raise FileNotFoundError(f"No view images in {dir_path}")
            res["_n_files"] = len(view_files)
            jobs.extend((k, v) for v in view_files)
        except Exception as e:
            _log_error(Path(config.get("reports_dir", "reports")), dir_path, type(e).__name__, e)
            res["note"] = f"failed: {e}"

    def key_of(k: int, shape) -> tuple:
        config = tasks[k][2]
        return (tuple(shape), int(config["resolution"]), float(config["crop_margin"]),
                bool(config["preprocess_crop"]), bool(config["use_silhouette"]), str(config["bg_color"]),
                str(config.get("silhouette_backend") or SILHOUETTE_BACKEND))

    # native batch decode of the PNG files that share the first PNG's size
    png = [j for j, (_, v) in enumerate(jobs) if v.suffix.lower() == ".png"]
    size = None
    for j in png[:8]:
        size = ingest.image_size(jobs[j][1])
        if size:
            break
    block = owner = staged = None
    native_ok = np.zeros(len(jobs), dtype=bool)
    row_of = {}
    if size and png and png_on_device():
        # the files' zlib streams go to the device as they are; inflate and row filters run there (finish_samples)
        staged = ingest.stage_png_batch([str(jobs[j][1]) for j in png], size[0], size[1], workers)
        r = 0
        for k, j in enumerate(png):
            if staged["status"][k] == 0:
                native_ok[j] = True
                row_of[j] = r
                r += 1
    elif size and png:
        block, owner = ingest.host_buffer((len(png), size[0], size[1]), np.uint8)
        status, _found = ingest.decode_png_batch([str(jobs[j][1]) for j in png], size[0], size[1], block, workers)
        for r, j in enumerate(png):
            if status[r] == 0:
                native_ok[j] = True
                row_of[j] = r
    rest = [j for j in range(len(jobs)) if not native_ok[j]]
    views: List[np.ndarray] = []
    owners: List[int] = []
    keys: List[tuple] = []
    seq: List[int] = []          # position in the task-by-task, view-by-view order
    for j, (ok, val) in zip(rest, ingest.map_ordered(lambda j: load_image(jobs[j][1]), rest, workers)):
        k, vpath = jobs[j]
        if not ok:
            _log_error(Path(tasks[k][2].get("reports_dir", "reports")), vpath, type(val).__name__, val)
            continue
        views.append(val)
        owners.append(k)
        keys.append(key_of(k, val.shape))
        seq.append(j)
    in_block = [j for j in range(len(jobs)) if native_ok[j]]
    block_owners = [jobs[j][0] for j in in_block]
    return {"tasks": tasks, "results": results, "views": views, "owners": owners, "keys": keys, "seq": seq,
            "paths": [str(v) for _, v in jobs],
            "workers": workers, "block": block, "block_owner_tensor": owner, "staged": staged, "block_size": size,
            "block_rows": [row_of[j] for j in in_block], "block_owners": block_owners,
            "block_keys": [key_of(k, size) for k in block_owners], "block_seq": in_block}


def finish_samples(decoded: Dict[str, Any]) -> List[Dict[str, Any]]:
    """Device half of process_samples: per distinct (view size, settings) one
    silhouette call and one moment launch, then one group mean/std launch over
    all samples (each sample's rows in file-name order, whatever the sizes of
    its views); writes <id>_hu.npy per sample."""
    import torch
    from . import engine, ingest

    tasks, results = decoded["tasks"], decoded["results"]
    views, owners, keys = decoded["views"], decoded["owners"], decoded["keys"]
    b_rows, b_owners, b_keys = decoded["block_rows"], decoded["block_owners"], decoded["block_keys"]
    all_seq = np.asarray(list(decoded["block_seq"]) + list(decoded["seq"]), dtype=np.int64)
    all_own = np.asarray(list(b_owners) + list(owners), dtype=np.int64)
    stats_by_task: Dict[int, np.ndarray] = {}
    if len(all_seq):
        try:
            dev = _device()
            order = np.argsort(all_seq, kind="stable")       # task by task, each sample's views in file-name order
            rank = np.empty(len(order), dtype=np.int64)
            rank[order] = np.arange(len(order))
            rows_all = torch.zeros((len(order), 7), dtype=torch.float32, device=dev)
            usable = np.ones(len(order), dtype=bool)     # per row of rows_all: its view went through
            d_block = None
            nb = len(b_rows)
            if decoded.get("staged") is not None and nb:
                # inflate + row filters of the staged PNG streams; a stream the device declines goes to PIL, and a
                # view PIL cannot read either is logged and left out, as any view that fails to load
                H0, W0 = decoded["block_size"]
                got = engine.png_inflate_batch(decoded["staged"], H0, W0, dev)
                d_block = got["gray"]
                declined = np.nonzero(got["status"].cpu().numpy())[0]
                row_to_pos = {r: i for i, r in enumerate(b_rows)}
                for r in declined.tolist():
                    i = row_to_pos[r]
                    j = decoded["block_seq"][i]
                    try:
                        img = load_image(Path(decoded["paths"][j]))
                        if img.shape != (H0, W0):
                            raise ValueError(f"view of {img.shape}, expected {(H0, W0)}")
                        d_block[r] = torch.from_numpy(np.ascontiguousarray(img)).to(dev)
                    except Exception as e:
                        usable[rank[i]] = False
                        _log_error(Path(tasks[b_owners[i]][2].get("reports_dir", "reports")), decoded["paths"][j],
                                   type(e).__name__, e)
            # one pass per distinct (view size, settings): normally a single one
            for key in sorted(set(keys) | set(b_keys)):
                idx = [i for i, kk in enumerate(keys) if kk == key]
                bidx = [i for i, kk in enumerate(b_keys) if kk == key]
                _, resolution, crop_margin, preprocess_crop, use_silhouette, bg_color, backend = key
                where = rank[np.asarray(bidx + [nb + i for i in idx], dtype=np.int64)]
                try:
                    parts = []
                    if bidx:
                        if d_block is None:
                            d_block = decoded["block_owner_tensor"].to(dev, non_blocking=True)
                        rows = [b_rows[i] for i in bidx]
                        parts.append(d_block if rows == list(range(d_block.shape[0]))
                                     else d_block[torch.as_tensor(rows, dtype=torch.int64, device=dev)])
                    if idx:
                        parts.append(ingest.stack_to_device([views[i] for i in idx], dev, decoded["workers"]))
                    raw = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
                    if use_silhouette:
                        # moments summed while a silhouette is on chip; the silhouettes are not written
                        got = engine.silhouette_hu_batch(raw, resolution, crop_margin, preprocess_crop,
                                                         backend=backend)["hulog7"]
                    else:
                        # without a crop the gray view does not depend on the silhouette (reference :372-374)
                        sil = engine.silhouette_batch(raw, resolution, crop_margin, preprocess_crop, info=True,
                                                      backend=backend) if preprocess_crop else None
                        target = engine.gray_fit_batch(raw, sil["info"] if preprocess_crop else None, resolution,
                                                       255 if bg_color == "white" else 0)
                        got = engine.hu_batch(target)["hulog7"]
                    rows_all[torch.from_numpy(where).to(dev)] = got
                except Exception as e:
                    # the reference logs a view that fails and goes on with the sample's other views (:629-636)
                    usable[where] = False
                    for j in [decoded["block_seq"][i] for i in bidx] + [decoded["seq"][i] for i in idx]:
                        k = all_own[np.nonzero(all_seq == j)[0][0]]
                        _log_error(Path(tasks[k][2].get("reports_dir", "reports")), decoded["paths"][j],
                                   type(e).__name__, e)
            kept = np.nonzero(usable)[0]
            if len(kept) < len(order):
                rows_all = rows_all[torch.from_numpy(kept).to(dev)].contiguous()
            uniq, counts = np.unique(all_own[order][kept], return_counts=True)   # owners ascend along the order
            off = np.zeros(len(counts) + 1, dtype=np.int64)
            np.cumsum(counts, out=off[1:])
            stats = engine.hu_groupstat(rows_all, torch.from_numpy(off).to(dev)).cpu().numpy()
            for g, k in enumerate(uniq.tolist()):
                stats_by_task[k] = stats[g]
        except Exception as e:
            for k in set(all_own.tolist()):
                if not results[k]["note"].startswith("failed"):
                    results[k]["note"] = f"failed: {e}"

    # descriptor files of the whole batch through the native writer (one call; np.save for what it declines)
    pending: List[tuple] = []   # (task, path, vector, files found)
    made: set = set()
    for k, (sample_id, dir_path_str, config) in enumerate(tasks):
        res = results[k]
        n_files = res.pop("_n_files", 0)
        if res["note"].startswith("failed"):
            continue
        try:
            if k not in stats_by_task:
                raise RuntimeError("No usable views for Hu computation")
            out_dir = Path(config["out_dir"])
            if str(out_dir) not in made:
                ensure_dir(out_dir)
                made.add(str(out_dir))
            pending.append((k, str(out_dir / f"{sample_id}_hu.npy"), stats_by_task[k].astype(np.float32), n_files))
        except Exception as e:
            _log_error(Path(config.get("reports_dir", "reports")), dir_path_str, type(e).__name__, e)
            res["note"] = f"failed: {e}"
    status = np.zeros(0, dtype=np.int32)
    if pending:
        try:
            status = ingest.write_rows_batch([p[1] for p in pending], np.stack([p[2] for p in pending]), decoded["workers"])
        except Exception:
            status = np.ones(len(pending), dtype=np.int32)
    for j, (k, path, vec, n_files) in enumerate(pending):
        sample_id, dir_path_str, config = tasks[k]
        res = results[k]
        try:
            if status[j] != 0:
                np.save(path, vec, allow_pickle=False)
            note = "partial_views" if n_files < int(config["expected_views"]) else "ok"
            res.update({"success": True, "vec": vec.tolist(), "note": note})
        except Exception as e:
            _log_error(Path(config.get("reports_dir", "reports")), dir_path_str, type(e).__name__, e)
            res["note"] = f"failed: {e}"
    return results


def process_samples(tasks: Iterable[Tuple[str, str, Dict[str, Any]]], workers: Optional[int] = None) -> List[Dict[str, Any]]:
    """Batched process_sample_worker: decodes every view of every sample on a
    host thread pool, then per distinct (view size, settings) one silhouette
    call, one moment launch and one group mean/std launch on the device;
    writes <id>_hu.npy per sample and returns one {"success", "id", "vec",
    "note"} dict per task in task order (reference :602-667).  Views that fail
    to load are logged to <reports_dir>/read_errors.log and skipped; a sample
    without usable views fails with note "failed: ..."."""
    return finish_samples(decode_samples(tasks, workers))


def process_sample_worker(args_tuple: Tuple) -> Dict[str, Any]:
    """(sample_id, dir, config) -> {"success", "id", "vec", "note"} (reference :602-667)."""
    return process_samples([args_tuple])[0]


def results_to_records(results: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
    """CSV rows; failed samples get NaN features and keep their note (reference :763-785)."""
    records = []
    for r in results:
        rec: Dict[str, Any] = {"id": r.get("id")}
        if r.get("success"):
            vec = np.asarray(r.get("vec"), dtype=np.float32)
            vals = [float(v) for v in vec.tolist()]
        else:
            vals = [float("nan")] * 14
        for i in range(7):
            rec[f"hu{i + 1}_mean"] = vals[i]
        for i in range(7):
            rec[f"hu{i + 1}_std"] = vals[7 + i]
        rec["note"] = r.get("note", "")
        records.append(rec)
    return records


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Extract Hu moments per view and aggregate per sample (MI355X)")
    p.add_argument("--views_dir", type=str, default="../../data/processed/views/")
    p.add_argument("--out_dir", type=str, default="../../data/processed/2d_hu")
    p.add_argument("--resolution", type=int, default=224)
    p.add_argument("--use_silhouette", action="store_true")
    p.add_argument("--preprocess_crop", action="store_true")
    p.add_argument("--crop_margin", type=float, default=0.05)
    p.add_argument("--bg_color", type=str, default="white", choices=["white", "black"])
    p.add_argument("--n_preview", type=int, default=0, help="collages of the first N samples under reports_dir/sample_images")
    p.add_argument("--workers", type=int, default=None,
                   help="decode threads on the host (default: the cores, at most 16)")
    p.add_argument("--fast", action="store_true")
    p.add_argument("--max_samples", type=int, default=None)
    p.add_argument("--expected_views", type=int, default=12)
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--reports_dir", type=str, default="../../reports")
    p.add_argument("--batch_samples", type=int, default=512, help="samples per device batch")
    p.add_argument("--silhouette_backend", type=str, default=None, choices=["opencv", "pil"],
                   help="branch of the reference's preprocessing to reproduce (default: opencv, its production path)")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--verbose", action="store_true")
    return p.parse_args(argv)


def main(argv=None) -> int:
    """CLI with the reference's flags and outputs (:703-884):
    hu_features_table.csv, <id>_hu.npy, hu_feature_stats.csv, the run config."""
    import pandas as pd

    args = parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO,
                        format="%(asctime)s [%(levelname)s] %(message)s")
    views_dir, out_dir, reports_dir = Path(args.views_dir), Path(args.out_dir), Path(args.reports_dir)
    ensure_dir(reports_dir)
    ensure_dir(out_dir)
    ids = list_sample_ids(views_dir)
    if not ids:
        logger.warning(f"No sample directories found under {views_dir}")
        return 0
    limit = 20 if args.fast else args.max_samples
    if limit is not None:
        ids = ids[:limit]
    config = {"out_dir": str(out_dir), "resolution": int(args.resolution), "use_silhouette": bool(args.use_silhouette),
              "preprocess_crop": bool(args.preprocess_crop), "crop_margin": float(args.crop_margin),
              "bg_color": args.bg_color, "expected_views": int(args.expected_views),
              "overwrite": bool(args.overwrite), "reports_dir": str(reports_dir),
              "silhouette_backend": args.silhouette_backend or SILHOUETTE_BACKEND}
    tasks = [(sid, str(d), config) for sid, d in ids]
    results: List[Dict[str, Any]] = []
    step = max(1, int(args.batch_samples))
    from . import ingest
    workers = int(args.workers) if args.workers else None
    # the next batch is decoded while the current one is on the device
    for decoded in ingest.prefetch(lambda chunk: decode_samples(chunk, workers),
                                   (tasks[i:i + step] for i in range(0, len(tasks), step))):
        results.extend(finish_samples(decoded))
    out_csv = out_dir / "hu_features_table.csv"
    append_to_csv(results_to_records(results), out_csv, overwrite=args.overwrite)
    try:
        num = pd.read_csv(out_csv).select_dtypes(include=[np.number])
        num.agg(["mean", "std"]).transpose().reset_index().rename(columns={"index": "feature"}).to_csv(
            reports_dir / "hu_feature_stats.csv", index=False)
    except Exception as e:
        logger.debug(f"Failed to write hu_feature_stats.csv: {e}")
    cfg = dict(config, timestamp=datetime.datetime.now().isoformat(sep=" ", timespec="seconds"),
               views_dir=str(views_dir), workers=args.workers, seed=int(args.seed))
    try:
        import yaml
        with open(out_dir / "extract_2d_hu_config.yaml", "w", encoding="utf-8") as fh:
            yaml.safe_dump(cfg, fh)
    except Exception:
        logger.debug("Failed to write config YAML")
    if args.n_preview > 0:
        write_previews(results, views_dir, reports_dir, int(args.n_preview), int(args.resolution), float(args.crop_margin),
                       args.bg_color, bool(args.preprocess_crop), bool(args.use_silhouette), config["silhouette_backend"])
    mode = "fast" if args.fast else "full"
    print(f"Hu extraction finished. Mode: {mode}. Samples processed: {len(results)}. Descriptors saved to {out_dir}.")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
