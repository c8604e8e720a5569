"""Multi-view renderer of sampled point clouds on the device, with the
function names, flags and outputs of the reference's script
(src/preprocessing/render_views.py) for its renderer without Open3D /
pyrender: render_fallback_matplotlib, i.e. a matplotlib scatter plot drawn by
the Agg backend.  The device kernel (fx_render_views_u8) reproduces that
drawing pixel for pixel: projection, axis limits and equal-aspect box, marker
snapping, marker coverage stamp, Agg's integer blend in point order.

    python -m image-processing_amd.render_views --sampled_dir ... --out_dir ... --views 12 --resolution 224

writes <out_dir>/<id>/view_XX.png, manifest_views.csv and render_config.yaml
like the reference; render_views_batch keeps the views in HBM for the Hu path.
"""
from __future__ import annotations

import argparse
import datetime
import logging
import math
import sys
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import engine
from .marker_stamps import STAMPS

logger = logging.getLogger("render_views")


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("image-processing_amd needs a ROCm device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def ensure_dir(p: Path) -> None:
    p.mkdir(parents=True, exist_ok=True)


def load_sample(path: Path) -> np.ndarray:
    """(N, 3) float64 from a sampled .npy file (reference :66-74)."""
    arr = np.load(path, allow_pickle=False)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError(f"invalid shape {arr.shape} (expected Nx3) for {path}")
    return arr[:, :3].astype(np.float64)


def bbox_and_center(points: np.ndarray):
    """min, max, centroid, diagonal (reference :77-87)."""
    mins = points.min(axis=0)
    maxs = points.max(axis=0)
    return mins, maxs, points.mean(axis=0), float(np.linalg.norm(maxs - mins))


def view_angles(n_views: int, seed: int = 42, elevation_levels: Optional[Sequence[float]] = None) -> List[Tuple[float, float]]:
    """(elevation, azimuth) in degrees of the n_views poses (reference :110-141):
    the same for every cloud, the azimuth jitter comes from RandomState(seed)."""
    levels = [-30.0, 0.0, 30.0] if elevation_levels is None else list(elevation_levels)
    per_level = max(1, n_views // len(levels))
    counts = [per_level] * len(levels)
    for i in range(n_views - per_level * len(levels)):
        counts[i] += 1
    rng = np.random.RandomState(seed)
    angles = []
    for elev, cnt in zip(levels, counts):
        for k in range(cnt):
            angles.append((elev, float((360.0 * k) / max(1, cnt)) + float(rng.uniform(-1e-6, 1e-6))))
    if len(angles) > n_views:
        angles = angles[:n_views]
    while len(angles) < n_views:
        angles.append(angles[len(angles) % len(angles)])
    return angles


def compute_camera_poses(n_views: int, bbox_min: np.ndarray, bbox_max: np.ndarray, centroid: np.ndarray,
                         radius_factor: float = 2.0, seed: int = 42,
                         elevation_levels: Optional[List[float]] = None) -> List[Dict[str, np.ndarray]]:
    """Camera poses of one cloud (reference :93-164): position, lookat, up."""
    diag = float(np.linalg.norm(bbox_max - bbox_min))
    radius = diag * radius_factor if diag > 1e-9 else radius_factor
    poses = []
    for elev, az in view_angles(n_views, seed, elevation_levels):
        el_rad, az_rad = math.radians(elev), math.radians(az)
        position = np.array([radius * math.cos(el_rad) * math.cos(az_rad),
                             radius * math.cos(el_rad) * math.sin(az_rad),
                             radius * math.sin(el_rad)], dtype=np.float64) + centroid
        up = np.array([0.0, 0.0, 1.0], dtype=np.float64)
        fwd = centroid - position
        if np.linalg.norm(fwd) < 1e-9:
            fwd = np.array([0.0, 0.0, 1.0], dtype=np.float64)
        fwd = fwd / np.linalg.norm(fwd)
        if abs(np.dot(fwd, up)) > 0.999:
            up = np.array([0.0, 1.0, 0.0], dtype=np.float64)
        poses.append({"position": position, "lookat": centroid.copy(), "up": up, "elevation": elev, "azimuth": az})
    return poses


def _norm3(v: np.ndarray) -> np.ndarray:
    return np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2])


def _cross(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def plane_basis(position: np.ndarray, lookat: np.ndarray, up: np.ndarray) -> np.ndarray:
    """[..., 9] = right, true_up, look-at of cameras given as arrays [..., 3]
    (reference :306-320), for any number of cameras at once."""
    forward = lookat - position
    n = _norm3(forward)
    small = n < 1e-12
    forward = np.where(small[..., None], np.array([0.0, 0.0, 1.0]), forward)
    forward = forward / _norm3(forward)[..., None]
    right = _cross(forward, up)
    flat = _norm3(right) < 1e-9
    if np.any(flat):
        right = np.where(flat[..., None], _cross(forward, np.broadcast_to(np.array([0.0, 1.0, 0.0]), forward.shape)), right)
    right = right / (_norm3(right) + 1e-12)[..., None]
    true_up = _cross(right, forward)
    true_up = true_up / (_norm3(true_up) + 1e-12)[..., None]
    return np.concatenate([right, true_up, np.broadcast_to(lookat, right.shape)], axis=-1)


def view_trig(n_views: int, seed: int = 42, elevation_levels: Optional[Sequence[float]] = None) -> np.ndarray:
    """[V, 4] = cos(elevation), sin(elevation), cos(azimuth), sin(azimuth) of the
    poses, computed with math.* as the reference computes them (:134-139)."""
    out = np.empty((n_views, 4), dtype=np.float64)
    for v, (elev, az) in enumerate(view_angles(n_views, seed, elevation_levels)):
        el_rad, az_rad = math.radians(elev), math.radians(az)
        out[v] = (math.cos(el_rad), math.sin(el_rad), math.cos(az_rad), math.sin(az_rad))
    return out


def camera_table(bounds: np.ndarray, n_views: int, resolution: int, radius_factor: float = 2.0, seed: int = 42,
                 elevation_levels: Optional[Sequence[float]] = None):
    """Camera table of B clouds from their bounds [B, 9] (min, max, mean):
    cam float64 [B, V, 9] for fx_render_views_u8 and scale float64 [B]."""
    bounds = np.asarray(bounds, dtype=np.float64)
    mins, maxs, cen = bounds[:, 0:3], bounds[:, 3:6], bounds[:, 6:9]
    diag = _norm3(maxs - mins)
    radius = np.where(diag > 1e-9, diag * radius_factor, radius_factor)
    angles = view_angles(n_views, seed, elevation_levels)
    el = [math.radians(e) for e, _ in angles]
    az = [math.radians(a) for _, a in angles]
    cel = np.array([math.cos(v) for v in el]); sel = np.array([math.sin(v) for v in el])
    caz = np.array([math.cos(v) for v in az]); saz = np.array([math.sin(v) for v in az])
    r = radius[:, None]
    offset = np.stack([(r * cel) * caz, (r * cel) * saz, r * sel], axis=-1)          # [B, V, 3]
    lookat = np.broadcast_to(cen[:, None, :], offset.shape)
    position = offset + lookat
    fwd = lookat - position
    n = _norm3(fwd)
    fwd = np.where((n < 1e-9)[..., None], np.array([0.0, 0.0, 1.0]), fwd)
    fwd = fwd / _norm3(fwd)[..., None]
    up = np.where((np.abs(fwd[..., 2]) > 0.999)[..., None], np.array([0.0, 1.0, 0.0]), np.array([0.0, 0.0, 1.0]))
    cam = np.ascontiguousarray(plane_basis(position, lookat, up))
    scale = (resolution * 0.35) / np.where(diag < 1e-9, 1.0, diag)
    return cam, np.ascontiguousarray(scale)


_MEASURED_STAMPS: Dict[float, np.ndarray] = {}
MAX_STAMP_SIDE = 7   # the kernel keeps a stamp in the lanes of a wave: at most 7 x 7 cells


def measure_marker_stamp(s: float) -> np.ndarray:
    """Coverage stamp of one round marker of area `s` points^2 at dpi 100, taken from matplotlib's Agg backend
    itself (the library the reference's renderer draws with): the marker is drawn on figures of four gray
    levels, and the alpha of every cell is the only value consistent with Agg's blend on all four.  Agg
    rasterises a scatter marker once and stamps it at whole-pixel positions, so one measurement serves every
    point.  Used for marker sizes that are not tabulated in marker_stamps.py."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception as e:  # pragma: no cover - depends on the installation
        raise ValueError(f"point_size {s}: not tabulated, and matplotlib is not importable to measure it ({e})")
    from io import BytesIO
    from PIL import Image

    def blend(p: int, a: int) -> int:   # Agg's plain RGBA blend of a black marker onto an opaque gray p
        if a == 0:
            return p
        r = p * 255
        return ((((0 << 8) - r) * a + (r << 8)) & 0xFFFFFFFF) // (((a + 255) << 8) - a * 255)

    res, R = 64, 15
    cands = None
    for gray in (255, 200, 128, 64):
        fig = plt.figure(figsize=(res / 100.0, res / 100.0), dpi=100, facecolor=(gray / 255.0,) * 3)
        ax = fig.add_subplot(111)
        ax.scatter([32.2], [31.7], s=s, color=(0, 0, 0), marker="o", linewidths=0, alpha=1.0)
        ax.axis("off")
        ax.set_frame_on(False)
        ax.set_xlim(0, res)
        ax.set_ylim(0, res)
        ax.set_aspect("equal", adjustable="box")
        fig.subplots_adjust(0, 0, 1, 1)
        buf = BytesIO()
        fig.canvas.print_png(buf)
        buf.seek(0)
        img = np.asarray(Image.open(buf).convert("RGB"))[:, :, 0].astype(int)
        plt.close(fig)
        win = img[32 - R:32 + R + 1, 32 - R:32 + R + 1]
        table = {}
        for a in range(256):
            table.setdefault(blend(gray, a), set()).add(a)
        sets = [[table.get(int(win[j, i]), set()) for i in range(2 * R + 1)] for j in range(2 * R + 1)]
        cands = sets if cands is None else [[cands[j][i] & sets[j][i] for i in range(2 * R + 1)] for j in range(2 * R + 1)]
    if not all(len(c) == 1 for row in cands for c in row):
        raise ValueError(f"point_size {s}: the marker's coverage could not be measured")
    al = np.array([[min(c) for c in row] for row in cands])
    ys, xs = np.nonzero(al)
    if ys.size == 0:
        return np.zeros((1, 1), dtype=np.uint8)
    k = int(max(np.abs(ys - R).max(), np.abs(xs - R).max()))
    return al[R - k:R + k + 1, R - k:R + k + 1].astype(np.uint8)


def marker_stamp(point_size: float) -> np.ndarray:
    """Coverage stamp Agg draws for scatter(s=max(0.1, point_size)) at dpi 100: from the table of
    marker_stamps.py, or measured from matplotlib once per size (the reference accepts any size)."""
    s = max(0.1, float(point_size))
    if s in STAMPS:
        return np.array(STAMPS[s], dtype=np.uint8)
    if s not in _MEASURED_STAMPS:
        _MEASURED_STAMPS[s] = measure_marker_stamp(s)
    stamp = _MEASURED_STAMPS[s]
    if stamp.shape[0] > MAX_STAMP_SIDE:
        raise ValueError(f"point_size {point_size}: a marker of {stamp.shape[0]} x {stamp.shape[0]} pixels; the device "
                         f"renderer takes markers up to {MAX_STAMP_SIDE} x {MAX_STAMP_SIDE}")
    return stamp


def pack_clouds(clouds: Sequence[np.ndarray], device: torch.device):
    """float32 [total, 3] and int64 offsets [B + 1] on the device."""
    sizes = [int(c.shape[0]) for c in clouds]
    offsets = np.zeros(len(clouds) + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    flat = np.concatenate([np.asarray(c, dtype=np.float32)[:, :3] for c in clouds]) if clouds else np.zeros((0, 3), np.float32)
    return torch.from_numpy(np.ascontiguousarray(flat)).to(device), torch.from_numpy(offsets).to(device)


def render_views_device(points: torch.Tensor, offsets: torch.Tensor, views: int = 12, resolution: int = 224,
                        point_size: float = 2.0, bg_color: str = "white", camera_radius_factor: float = 2.0,
                        seed: int = 42) -> torch.Tensor:
    """uint8 [B, views, res, res] (device) for clouds already in HBM; every cloud
    must hold at least one point."""
    bounds = engine.cloud_bounds(points, offsets)
    if views <= 64:
        # poses and camera bases on the device: no host round trip inside the chain
        cam, scale = engine.camera_table(bounds, view_trig(views, seed), camera_radius_factor, resolution)
    else:
        cam_h, scale_h = camera_table(bounds.cpu().numpy(), views, resolution, camera_radius_factor, seed)
        cam, scale = torch.from_numpy(cam_h).to(points.device), torch.from_numpy(scale_h).to(points.device)
    return engine.render_views(points, offsets, cam, scale, resolution, marker_stamp(point_size),
                               0 if bg_color == "white" else 255)


def render_views_batch(clouds: Sequence[np.ndarray], views: int = 12, resolution: int = 224, point_size: float = 2.0,
                       bg_color: str = "white", camera_radius_factor: float = 2.0, seed: int = 42) -> torch.Tensor:
    """Views of a list of (N_i, 3) clouds: uint8 [B, views, res, res] on the device."""
    for c in clouds:
        if c.shape[0] < 1:
            raise ValueError("empty point cloud")
    dev = _device()
    pts, off = pack_clouds(clouds, dev)
    return render_views_device(pts, off, views, resolution, point_size, bg_color, camera_radius_factor, seed)


def render_fallback_matplotlib(points: np.ndarray, camera_pose: Dict[str, np.ndarray], resolution: int, point_size: float,
                               bg_color: str, ortho: bool):
    """One view as a PIL image, RGB for a white background and RGBA otherwise
    (reference :294-366; the projection is orthographic whatever `ortho` says)."""
    from PIL import Image
    if points.shape[0] < 1:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    dev = _device()
    pts, off = pack_clouds([points], dev)
    cam = plane_basis(camera_pose["position"][None, :], camera_pose["lookat"][None, :],
                      camera_pose["up"].astype(np.float64)[None, :])[None]
    diag = float(np.linalg.norm(points.max(axis=0) - points.min(axis=0)))
    scale = np.array([(resolution * 0.35) / (1.0 if diag < 1e-9 else diag)], dtype=np.float64)
    img = engine.render_views(pts, off, torch.from_numpy(np.ascontiguousarray(cam)).to(dev), torch.from_numpy(scale).to(dev),
                              resolution, marker_stamp(point_size), 0 if bg_color == "white" else 255)[0, 0].cpu().numpy()
    rgb = Image.fromarray(np.repeat(img[:, :, None], 3, axis=2), mode="RGB")
    return rgb if bg_color == "white" else rgb.convert("RGBA")


def save_image(img, path: Path, transparent: bool) -> None:
    """PNG on disk; with `transparent`, near-white pixels get alpha 0 (reference :369-386)."""
    ensure_dir(path.parent)
    if transparent and img.mode != "RGBA":
        a = np.asarray(img.convert("RGBA")).copy()
        white = (a[:, :, 0] > 250) & (a[:, :, 1] > 250) & (a[:, :, 2] > 250)
        a[white] = (255, 255, 255, 0)
        from PIL import Image
        img = Image.fromarray(a, mode="RGBA")
    img.save(str(path), format="PNG")


def save_collage(img_paths: List[Path], out_preview_path: Path, thumbnail_w: int = 224, bg_color: str = "white") -> None:
    """Up to three views side by side (reference :389-416)."""
    from PIL import Image, ImageOps
    if not img_paths:
        return
    fill = (255, 255, 255) if bg_color == "white" else (0, 0, 0)
    thumbs = []
    for p in img_paths[:3]:
        try:
            thumbs.append(ImageOps.contain(Image.open(p), (thumbnail_w, thumbnail_w)).convert("RGB"))
        except Exception:
            thumbs.append(Image.new("RGB", (thumbnail_w, thumbnail_w), color=fill))
    collage = Image.new("RGB", (sum(im.width for im in thumbs), max(im.height for im in thumbs)), color=fill)
    x = 0
    for im in thumbs:
        collage.paste(im, (x, 0))
        x += im.width
    ensure_dir(out_preview_path.parent)
    collage.save(str(out_preview_path), format="PNG")


def _record(sample_path: Path) -> Dict[str, Any]:
    return {"id": sample_path.stem, "sample_path": str(sample_path.resolve()), "views_dir": "", "n_views": 0, "note": "",
            "rendered_timestamp": ""}


def _now() -> str:
    return datetime.datetime.now().isoformat(sep=" ", timespec="seconds")


def _log_error(reports_dir: Path, sample_path: Path, e: Exception) -> None:
    try:
        ensure_dir(reports_dir)
        with open(reports_dir / "read_errors.log", "a", encoding="utf-8") as fh:
            fh.write(f"{sample_path}\t{type(e).__name__}\t{e}\n")
    except Exception:
        pass


def render_samples(tasks: Sequence[Tuple]) -> List[Dict[str, Any]]:
    """The reference's render_sample_worker (:422-505) for a batch of task
    tuples: files are read on the host, the views that have to be drawn are
    rendered in one device call per distinct rendering settings, PNGs written.
    A sample that fails at any step is logged and gets a "failed: ..." record;
    the others are unaffected."""
    from PIL import Image
    results: List[Optional[Dict[str, Any]]] = [None] * len(tasks)
    loaded = []
    for k, task in enumerate(tasks):
        (sample_path_str, out_dir_str, resolution, views, point_size, bg_color, transparent, ortho, radius_factor, seed,
         overwrite, reports_dir_str) = task
        sample_path, rec = Path(sample_path_str), _record(Path(sample_path_str))
        try:
            pts = load_sample(sample_path)
            if pts.shape[0] < 1:
                raise ValueError("empty point cloud")
            views_dir = Path(out_dir_str) / sample_path.stem
            ensure_dir(views_dir)
            names = [views_dir / f"view_{v + 1:02d}.png" for v in range(views)]
            need = [v for v, p in enumerate(names) if overwrite or not p.exists()]
            loaded.append((k, pts, names, need, rec))
        except Exception as e:
            _log_error(Path(reports_dir_str), sample_path, e)
            rec["note"], rec["rendered_timestamp"] = f"failed: {e}", _now()
            results[k] = {"success": False, "record": rec, "saved_paths": []}
    # the reference catches every exception per sample, logs it and returns a "failed: ..." record
    # (:442-498): one bad cloud, marker size or unwritable path must not take the batch with it
    failed: Dict[int, Exception] = {}
    todo = [item for item in loaded if item[3]]
    groups: Dict[tuple, list] = {}
    for item in todo:   # one device call per distinct rendering settings
        _, _, resolution, views, point_size, bg_color, transparent, _, radius_factor, seed, _, _ = tasks[item[0]]
        groups.setdefault((resolution, views, point_size, bg_color, transparent, radius_factor, seed), []).append(item)
    from . import ingest
    for (resolution, views, point_size, bg_color, transparent, radius_factor, seed), items in groups.items():
        def render(batch):
            return render_views_batch([it[1] for it in batch], views, resolution, point_size, bg_color, radius_factor,
                                      seed).cpu().numpy()
        stacks: Dict[int, np.ndarray] = {}
        try:
            for it, stack in zip(items, render(items)):
                stacks[it[0]] = stack
        except Exception:
            for it in items:   # find the sample(s) at fault: cloud by cloud
                try:
                    stacks[it[0]] = render([it])[0]
                except Exception as e:
                    failed[it[0]] = e

        # PNG encoding is the slow part of a run (zlib on the host): on the ingest thread pool
        def write(job):
            gray, path = job
            rgb = Image.fromarray(np.repeat(gray[:, :, None], 3, axis=2), mode="RGB")
            save_image(rgb if bg_color == "white" else rgb.convert("RGBA"), path, transparent)

        jobs, owner = [], []
        for it in items:
            if it[0] in stacks:
                for v in it[3]:
                    jobs.append((stacks[it[0]][v], it[2][v]))
                    owner.append(it[0])
        for k, (ok, err) in zip(owner, ingest.map_ordered(write, jobs, ingest.default_workers())):
            if not ok and k not in failed:
                failed[k] = err
    for k, pts, names, need, rec in loaded:
        rec["rendered_timestamp"] = _now()
        if k in failed:
            _log_error(Path(tasks[k][11]), Path(tasks[k][0]), failed[k])
            rec["note"] = f"failed: {failed[k]}"
            results[k] = {"success": False, "record": rec, "saved_paths": []}
            continue
        rec["views_dir"], rec["n_views"], rec["note"] = str(names[0].parent.resolve()), len(names), "ok"
        results[k] = {"success": True, "record": rec, "saved_paths": [str(p) for p in names]}
    return results  # type: ignore[return-value]


def render_sample_worker(args_tuple: Tuple) -> Dict[str, Any]:
    """One sample (reference :422-505)."""
    return render_samples([args_tuple])[0]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Render multi-view images for sampled point clouds.")
    p.add_argument("--sampled_dir", type=str, default="../../data/processed/sampled")
    p.add_argument("--raw_dir", type=str, default="")
    p.add_argument("--out_dir", type=str, default="../../data/processed/views/")
    p.add_argument("--views", type=int, default=12)
    p.add_argument("--resolution", type=int, default=224)
    p.add_argument("--point_size", type=float, default=2.0)
    p.add_argument("--bg_color", type=str, default="white", choices=["white", "black"])
    p.add_argument("--transparent", action="store_true")
    p.add_argument("--ortho", action="store_true")
    p.add_argument("--n_preview", type=int, default=10)
    p.add_argument("--workers", type=int, default=1, help="accepted for compatibility; the device renders the batch")
    p.add_argument("--fast", action="store_true")
    p.add_argument("--max_samples", type=int, default=None)
    p.add_argument("--camera_radius_factor", type=float, default=2.0)
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--reports_dir", type=str, default="../../reports")
    p.add_argument("--batch_files", type=int, default=256, help="clouds per device call")
    return p.parse_args(argv)


def main(argv=None) -> int:
    import pandas as pd
    import yaml
    args = parse_args(argv)
    if args.verbose:
        logger.setLevel(logging.DEBUG)
    sampled_dir, out_dir, reports_dir = Path(args.sampled_dir), Path(args.out_dir), Path(args.reports_dir)
    ensure_dir(out_dir)
    ensure_dir(reports_dir / "sample_images")
    max_samples = 20 if args.fast else args.max_samples
    n_preview = min(args.n_preview, 10) if args.fast else args.n_preview
    files = sorted([p for p in sampled_dir.rglob("*.npy") if p.is_file()], key=lambda p: str(p))
    if max_samples is not None:
        files = files[:max_samples]
    logger.info(f"Found {len(files)} sampled .npy files under {sampled_dir}.")
    if not files:
        logger.warning("No files to process; exiting.")
        return 0
    tasks = [(str(p), str(out_dir), int(args.resolution), int(args.views), float(args.point_size), args.bg_color,
              bool(args.transparent), bool(args.ortho), float(args.camera_radius_factor), int(args.seed),
              bool(args.overwrite), str(reports_dir)) for p in files]
    results: List[Dict[str, Any]] = []
    for i in range(0, len(tasks), max(1, args.batch_files)):
        results.extend(render_samples(tasks[i:i + max(1, args.batch_files)]))
    manifest_df = pd.DataFrame([r.get("record", {}) for r in results],
                               columns=["id", "sample_path", "views_dir", "n_views", "note", "rendered_timestamp"])
    manifest_path = out_dir / "manifest_views.csv"
    if manifest_path.exists() and not args.overwrite:
        try:
            combined = pd.concat([pd.read_csv(manifest_path), manifest_df], ignore_index=True)
            combined.drop_duplicates(subset=["id"], keep="last").to_csv(manifest_path, index=False)
        except Exception:
            manifest_df.to_csv(manifest_path, index=False)
    else:
        manifest_df.to_csv(manifest_path, index=False)
    config = {"timestamp": _now(), "seed": int(args.seed), "views": int(args.views), "resolution": int(args.resolution),
              "point_size": float(args.point_size), "bg_color": args.bg_color, "transparent": bool(args.transparent),
              "ortho": bool(args.ortho), "n_preview": int(n_preview), "workers": int(args.workers),
              "camera_radius_factor": float(args.camera_radius_factor), "overwrite": bool(args.overwrite)}
    try:
        with open(out_dir / "render_config.yaml", "w", encoding="utf-8") as fh:
            yaml.safe_dump(config, fh)
    except Exception:
        logger.debug("Failed to write render_config.yaml")
    for _, row in manifest_df[manifest_df["note"] == "ok"].head(n_preview).iterrows():
        views_dir = Path(row["views_dir"])
        chosen = [views_dir / v for v in ("view_01.png", "view_05.png", "view_09.png") if (views_dir / v).exists()]
        if len(chosen) < 3:
            chosen = sorted(views_dir.glob("*.png"))[:3]
        save_collage(chosen, reports_dir / "sample_images" / f"{row['id']}_preview.png", thumbnail_w=args.resolution,
                     bg_color=args.bg_color)
    mode = "fast" if args.fast else "full"
    print(f"Render finished. Mode: {mode}. Samples processed: {len(manifest_df)}. Views saved to {out_dir}.")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
