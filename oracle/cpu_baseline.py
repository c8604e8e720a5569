"""Worker functions of bench.py's cpu_baseline leg (TEST/BENCH INFRASTRUCTURE:
the CPU oracle timed on the host cores; never part of the product path).

Each worker processes its slice of the sample one unit at a time, the way the
reference's pool workers process one file at a time
(extract_2d_hu.py:751-752, extract_3d_pca_features.py:349-350).
"""
import time

import numpy as np

from oracle import cpu_ref


def warm(job):
    """Imports, builds/loads the C oracle and maps the sample; untimed."""
    cpu_ref.load_c_oracle()
    np.load(job[1], mmap_mode="r")
    return True


def run(job):
    kind, path, lo, hi = job
    data = np.load(path, mmap_mode="r")[lo:hi]
    data = np.ascontiguousarray(data)
    c = cpu_ref.load_c_oracle()
    t0 = time.perf_counter()
    if kind == "pca":
        for pts in data:
            cpu_ref.compute_pca_features(pts.astype(np.float64), 8, False)
    elif kind == "hu":
        for img in data:
            c.hu_batch(img[None], thresh=127, on_value=255)
    elif kind == "views+hu":
        from importlib import import_module
        stamp = np.array(import_module("image-processing_amd.marker_stamps").STAMPS[2.0], dtype=np.uint8)  # data table only
        for pts in data:
            p64 = pts.astype(np.float64)
            mins, maxs, cen, _ = cpu_ref.bbox_and_center(p64)
            rows = []
            for pose in cpu_ref.compute_camera_poses(12, mins, maxs, cen):
                view = cpu_ref.render_view(p64, pose, 224, stamp)
                _, mask = cpu_ref.preprocess_image_cv(view, 224, 0.05, "white", True)
                rows.append(c.hu_batch(mask[None])["hulog7"][0])
            rows = np.stack(rows).astype(np.float32)
            np.concatenate([rows.mean(axis=0), rows.std(axis=0)])
    elif kind == "tail":
        for img in data:
            c.hu_batch(img[None], thresh=127, on_value=255)
    elif kind == "sil+hu":
        for img in data:
            mask, _ = cpu_ref.silhouette_cv(img, 224, 0.05, True)
            c.hu_batch(mask[None])
    else:
        for img in data:
            c.hu_batch(img[None], thresh=127, on_value=255)
            pts = c.contour_points(img, 127)
            if pts.shape[0] >= 3:
                cpu_ref.compute_pca_features(pts, 8, False)
    return hi - lo, time.perf_counter() - t0
