"""PCA descriptor extraction on MI355X behind the reference's function names.

Mirrors reference src/features/extract_3d_pca_features.py: the same function
names, argument meaning, return shapes, record/CSV layout and error behaviour;
the arithmetic of compute_pca_features (:59-145) runs in the HIP kernels of
libfx (csrc/fx_pca.hip).  There is no CPU fallback.

Differences from the reference that a caller can observe:
  * `compute_pca_features_batch` / `process_files` exist in addition: they put
    many sets through one kernel launch (the reference has only the per-file
    functions and a multiprocessing pool around them);
  * point sets whose values float32 holds exactly (the float32 files the
    pipeline writes) run through the float32 kernels, which compute what the
    reference computes after widening; any other input -- float64 files or
    arrays with values float32 would round -- is evaluated from its float64
    values by fx_pca_batch_f64, as load_sample / compute_pca_features do.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from pathlib import Path
from typing import Iterable, List, Optional, Sequence

import numpy as np

logger = logging.getLogger("pca_features")

BASE_COLS = ["id", "raw_path", "n_points", "lambda1", "lambda2", "lambda3",
             "r1", "r2", "r3", "ratio12", "ratio23", "trace"]


def column_names(n_bins: int) -> List[str]:
    """CSV header of descriptors_table_pca.csv (reference :383-386)."""
    return BASE_COLS + [f"proj_hist_bin_{i}" for i in range(n_bins)]


def ensure_dir(p: Path) -> None:
    p.mkdir(parents=True, exist_ok=True)


def load_sample(path: Path) -> np.ndarray:
    """Nx3 float64 array from a sampled .npy file; ValueError on a bad shape
    (reference :48-56)."""
    arr = np.load(path, allow_pickle=False)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError(f"invalid shape {arr.shape} (expected Nx3) for {path}")
    return arr[:, :3].astype(np.float64)


def _device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the PCA extractor runs on the GPU only")
    return torch.device("cuda", torch.cuda.current_device())


def compute_pca_features_batch(point_sets: Sequence[np.ndarray], n_bins: int = 8, normalize: bool = False,
                               return_eig: bool = False):
    """Descriptors of many point sets in one launch -> float32 [B, 9+n_bins].
    Sets may differ in size; a set with fewer than 3 points raises ValueError
    like the per-set function."""
    import torch
    from . import engine

    sets = [np.asarray(p) for p in point_sets]
    for p in sets:
        if p.ndim != 2 or p.shape[1] < 2:
            raise ValueError(f"invalid shape {p.shape} (expected Nx3)")
        if p.shape[0] < 3:
            raise ValueError("too few points (<3)")
    dev = _device()
    if not sets:
        empty = np.zeros((0, 9 + n_bins), dtype=np.float32)
        return (empty, np.zeros((0, 6))) if return_eig else empty
    from . import ingest
    dims = 3 if sets[0].shape[1] >= 3 else 2
    sizes = np.array([p.shape[0] for p in sets], dtype=np.int64)
    sets = [ingest.as_device_points(p, dims) for p in sets]
    if any(p.dtype == np.float64 for p in sets):      # one wide set makes the batch wide (exact either way)
        sets = [p.astype(np.float64, copy=False) for p in sets]
    if np.all(sizes == sizes[0]):
        d_pts = ingest.stack_to_device(sets, dev, ingest.default_workers())
        res = engine.pca_batch(d_pts, None, n_bins, normalize, eig=return_eig)
    else:
        flat = np.concatenate(sets, axis=0)
        d_pts = torch.from_numpy(flat).to(dev)
        off = np.zeros(len(sets) + 1, dtype=np.int64)
        np.cumsum(sizes, out=off[1:])
        res = engine.pca_batch(d_pts, torch.from_numpy(off).to(dev), n_bins, normalize, eig=return_eig)
    out = res["out"].cpu().numpy()
    return (out, res["eig"].cpu().numpy()) if return_eig else out


def compute_pca_features(points: np.ndarray, n_bins: int = 8, normalize: bool = False) -> np.ndarray:
    """float32 [9 + n_bins]: lambda1..3, r1..3, ratio12, ratio23, trace,
    |projection on PC1| histogram (reference :59-145)."""
    if points.shape[0] < 3:
        raise ValueError("too few points (<3)")
    return compute_pca_features_batch([points], n_bins=n_bins, normalize=normalize)[0]


def save_descriptor(sample_id: str, vec: np.ndarray, out_dir: Path) -> Path:
    """pca_<id>.npy, float32 (reference :148-157)."""
    ensure_dir(out_dir)
    out_path = out_dir / f"pca_{sample_id}.npy"
    np.save(str(out_path), vec.astype(np.float32), allow_pickle=False)
    return out_path


def build_manifest_map(manifest_path: Path) -> dict:
    """id -> raw_path from a manifest CSV with those two columns (reference :164-174)."""
    import pandas as pd
    mapping = {}
    if manifest_path.exists():
        try:
            df = pd.read_csv(manifest_path)
            if "id" in df.columns and "raw_path" in df.columns:
                mapping = {str(i): str(r) for i, r in zip(df["id"], df["raw_path"])}
        except Exception as e:  # unreadable manifest is not fatal in the reference either
            logger.warning(f"Could not read manifest {manifest_path}: {e}")
    return mapping


def _blank_record(file_path: Path, manifest_map: dict) -> dict:
    return {"id": file_path.stem, "raw_path": manifest_map.get(file_path.stem, ""), "n_points": None,
            "lambda1": None, "lambda2": None, "lambda3": None, "r1": None, "r2": None, "r3": None,
            "ratio12": None, "ratio23": None, "trace": None, "proj_hist": None, "descriptor_path": ""}


def _log_error(reports_dir: Path, file_path: Path, e: Exception) -> None:
    try:
        ensure_dir(Path(reports_dir))
        with open(Path(reports_dir) / "read_errors.log", "a", encoding="utf-8") as fh:
            fh.write(f"{file_path}\t{type(e).__name__}\t{e}\n")
    except Exception:
        pass


def _fill_record(record: dict, n_pts: int, vec: np.ndarray, desc_path: Path) -> None:
    vals = vec[:9].astype(float).tolist()
    record.update(dict(zip(["lambda1", "lambda2", "lambda3", "r1", "r2", "r3", "ratio12", "ratio23", "trace"], vals)))
    record["n_points"] = n_pts
    record["proj_hist"] = vec[9:].astype(float).tolist()
    record["descriptor_path"] = str(desc_path.resolve())


def load_files(tasks: Iterable[tuple], workers: Optional[int] = None) -> dict:
    """Host half of process_files.  Files with the batch's common row count
    (float32/float64, C order) are read by native threads straight into one
    page-locked block; the others go through np.load one by one.  Files that
    cannot be read or hold fewer than 3 points get their failed result here
    (reference :194-230)."""
    import os
    from . import ingest

    tasks = list(tasks)
    workers = ingest.default_workers() if workers is None else int(workers)
    n = len(tasks)
    results: List[Optional[dict]] = [None] * n
    records: List[dict] = []
    for t in tasks:
        stem = os.path.splitext(os.path.basename(t[0]))[0]
        mm = t[6] if isinstance(t[6], dict) else {}
        records.append({"id": stem, "raw_path": mm.get(stem, ""), "n_points": None,
                        "lambda1": None, "lambda2": None, "lambda3": None, "r1": None, "r2": None, "r3": None,
                        "ratio12": None, "ratio23": None, "trace": None, "proj_hist": None, "descriptor_path": ""})
    rows = -1
    for t in tasks[:8]:          # the common row count, from the first header that can be read
        rows = ingest.npy_rows(t[0])
        if rows >= 3:
            break
    block = owner = None
    status = np.full(n, -1, dtype=np.int32)
    if rows >= 3 and n:
        block, owner = ingest.host_buffer((n, rows, 3), np.float32)
        status, _found = ingest.read_points_batch([t[0] for t in tasks], rows, block, workers)
    others = []                  # (k, float32 [N, 3]) of files the native reader left
    for k in np.nonzero(status != 0)[0].tolist():
        t = tasks[k]
        try:
            pts = ingest.read_npy_points(Path(t[0]))
            if int(pts.shape[0]) < 3:
                raise ValueError("too few points (<3) after loading")
            others.append((k, pts))
        except Exception as e:
            _log_error(Path(t[7]), Path(t[0]), e)
            results[k] = {"success": False, "record": records[k], "error": str(e)}
    return {"tasks": tasks, "results": results, "records": records, "rows": rows, "block": block, "owner": owner,
            "in_block": np.nonzero(status == 0)[0], "others": others, "workers": workers}


def finish_files(batch: dict) -> List[dict]:
    """Device half of process_files: one launch per (n_bins, normalize)
    combination and storage form present, pca_<id>.npy per set written by
    native threads, result dicts in task order."""
    import os
    import torch
    from . import engine, ingest

    tasks, results, records = batch["tasks"], batch["results"], batch["records"]
    workers = batch["workers"]
    done: List[tuple] = []       # (task indices, float32 [m, 9 + n_bins], point counts)

    def fail(ks, e):
        for k in ks:
            _log_error(Path(tasks[k][7]), Path(tasks[k][0]), e)
            results[k] = {"success": False, "record": records[k], "error": str(e)}

    in_block = batch["in_block"]
    if len(in_block):
        settings = {(int(tasks[k][3]), bool(tasks[k][4])) for k in in_block.tolist()}
        for (n_bins, normalize) in sorted(settings):
            ks = np.asarray([k for k in in_block.tolist() if (int(tasks[k][3]), bool(tasks[k][4])) == (n_bins, normalize)],
                            dtype=np.int64)
            try:
                host = batch["owner"]
                if len(ks) != host.shape[0]:
                    host = host[torch.from_numpy(ks)]
                d_pts = host.to(_device(), non_blocking=True)
                vecs = engine.pca_batch(d_pts, None, n_bins, normalize)["out"].cpu().numpy()
                done.append((ks.tolist(), vecs, [batch["rows"]] * len(ks)))
            except Exception as e:
                fail(ks.tolist(), e)
    groups = {}
    for k, pts in batch["others"]:
        groups.setdefault((int(tasks[k][3]), bool(tasks[k][4])), []).append((k, pts))
    for (n_bins, normalize), items in groups.items():
        try:
            vecs = compute_pca_features_batch([p for _, p in items], n_bins=n_bins, normalize=normalize)
            done.append(([k for k, _ in items], vecs, [int(p.shape[0]) for _, p in items]))
        except Exception as e:
            fail([k for k, _ in items], e)

    keys = ["lambda1", "lambda2", "lambda3", "r1", "r2", "r3", "ratio12", "ratio23", "trace"]
    for ks, vecs, counts in done:
        out_dirs = {tasks[k][2] for k in ks}
        real = {}
        for d in out_dirs:
            os.makedirs(d, exist_ok=True)
            real[d] = os.path.realpath(d)
        paths = [os.path.join(real[tasks[k][2]], f"pca_{records[k]['id']}.npy") for k in ks]
        status = ingest.write_rows_batch(paths, vecs, workers)
        as_list = vecs.astype(np.float64).tolist()
        for j, k in enumerate(ks):
            if status[j] != 0:
                fail([k], OSError(f"could not write {paths[j]} (status {int(status[j])})"))
                continue
            rec = records[k]
            row = as_list[j]
            rec.update(zip(keys, row[:9]))
            rec["n_points"] = counts[j]
            rec["proj_hist"] = row[9:]
            rec["descriptor_path"] = paths[j]
            results[k] = {"success": True, "record": rec, "error": ""}
    return results  # type: ignore[return-value]


def process_files(tasks: Iterable[tuple], workers: Optional[int] = None) -> List[dict]:
    """Batched process_file: reads every file on a host thread pool, runs one
    kernel launch over all loadable sets, saves pca_<id>.npy per set and
    returns one result dict per task, in task order, with the reference's
    layout ({"success", "record", "error"}, :194-262)."""
    return finish_files(load_files(tasks, workers))


def process_file(args_tuple):
    """One file -> {"success", "record", "error"} (reference :194-262)."""
    return process_files([args_tuple])[0]


def records_to_frame(records: List[dict], n_bins: int):
    """DataFrame with the exact column order of descriptors_table_pca.csv
    (reference :383-414).  Records of sets that went through (every base
    column present and set) are laid out column by column; anything else is
    built row by row as the reference does -- same frame either way."""
    import pandas as pd
    cols = column_names(n_bins)
    if records and all(rec.get("n_points") is not None and all(c in rec for c in BASE_COLS) for rec in records):
        data = {c: [rec[c] for rec in records] for c in BASE_COLS}
        hists = []
        for rec in records:
            hist = rec.get("proj_hist") or [0.0] * n_bins
            if len(hist) != n_bins:
                hist = (list(hist) + [0.0] * max(0, n_bins - len(hist)))[:n_bins]
            hists.append(hist)
        for i in range(n_bins):
            data[f"proj_hist_bin_{i}"] = [float(h[i]) for h in hists]
        return pd.DataFrame(data, columns=cols)
    rows = []
    for rec in records:
        row = {c: rec.get(c, np.nan) for c in BASE_COLS}
        row["id"] = rec.get("id", "")
        row["raw_path"] = rec.get("raw_path", "")
        hist = rec.get("proj_hist") or [0.0] * n_bins
        hist = list(hist) + [0.0] * max(0, n_bins - len(hist))
        for i in range(n_bins):
            row[f"proj_hist_bin_{i}"] = float(hist[i])
        rows.append(row)
    return pd.DataFrame(rows, columns=cols)


def list_npy_paths(root) -> List[str]:
    """The .npy files below `root` as strings, in the order of list_npy_files."""
    found: List[str] = []
    stack = [str(root)]
    while stack:
        d = stack.pop()
        try:
            with os.scandir(d) as it:
                entries = list(it)
        except PermissionError:
            continue
        for e in entries:
            try:
                if e.is_dir():
                    if not e.is_symlink():
                        stack.append(e.path)
                elif e.name.endswith(".npy") and e.is_file():
                    found.append(e.path)
            except OSError:
                continue
    found.sort()
    return found


def list_npy_files(root: Path) -> List[Path]:
    """What sorted(p for p in root.rglob("*.npy") if p.is_file()) gives, ordered by str(path) (reference
    :312-316), from os.scandir: directories that are symbolic links are not entered, links to files count as
    files, a directory that may not be read is skipped.  (pathlib takes 7 us per file for the same list; main()
    works on list_npy_paths, the strings, and saves the 4 us per file that a Path object costs.)"""
    return [Path(p) for p in list_npy_paths(root)]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Extract PCA-based 3D descriptors from sampled point clouds (MI355X).")
    p.add_argument("--sampled_dir", type=str, default="../../data/processed/sampled")
    p.add_argument("--out_dir", type=str, default="../../data/processed/descriptors")
    p.add_argument("--n_bins", type=int, default=8)
    p.add_argument("--max_samples", type=int, default=None)
    p.add_argument("--fast", action="store_true", help="process the first 20 files only")
    p.add_argument("--normalize", action="store_true")
    p.add_argument("--workers", type=int, default=None, help="reader threads on the host (default: the cores, at most 16)")
    p.add_argument("--batch_files", type=int, default=65536, help="files per device batch")
    p.add_argument("--overwrite", action="store_true")
    p.add_argument("--processed_manifest", type=str, default="")
    p.add_argument("--reports_dir", type=str, default="../../reports")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--verbose", action="store_true")
    return p.parse_args(argv)


def main(argv=None) -> int:
    """CLI with the reference's flags, outputs and exit codes (:288-449):
    0 = ok / nothing to do, 2 = sampled_dir missing, 3 = every file failed."""
    args = parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO,
                        format="%(asctime)s [%(levelname)s] %(message)s")
    sampled_dir, out_dir, reports_dir = Path(args.sampled_dir), Path(args.out_dir), Path(args.reports_dir)
    ensure_dir(out_dir)
    ensure_dir(reports_dir)
    manifest_map = {}
    if args.processed_manifest:
        if Path(args.processed_manifest).exists():
            manifest_map = build_manifest_map(Path(args.processed_manifest))
    elif (sampled_dir.parent / "manifest.csv").exists():
        manifest_map = build_manifest_map(sampled_dir.parent / "manifest.csv")
    if not sampled_dir.exists():
        logger.error(f"Sampled directory not found: {sampled_dir}")
        return 2
    files = list_npy_paths(sampled_dir)      # str(p) of list_npy_files' entries, without the Path objects
    if str(sampled_dir) == ".":               # (pathlib drops the "./" that os.scandir(".") puts in front)
        files = [f[2:] for f in files]
    limit = 20 if args.fast else args.max_samples
    if limit is not None:
        files = files[:limit]
    if not files:
        logger.warning("No .npy files found; exiting.")
        return 0
    tasks = [(p, str(sampled_dir), str(out_dir), int(args.n_bins), bool(args.normalize), bool(args.overwrite),
              manifest_map, str(reports_dir)) for p in files]
    from . import ingest
    workers = int(args.workers) if args.workers else None
    step = max(1, int(args.batch_files))
    results: List[dict] = []
    # the next batch is read while the current one is on the device
    for batch in ingest.prefetch(lambda chunk: load_files(chunk, workers),
                                 (tasks[i:i + step] for i in range(0, len(tasks), step))):
        results.extend(finish_files(batch))
    records = [r["record"] for r in results if r.get("success")]
    if not records:
        logger.error("No descriptors produced (all files failed). Check reports/read_errors.log for details.")
        return 3
    df = records_to_frame(records, int(args.n_bins))
    df.to_csv(out_dir / "descriptors_table_pca.csv", index=False)
    stats = df.select_dtypes(include=[np.number]).agg(["mean", "std"]).transpose().reset_index()
    stats.rename(columns={"index": "feature"}, inplace=True)
    stats.to_csv(reports_dir / "pca_feature_stats.csv", index=False)
    mode = "fast" if args.fast else "full"
    print(f"PCA descriptor extraction finished. Mode: {mode}. Samples processed: {len(records)}. "
          f"Descriptors saved to {out_dir}.")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
