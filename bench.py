#!/usr/bin/env python3
"""Throughput of the Hu + PCA feature path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4]

One step = one pass of the hot path over the batch that is resident in HBM:

  c3 (default)  per GPU 125 000 synthetic 512x512 uint8 shape images (the 1M-image
                configuration split over 8 GPUs; weak scaling: every rank holds
                one such shard): fused binarise + raw moments + Hu + contour-point
                PCA in one kernel launch, then (N > 1) one all-gather of the
                [rows, 24] float32 feature block over RCCL.
  c2            100 000 x 256x256, Hu only.
  c4            1 000 000 point sets of 1024 x 3 float32 split over the ranks
                (strong scaling), PCA descriptor.
  f1            20 000 gray 224x224 views per GPU (dark synthetic shapes on a
                light background + noise): silhouette preprocessing + Hu, the
                stage in front of the headline path.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events
that libfx records on the launch stream around its own kernel; `cpu_baseline`
times the CPU oracle on this host's cores on a bounded sample of the same
images (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak

WORKLOADS = {
    # name: (units per GPU or total, H, W / N, algorithmic bytes per unit)
    "c3": dict(kind="hu+cpca", per_gpu=125_000, H=512, W=512, bytes_per_unit=512 * 512, scaling="weak",
               desc="1M x 512x512 u8 shapes / 8 GPUs = 125k per GPU; Hu + contour-point PCA"),
    "c2": dict(kind="hu", per_gpu=100_000, H=256, W=256, bytes_per_unit=256 * 256, scaling="weak",
               desc="100k x 256x256 u8 shapes per GPU; Hu only"),
    "c4": dict(kind="pca", total=1_000_000, N=1024, bytes_per_unit=1024 * 3 * 4, scaling="strong",
               desc="1M point sets of 1024x3 f32 over all GPUs; PCA descriptor"),
    "f1": dict(kind="sil+hu", per_gpu=20_000, H=224, W=224, bytes_per_unit=224 * 224, scaling="weak",
               desc="20k gray 224x224 views per GPU; silhouette preprocessing (blur, Otsu, morphology, crop, "
                    "resize, component filter) + Hu"),
    "p1": dict(kind="views+hu", per_gpu=2_000, N=1024, bytes_per_unit=1024 * 3 * 4, scaling="weak",
               desc="2k sampled clouds of 1024x3 f32 per GPU; 12 rendered 224x224 views each (the reference's matplotlib "
                    "renderer) -> preprocessing -> Hu -> per-sample mean/std, all in HBM"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--units", type=int, default=None, help="override units per GPU (smaller smoke runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=None, help="units in the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(kind: str, sample, cores: int) -> dict:
    """Times the CPU oracle on `sample` (numpy array on the host) with a pool
    of `cores` spawned workers, the reference's parallel model
    (multiprocessing.Pool over files)."""
    import multiprocessing as mp
    import numpy as np

    n = sample.shape[0]
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        path = os.path.join(tmp, "sample.npy")
        np.save(path, sample)
        per = -(-n // cores)
        jobs = [(kind, path, lo, min(n, lo + per)) for lo in range(0, n, per)]
        ctx = mp.get_context("spawn")
        with ctx.Pool(processes=len(jobs)) as pool:
            from oracle import cpu_baseline as cb
            pool.map(cb.warm, [j[:2] for j in jobs])  # import + library build outside the timed region
            t0 = time.perf_counter()
            done = pool.map(cb.run, jobs)
            wall = time.perf_counter() - t0
    units = sum(d[0] for d in done)
    return {"value": units / wall, "unit": {"pca": "sets/s", "views+hu": "clouds/s"}.get(kind, "images/s"), "cores": len(jobs),
            "kind": "port",
            "sample": f"{units} units of the benchmark's own synthetic data, {wall:.1f} s wall on {len(jobs)} worker processes "
                      f"(C restatement of cv2.moments/HuMoments + NumPy cov/eigh/histogram per unit"
                      + ("; PIL + scipy.ndimage silhouette as in the reference" if kind == "sil+hu" else "")
                      + ("; NumPy restatement of the matplotlib renderer, 12 views per cloud, PIL + scipy.ndimage preprocessing"
                         if kind == "views+hu" else "") + ")"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    eng = importlib.import_module("image-processing_amd.engine")
    fxdist = importlib.import_module("image-processing_amd.dist")
    wl = WORKLOADS[args.workload]
    kind = wl["kind"]

    # ---------------- resident synthetic batch ------------------------------
    if kind == "views+hu":
        hu2d = importlib.import_module("image-processing_amd.extract_2d_hu")
        units = args.units or wl["per_gpu"]
        total = units * world
        lo = rank * units
        data = eng.synth_points(units, wl["N"], seed=42, first_index=lo, device=dev)
        cloud_off = torch.arange(units + 1, dtype=torch.int64, device=dev) * wl["N"]
        flat = data.reshape(units * wl["N"], 3)
    elif kind == "pca":
        total = args.units * world if args.units else wl["total"]
        lo, hi = fxdist.shard_range(total, rank, world)
        units = hi - lo
        data = eng.synth_points(units, wl["N"], seed=42, first_index=lo, device=dev)
    else:
        units = args.units or wl["per_gpu"]
        total = units * world
        lo = rank * units
        data = torch.empty((units, wl["H"], wl["W"]), dtype=torch.uint8, device=dev)
        chunk = 8192
        for a in range(0, units, chunk):  # chunked so that a launch stays short
            b = min(units, a + chunk)
            eng.synth_shapes(b - a, wl["H"], wl["W"], seed=42, first_index=lo + a, device=dev, out=data[a:b])
            if kind == "sil+hu":  # rendered-view look-alike: dark shape, light background, sensor-like noise
                g = torch.Generator(device=dev)
                g.manual_seed(42 + lo + a)
                v = 235.0 - data[a:b].to(torch.float32) * (195.0 / 255.0)
                v += torch.randn(v.shape, generator=g, device=dev) * 8.0
                data[a:b] = v.clamp_(0, 255).round_().to(torch.uint8)
    torch.cuda.synchronize()

    sil_ms = []
    gather = fxdist.GatherPipeline(depth=2)

    def step():
        if kind == "pca":
            feats = eng.pca_batch(data, None, 8, False)["out"]
        elif kind == "hu":
            feats = eng.hu_batch(data, thresh=127, on_value=255)["hulog7"]
        elif kind == "views+hu":
            feats = hu2d.hu_from_clouds_device(flat, cloud_off, 12, 224)
            sil_ms.append(0.0)
        elif kind == "sil+hu":
            masks = eng.silhouette_batch(data, 224, 0.05, True)["mask"]
            sil_ms.append(eng.profile_last_ms())   # the Hu launch below overwrites the event pair
            feats = eng.hu_batch(masks)["hulog7"]
        else:
            r = eng.hu_batch(data, thresh=127, on_value=255, contour_bins=8)
            feats = torch.cat([r["hulog7"], r["cpca"]], dim=1)
        if world > 1:
            if wl["scaling"] == "strong":
                feats = fxdist.all_gather_rows(feats, total)
            else:
                # equal shards: the collective of this step runs while the next step's kernels do
                # (at most two in flight; all are waited for before the timed region ends)
                gather.submit(feats)
        return feats

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gather.drain()
    fence()
    eng.profile(True)
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # reading the event pair waits for this step's kernel; the next step is
        # enqueued right after, which costs one launch latency per step
        kernel_ms.append(sil_ms[-1] if kind == "sil+hu" else eng.profile_last_ms())
    table = gather.drain()      # every step's all-gather has completed inside the timed region
    if world > 1 and wl["scaling"] == "weak":
        assert table is not None and table.shape[0] == total
    fence()
    elapsed = time.perf_counter() - t0
    if kind == "views+hu":   # the renderer's kernel, timed on one extra call after the timed region
        rvm = importlib.import_module("image-processing_amd.render_views")
        rvm.render_views_device(flat, cloud_off, 12, 224)
        kernel_ms = [eng.profile_last_ms()]
    eng.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        kavg = sum(kernel_ms) / len(kernel_ms)
        achieved = units * wl["bytes_per_unit"] / (kavg * 1e-3) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        unit = "sets/s" if kind == "pca" else ("clouds/s" if kind == "views+hu" else "images/s")
        out = {
            "metric": {"pca": "point sets/sec PCA features",
                       "views+hu": "clouds/sec rendered views + Hu features"}.get(kind, "images/sec Hu+PCA features"),
            "value": total * args.steps / elapsed, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": wl["scaling"], "vs_baseline": None,
            "dtype": "f64" if kind == "pca" else ("u32" if kind in ("sil+hu", "views+hu") else "int64"),
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "units_per_gpu": units, "total_units": total,
                       "parallelism": f"dp{world} (units sharded, all-gather of features)" if world > 1 else "single GPU",
                       "bytes_per_unit": wl["bytes_per_unit"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": {"pca": "pca_sets_reg_kernel", "hu": "hu_band_kernel", "hu+cpca": "hu_band_kernel",
                                    "sil+hu": "sil_mask_kernel", "views+hu": "render_views_kernel"}[kind],
                         "kernel_ms": kavg, "units_per_launch": units},
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = min(16, os.cpu_count() or 1)
            n_s = args.cpu_sample or (cores * {"pca": 12000, "hu+cpca": 2400, "hu": 12000, "sil+hu": 400, "views+hu": 16}[kind])
            n_s = min(n_s, units)
            sample = data[:n_s].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(kind, sample, cores)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
