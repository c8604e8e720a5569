#!/usr/bin/env python3
"""Throughput of the Hu + PCA feature path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5|f1|p1]

`--gpus N` with N > 1 works as typed: a process that was not started by a
launcher (no WORLD_SIZE in its environment) starts
`python -m torch.distributed.run --nproc-per-node N` on this same file as a
child, before anything touches the GPU, and relays its output.  Under a
launcher (the form the round driver uses) every rank runs main() directly.

One step = one pass of the hot path over the batch that is resident in HBM:

  c3 (default)  per GPU 125 000 synthetic 512x512 uint8 shape images (the 1M-image
                configuration split over 8 GPUs; weak scaling: every rank holds
                one such shard): fused binarise + raw moments + Hu + contour-point
                PCA in one kernel launch, then (N > 1) one all-gather of the
                [rows, 24] float32 feature block over RCCL.
  c2            100 000 x 256x256, Hu only.
  c4            1 000 000 point sets of 1024 x 3 float32 split over the ranks
                (strong scaling), PCA descriptor.
  c5            the tail configuration: 64 images of 8192x8192 (Hu, tiled path)
                paired with 64 clouds of 10^3 .. 4*10^6 points (PCA descriptor,
                CSR batch); strong scaling: images in contiguous shards, clouds
                assigned largest-first (LPT) by point count.
  f1            20 000 gray 224x224 views per GPU (dark synthetic shapes on a
                light background + noise): silhouette preprocessing + Hu, the
                stage in front of the headline path.
  p1            2 000 sampled clouds per GPU -> 12 rendered views -> cropped, cleaned
                silhouettes -> Hu -> per-sample mean/std (the flags of the
                reference's production run: --preprocess_crop --use_silhouette).

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events
that libfx records on the launch stream around its own kernel; `cpu_baseline`
times the CPU oracle on this host's cores on a bounded sample of the same
images (rank 0, N = 1 only); `verified` says whether a strided sample of the
last step's features (the last unit included) equals the CPU oracle's, checked
after the timed region.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
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
    "c5": dict(kind="tail", total=64, H=8192, W=8192, bytes_per_unit=8192 * 8192, scaling="strong",
               desc="64 x 8192x8192 u8 shapes (Hu, tiled path) + 64 clouds of 1e3..4e6 points (PCA, CSR batch); "
                    "images in contiguous shards, clouds LPT-assigned"),
    "f1": dict(kind="sil+hu", per_gpu=20_000, H=224, W=224, bytes_per_unit=224 * 224, scaling="weak",
               desc="20k gray 224x224 views per GPU; silhouette preprocessing (blur, Otsu, morphology, crop, "
                    "resize, component filter) + Hu"),
    "p1": dict(kind="views+hu", per_gpu=2_000, N=1024, bytes_per_unit=1024 * 3 * 4, scaling="weak",
               desc="2k sampled clouds of 1024x3 f32 per GPU; 12 rendered 224x224 views each (the reference's matplotlib "
                    "renderer) -> cropped silhouettes (--preprocess_crop --use_silhouette) -> Hu -> per-sample mean/std, all in HBM"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--units", type=int, default=None, help="override units per GPU (smaller smoke runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=None, help="units in the CPU baseline sample")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous over gloo on the CPU, one tiny all-gather, no GPU work: exercises the launcher path")
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(args, argv) -> list:
    """The child command for `--gpus N` typed without a launcher."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + list(argv)


def relaunch_under_launcher(args, argv) -> int:
    """Starts N fresh ranks as child processes (this process has not touched
    the GPU and never will) and relays rank 0's JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(launcher_command(args, argv), env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


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
            "cores_available": os.cpu_count(),
            "cores_note": "os.cpu_count() is the whole machine; a one-GPU box grants 16 CPUs, and the pool uses min(16, that)",
            "kind": "port",
            "sample": f"{units} units of the benchmark's own synthetic data, {wall:.1f} s wall on {len(jobs)} worker processes "
                      f"(C restatement of cv2.moments/HuMoments + NumPy cov/eigh/histogram per unit"
                      + ("; NumPy/scipy restatement of the reference's OpenCV silhouette branch" if kind == "sil+hu" else "")
                      + ("; NumPy restatement of the matplotlib renderer, 12 views per cloud, restated OpenCV-branch preprocessing"
                         if kind == "views+hu" else "")
                      + ("; 8192x8192 images only, the clouds are not part of this sample" if kind == "tail" else "") + ")"}


def tail_cloud_sizes(n: int = 64, seed: int = 42):
    """Point counts of the tail configuration's clouds: log-uniform in [1e3, 4e6]."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return np.exp(rng.uniform(np.log(1e3), np.log(4e6), size=n)).astype(np.int64)


def sample_rows(units: int, want: int):
    import numpy as np
    return np.unique(np.linspace(0, units - 1, min(want, units)).round().astype(np.int64))


def hist_close(got, ref, n) -> bool:
    """Histogram counts equal up to edge ties (a point within an ulp of a bin edge)."""
    import numpy as np
    return bool((np.abs(got.astype(np.float64) - ref) * n).max() <= 2.5)


def verify(kind: str, state: dict, feats) -> dict:
    """Compares a strided sample of the last step's features with the CPU
    oracle (outside the timed region)."""
    import numpy as np
    import torch
    from oracle import cpu_ref

    c = cpu_ref.load_c_oracle()
    ok, rows = True, 0
    if kind in ("hu", "hu+cpca"):
        data = state["data"]
        idx = sample_rows(data.shape[0], 256 if kind == "hu" else 128)
        host = data[torch.from_numpy(idx).to(data.device)].cpu().numpy()
        got = feats[torch.from_numpy(idx).to(feats.device)].cpu().numpy()
        ref = c.hu_batch(host, thresh=127, on_value=255)["hulog7"]
        ok &= bool(np.all(np.abs(got[:, :7] - ref) <= np.spacing(np.abs(ref))))
        if kind == "hu+cpca":
            for k in range(len(idx)):
                want = cpu_ref.contour_pca_features(host[k], 127, 8)
                n = cpu_ref.contour_points(host[k]).shape[0]
                ok &= bool(np.allclose(got[k, 7:16], want[:9], rtol=1e-6, atol=1e-30)) and hist_close(got[k, 16:], want[9:], n)
        rows = len(idx)
    elif kind == "pca":
        data = state["data"]
        idx = sample_rows(data.shape[0], 256)
        host = data[torch.from_numpy(idx).to(data.device)].cpu().numpy().astype(np.float64)
        got = feats[torch.from_numpy(idx).to(feats.device)].cpu().numpy()
        for k in range(len(idx)):
            want = cpu_ref.compute_pca_features(host[k], 8, False)
            ok &= bool(np.allclose(got[k, :9], want[:9], rtol=1e-6, atol=1e-30)) and hist_close(got[k, 9:], want[9:], host.shape[1])
        rows = len(idx)
    elif kind == "sil+hu":
        data = state["data"]
        idx = sample_rows(data.shape[0], 48)
        host = data[torch.from_numpy(idx).to(data.device)].cpu().numpy()
        got = feats[torch.from_numpy(idx).to(feats.device)].cpu().numpy()
        for k in range(len(idx)):
            mask, _ = cpu_ref.silhouette_cv(host[k], 224, 0.05, True)
            ref = c.hu_batch(mask[None])["hulog7"][0]
            ok &= bool(np.all(np.abs(got[k] - ref) <= np.spacing(np.abs(ref))))
        rows = len(idx)
    elif kind == "views+hu":
        stamp = np.array(importlib.import_module("image-processing_amd.marker_stamps").STAMPS[2.0], dtype=np.uint8)
        data = state["data"]
        idx = sample_rows(data.shape[0], 3)
        got = feats[torch.from_numpy(idx).to(feats.device)].cpu().numpy()
        for k, i in enumerate(idx):
            p64 = data[int(i)].cpu().numpy().astype(np.float64)
            mins, maxs, cen, _ = cpu_ref.bbox_and_center(p64)
            r = []
            for pose in cpu_ref.compute_camera_poses(12, mins, maxs, cen):
                view = cpu_ref.render_view(p64, pose, 224, stamp)
                _, mask = cpu_ref.preprocess_image_cv(view, 224, 0.05, "white", True)
                r.append(c.hu_batch(mask[None])["hulog7"][0])
            r = np.stack(r).astype(np.float32)
            want = np.concatenate([r.mean(axis=0), r.std(axis=0)])
            ok &= bool(np.allclose(got[k], want, rtol=1e-5, atol=1e-6))
        rows = len(idx)
    elif kind == "tail":
        hu, pca = feats
        imgs = state["data"]
        for i in sorted({0, imgs.shape[0] - 1}):
            ref = c.hu_batch(imgs[i:i + 1].cpu().numpy(), thresh=127, on_value=255, exact=True)["hulog7"][0]
            ok &= bool(np.all(np.abs(hu[i].cpu().numpy() - ref) <= np.spacing(np.abs(ref))))
            rows += 1
        off = state["cloud_off"].cpu().numpy()
        sizes = np.diff(off)
        pick = sorted({int(np.argmin(sizes)), int(np.argmax(sizes)), 0, len(sizes) - 1})
        for j in pick:
            pts = state["flat"][off[j]:off[j + 1]].cpu().numpy().astype(np.float64)
            want = cpu_ref.compute_pca_features(pts, 8, False)
            got = pca[j].cpu().numpy()
            ok &= bool(np.allclose(got[:9], want[:9], rtol=1e-6, atol=1e-30)) and hist_close(got[9:], want[9:], pts.shape[0])
            rows += 1
    return {"verified": bool(ok), "checked_rows": int(rows)}


def dry_run(args) -> None:
    """Launcher rehearsal on the CPU: rendezvous over gloo, one all-gather."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    fxdist = importlib.import_module("image-processing_amd.dist")
    rows = 5
    local = torch.full((rows, 24), float(rank), dtype=torch.float32)
    table = fxdist.all_gather_rows(local, rows * world)
    want = torch.arange(world, dtype=torch.float32).repeat_interleave(rows)
    assert torch.equal(table[:, 0], want)
    sizes = tail_cloud_sizes()
    assign = fxdist.lpt_assign([float(s) for s in sizes], world)
    mine = assign[rank]
    tagged = fxdist.all_gather_tagged_rows(torch.tensor([[float(i), float(sizes[i])] for i in mine]).reshape(len(mine), 2), mine,
                                           max(len(a) for a in assign), len(sizes))
    assert torch.equal(tagged[:, 0], torch.arange(len(sizes), dtype=torch.float32))
    report = fxdist.describe_ranks(1.0 + rank, "cpu")
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rows": int(table.shape[0]), "lpt_units_rank0": len(mine), **report}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch_under_launcher(args, argv))
    if args.dry_run:
        return dry_run(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    state = {}

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
    elif kind == "tail":
        import numpy as np
        total = args.units or wl["total"]
        lo, hi = fxdist.shard_range(total, rank, world)
        units = hi - lo
        data = torch.empty((units, wl["H"], wl["W"]), dtype=torch.uint8, device=dev)
        for a in range(units):
            eng.synth_shapes(1, wl["H"], wl["W"], seed=42, first_index=lo + a, device=dev, out=data[a:a + 1])
        sizes = tail_cloud_sizes(total)
        assign = fxdist.lpt_assign([float(s) for s in sizes], world)
        mine = assign[rank]
        slot = max(len(a) for a in assign)              # padded rows per rank in the feature all-gather
        clouds = [eng.synth_points(1, int(sizes[i]), seed=43, first_index=i, device=dev).reshape(-1, 3) for i in mine]
        flat = torch.cat(clouds) if clouds else torch.empty((0, 3), dtype=torch.float32, device=dev)
        del clouds
        cloud_off = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes[mine])]).astype(np.int64)).to(dev)
        state.update(flat=flat, cloud_off=cloud_off)
        wl = dict(wl)
        wl["cloud_points_total"] = int(sizes.sum())
        wl["cloud_bytes_this_rank"] = int(sizes[mine].sum()) * 12
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
    state["data"] = data
    torch.cuda.synchronize()

    first_ms = []   # kernel time of the first of two timed library calls in a step
    gather = fxdist.GatherPipeline(depth=2)

    def step():
        if kind == "pca":
            feats = eng.pca_batch(data, None, 8, False)["out"]
        elif kind == "hu":
            feats = eng.hu_batch(data, thresh=127, on_value=255)["hulog7"]
        elif kind == "views+hu":
            feats = hu2d.hu_from_clouds_device(flat, cloud_off, 12, 224, preprocess_crop=True, use_silhouette=True)
            first_ms.append(0.0)
        elif kind == "sil+hu":
            feats = eng.silhouette_hu_batch(data, 224, 0.05, True)["hulog7"]   # Hu summed from the silhouette on chip
            first_ms.append(eng.profile_last_ms())
        elif kind == "tail":
            # (Launch timing is switched off around the cloud call so that the event pairs read back after the
            # loop are the image kernel's.  A second stream for the clouds gains nothing: the image kernel's blocks
            # stay on the chip until its last band is done, and the clouds' launches wait for them.)
            hu = eng.hu_batch(data, thresh=127, on_value=255)["hulog7"] if units else torch.empty((0, 7), device=dev)
            if state.get("timing"):
                eng.profile(False)
            pca = eng.pca_batch(flat, cloud_off, 8, False)["out"]
            if state.get("timing"):
                eng.profile(True)
            if world > 1:
                # one all-gather per feature block: image rows are equal contiguous shards (padded), cloud rows
                # carry their unit index because the LPT assignment is not contiguous
                fxdist.all_gather_rows(hu, total)
                fxdist.all_gather_tagged_rows(pca, mine, slot, total)
            return hu, pca
        else:
            r = eng.hu_batch(data, thresh=127, on_value=255, contour_bins=8)
            feats = torch.cat([r["hulog7"], r["cpca"]], dim=1)
        if world > 1:
            if wl["scaling"] == "strong":
                fxdist.all_gather_rows(feats, total)
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
    state["timing"] = True
    kernel_ms = []
    feats = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats = step()
        # reading the event pair waits for this step's kernel; the next step is
        # enqueued right after, which costs one launch latency per step
        if kind != "tail":   # (the tail step's image kernel times are read back after the loop)
            kernel_ms.append(first_ms[-1] if kind == "sil+hu" else eng.profile_last_ms())
    table = gather.drain()      # every step's all-gather has completed inside the timed region
    if world > 1 and wl["scaling"] == "weak":
        assert table is not None and table.shape[0] == total
    fence()
    elapsed = time.perf_counter() - t0
    if kind == "tail":   # the image kernel of the last steps (the library keeps 32 timed launches)
        kernel_ms = [eng.profile_ms_back(k) for k in range(min(args.steps, 32))] if units else [0.0]
    if kind == "views+hu":   # the renderer's kernel, timed on one extra call after the timed region
        rvm = importlib.import_module("image-processing_amd.render_views")
        rvm.render_views_device(flat, cloud_off, 12, 224)
        kernel_ms = [eng.profile_last_ms()]
    eng.profile(False)
    # What follows describes the run; a failure in it must not cost the measurement (every rank takes the same path:
    # the collectives inside are entered by all or by none).
    own_ms = elapsed / args.steps * 1e3
    gather_ms = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    try:
        ranks = fxdist.describe_ranks(own_ms, torch.cuda.get_device_name(dev))
    except Exception as e:   # pragma: no cover
        ranks = {"ranks": world, "ranks_error": f"{type(e).__name__}: {e}"}
    if world > 1:
        try:
            # the exchange step on its own: one all-gather of a step's feature rows with nothing beside it
            rows = feats[0] if kind == "tail" else feats
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fence()
            e0.record()
            if kind != "tail" and wl["scaling"] == "weak":
                gather.submit(rows)
                gather.drain()
            else:
                fxdist.all_gather_rows(rows, total)
            e1.record()
            torch.cuda.synchronize()
            gather_ms = float(e0.elapsed_time(e1))
        except Exception as e:   # pragma: no cover
            ranks["all_gather_error"] = f"{type(e).__name__}: {e}"

    out = None
    if rank == 0:
        kavg = sum(kernel_ms) / len(kernel_ms)
        # a call that ran its units as several launches of the kernel timed one of them
        launch_units = (eng.profile_last_units() if kind == "sil+hu" else 0) or units
        achieved = launch_units * wl["bytes_per_unit"] / (kavg * 1e-3) / 1e9 if kavg > 0 else 0.0
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
                       "views+hu": "clouds/sec rendered views + Hu features",
                       "tail": "images/sec Hu+PCA features (image + cloud pairs)"}.get(kind, "images/sec Hu+PCA features"),
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
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc, separate run)" if traffic is not None else None,
                         "kernel": {"pca": "pca_sets_full_kernel", "hu": "hu_band_kernel", "hu+cpca": "hu_band_kernel",
                                    "tail": "hu_band_kernel", "sil+hu": "sil_mask_kernel",
                                    "views+hu": "render_views_kernel"}[kind],
                         "kernel_ms": kavg, "units_per_launch": launch_units},
        }
        out.update(ranks)
        out["all_gather_ms"] = gather_ms
        if kind == "tail":
            out["config"]["cloud_points_total"] = wl["cloud_points_total"]
            out["config"]["cloud_bytes_rank0"] = wl["cloud_bytes_this_rank"]
        if not args.no_verify:
            out.update(verify(kind, state, feats))
        if world == 1 and not args.no_cpu_baseline:
            cores = min(16, os.cpu_count() or 1)
            n_s = args.cpu_sample or (cores * {"pca": 12000, "hu+cpca": 2400, "hu": 12000, "sil+hu": 400, "views+hu": 16,
                                               "tail": 1}[kind])
            n_s = min(n_s, units)
            sample = data[:n_s].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(kind, sample, cores)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
