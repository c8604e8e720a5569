"""Kernel timing harness for development: times hu_batch / pca_batch variants with HIP events."""
import importlib, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
eng = importlib.import_module("image-processing_amd.engine")
dev = torch.device("cuda", 0)

def time_hu(B, H, W, contour, reps=8, thresh=127):
    imgs = eng.synth_shapes(B, H, W, seed=42, device=dev)
    kw = dict(contour_bins=8) if contour else {}
    for _ in range(2): eng.hu_batch(imgs, thresh=thresh, **kw)
    torch.cuda.synchronize()
    eng.profile(True); ms = []
    for _ in range(reps):
        eng.hu_batch(imgs, thresh=thresh, **kw); ms.append(eng.profile_last_ms())
    eng.profile(False)
    ms.sort(); med = ms[len(ms)//2]
    return med, B*H*W/med/1e6

def time_pca(B, N, reps=8):
    pts = eng.synth_points(B, N, seed=42, device=dev)
    for _ in range(2): eng.pca_batch(pts)
    torch.cuda.synchronize()
    eng.profile(True); ms = []
    for _ in range(reps):
        eng.pca_batch(pts); ms.append(eng.profile_last_ms())
    eng.profile(False)
    ms.sort(); med = ms[len(ms)//2]
    return med, B*N*12/med/1e6

def gray_views(B, H, W, seed=42):
    """Rendered-view look-alikes: dark synthetic shapes on a light background + noise."""
    shapes = eng.synth_shapes(B, H, W, seed=seed, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    out = torch.empty_like(shapes)
    step = max(1, (1 << 28) // (H * W))
    for i in range(0, B, step):
        s = shapes[i:i + step].to(torch.float32)
        v = 235.0 - s * (195.0 / 255.0) + torch.randn(s.shape, generator=g, device=dev) * 8.0
        out[i:i + step] = v.clamp_(0, 255).round_().to(torch.uint8)
    return out

def time_sil(B, H, W, res, reps=5):
    imgs = gray_views(B, H, W)
    for _ in range(2): eng.silhouette_batch(imgs, res, 0.05, True)
    torch.cuda.synchronize()
    eng.profile(True); ms = []
    for _ in range(reps):
        eng.silhouette_batch(imgs, res, 0.05, True); ms.append(eng.profile_last_ms())
    eng.profile(False)
    ms.sort(); med = ms[len(ms)//2]
    return med, B*H*W/med/1e6

def time_fps(B, M, n, reps=3):
    import numpy as np
    pts = eng.synth_points(B, M, seed=42, device=dev).reshape(B * M, 3).contiguous()
    off = np.arange(B + 1, dtype=np.int64) * M
    start = np.zeros(B, dtype=np.int64)
    eng.center_scale(pts, off); eng.fps(pts, off, n, start)
    torch.cuda.synchronize()
    eng.profile(True); a = []; b = []
    for _ in range(reps):
        r = eng.center_scale(pts, off); a.append(eng.profile_last_ms())
        eng.fps(r["points"], off, n, start); b.append(eng.profile_last_ms())
    eng.profile(False)
    a.sort(); b.sort()
    return a[len(a)//2], b[len(b)//2]

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "hu"
    if what == "h2d":
        # the rate a caller with host buffers sees: page-locked memory -> device -> features
        B, H, W = 8192, 512, 512
        host = eng.synth_shapes(B, H, W, seed=42, device=dev).cpu().pin_memory()
        for _ in range(2):
            d = host.to(dev, non_blocking=True); eng.hu_batch(d, thresh=127, contour_bins=8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            d = host.to(dev, non_blocking=True); eng.hu_batch(d, thresh=127, contour_bins=8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"h2d pinned -> device -> Hu + contour PCA, {B} x {H}x{W}: {dt*1e3:.2f} ms  {B/dt/1e6:.3f} M img/s  {B*H*W/dt/1e9:.1f} GB/s over PCIe", flush=True)
    elif what in ("c5", "c5p", "c5n"):
        # c5p: the point clouds only; c5n: the same with normalize (the launch chain instead of the persistent launch)
        import numpy as np
        if what == "c5":
            ms, gbps = time_hu(64, 8192, 8192, False, reps=5)
            print(f"c5 hu 64 x 8192x8192 (tiled): {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
        kw = {"normalize": True} if what == "c5n" else {}
        rng = np.random.default_rng(5)
        sizes = np.exp(rng.uniform(np.log(1e4), np.log(4e6), 64)).astype(np.int64)
        off = np.zeros(65, dtype=np.int64); np.cumsum(sizes, out=off[1:])
        pts = eng.synth_points(1, int(off[-1]), seed=5, device=dev).reshape(-1, 3).contiguous()
        d_off = torch.from_numpy(off).to(dev)
        for _ in range(2): eng.pca_batch(pts, d_off, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.pca_batch(pts, d_off, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{what} pca 64 clouds, {int(off[-1])} points (1e4..4e6 each): {dt:.3f} ms  {off[-1]*12/dt/1e6:.0f} GB/s of input", flush=True)
    elif what == "png":
        # PNG files -> gray views on the device: staging on host threads, inflate + filters on the device, and the
        # host decoder on the same threads for comparison
        import numpy as np, tempfile, shutil, os
        from pathlib import Path
        from PIL import Image
        ingest = importlib.import_module("image-processing_amd.ingest")
        n, workers = int(os.environ.get("KB_PNG_N", "3072")), int(os.environ.get("KB_PNG_WORKERS", "16"))
        root = Path(tempfile.mkdtemp(prefix="fx_png_"))
        try:
            views = gray_views(256, 224, 224).cpu().numpy()
            flat = np.where(views > 128, 255, 30).astype(np.uint8)   # renderer-like: two flat levels
            for kind, arr in (("noisy", views), ("flat", flat)):
                for k in range(256):
                    Image.fromarray(arr[k]).save(root / f"{kind}_{k:03d}.png")
                paths = [str(root / f"{kind}_{k % 256:03d}.png") for k in range(n)]
                size = sum(os.path.getsize(p) for p in paths[:256]) / 256
                ingest.stage_png_batch(paths[:64], 224, 224, workers)
                t0 = time.perf_counter(); staged = ingest.stage_png_batch(paths, 224, 224, workers); t_stage = time.perf_counter() - t0
                eng.png_inflate_batch(staged, 224, 224, dev); torch.cuda.synchronize()
                t0 = time.perf_counter(); got = eng.png_inflate_batch(staged, 224, 224, dev); torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
                eng.profile(True); eng.png_inflate_batch(staged, 224, 224, dev); k_ms = eng.profile_last_ms(); eng.profile(False)
                host = np.empty((n, 224, 224), dtype=np.uint8)
                t0 = time.perf_counter(); ingest.decode_png_batch(paths, 224, 224, host, workers); t_host = time.perf_counter() - t0
                ok = bool((got["status"] == 0).all()) and bool((got["gray"].cpu().numpy() == host).all())
                print(f"png {kind} {n} x 224x224 ({size/1e3:.1f} KB each), {workers} host threads: stage {n/t_stage/1e3:.1f} k files/s, "
                      f"copy + device decode {t_dev*1e3:.2f} ms ({n/t_dev/1e3:.0f} k views/s; kernels {k_ms:.2f} ms), "
                      f"host decode {n/t_host/1e3:.1f} k views/s; equal {ok}", flush=True)
        finally:
            shutil.rmtree(root, ignore_errors=True)
    elif what == "hu1k":
        for (B,H,W,c) in [(7500,1024,1024,False),(7500,1024,1024,True),(30000,512,512,True),(60000,512,256,True)]:
            ms, gbps = time_hu(B,H,W,c)
            print(f"hu B={B} {H}x{W} contour={c}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
    elif what == "husmall":   # small images: block tails against streaming
        for (B,H,W,c) in [(200000,224,224,False),(200000,224,224,True),(400000,128,128,False),(100000,256,256,False),(100000,256,256,True),(60000,512,512,False)]:
            ms, gbps = time_hu(B,H,W,c)
            print(f"hu B={B} {H}x{W} contour={c}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
            torch.cuda.empty_cache()
    elif what == "hutail":   # the contour variant's block tail at the reference's view sizes
        for (B,H,W,c) in [(200000,224,224,True),(100000,256,256,True),(125000,512,512,True)]:
            ms, gbps = time_hu(B,H,W,c)
            print(f"hu B={B} {H}x{W} contour={c}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
            torch.cuda.empty_cache()
    elif what == "hu1":   # one case from the environment, for counter passes: HU1="B,H,W,contour"
        B,H,W,c = [int(x) for x in os.environ.get("HU1", "200000,224,224,1").split(",")]
        ms, gbps = time_hu(B,H,W,bool(c),reps=2)
        print(f"hu B={B} {H}x{W} contour={bool(c)}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
    elif what == "hunc":   # moments only, from views to the tiled path
        for (B,H,W,c) in [(200000,224,224,False),(100000,256,256,False),(125000,512,512,False),(8000,1024,1024,False),(64,8192,8192,False)]:
            ms, gbps = time_hu(B,H,W,c)
            print(f"hu B={B} {H}x{W} contour={c}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
            torch.cuda.empty_cache()
    elif what == "hulds":
        for (B,H,W,c) in [(60000,512,512,True),(60000,512,512,False),(100000,256,256,True)]:
            ms, gbps = time_hu(B,H,W,c)
            print(f"hu B={B} {H}x{W} contour={c}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
    elif what == "hufoot":   # footprint and contour work against the streaming rate
        for (B,H,W,c) in [(125000,512,512,False),(125000,512,512,True),(16000,512,512,False),(16000,512,512,True),(500,8192,8192//16,False),(2000,2048,2048,False),(8000,1024,1024,False),(8000,1024,1024,True)]:
            ms, gbps = time_hu(B,H,W,c)
            print(f"hu B={B} {H}x{W} contour={c}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
            torch.cuda.empty_cache()
    elif what == "hu":
        for (B,H,W,c) in [(30000,512,512,False),(30000,512,512,True),(100000,256,256,False),(100000,256,256,True)]:
            ms, gbps = time_hu(B,H,W,c)
            print(f"hu B={B} {H}x{W} contour={c} dbg={os.environ.get('FX_DBG','0')}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
    elif what == "fps":
        for (B,M,n) in [(4000,2048,1024),(2000,8192,1024),(512,16384,1024),(64,100000,1024),(64,100000,2),(256,100000,1024),(1024,20000,1024),(8,4000000,1024)]:
            cs, fp = time_fps(B,M,n)
            print(f"fps B={B} M={M} n={n}: center_scale {cs:.3f} ms ({B*M*12/cs/1e6:.0f} GB/s)  fps {fp:.3f} ms  {B/fp*1e3:.0f} clouds/s  {B*M*n/fp/1e6:.1f} G point-updates/s", flush=True)
    elif what == "bin":
        import numpy as np
        for rows, cols in ((1000000, 31), (100000, 31)):
            x = torch.randn((rows, cols), dtype=torch.float64, device=dev)
            for method in ("median", "quantile", "otsu"):
                eng.column_thresholds(x, method, 0.3); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5): eng.column_thresholds(x, method, 0.3)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
                print(f"thresholds {rows}x{cols} {method}: {dt:.3f} ms", flush=True)
            thr = eng.column_thresholds(x)["thr"]
            eng.binarize_ge(x, thr); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): eng.binarize_ge(x, thr)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
            print(f"binarize {rows}x{cols}: {dt:.3f} ms  {rows*cols*9/dt/1e6:.0f} GB/s", flush=True)
    elif what == "render":
        rv = importlib.import_module("image-processing_amd.render_views")
        for B, N, V, res in ((2000, 1024, 12, 224), (557, 1024, 12, 224), (200, 8192, 12, 224)):
            pts = eng.synth_points(B, N, seed=42, device=dev).reshape(B * N, 3).contiguous()
            off = torch.arange(B + 1, dtype=torch.int64, device=dev) * N
            rv.render_views_device(pts, off, V, res); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): rv.render_views_device(pts, off, V, res)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
            eng.profile(True); rv.render_views_device(pts, off, V, res); k = eng.profile_last_ms(); eng.profile(False)
            print(f"render B={B} N={N} V={V} {res}px: {dt:.2f} ms per call ({B*V/dt:.0f} k views/s), kernel {k:.2f} ms", flush=True)
    elif what == "pcar":
        import numpy as np
        rng = np.random.default_rng(0)
        for lo, hi, B in ((500, 1024, 300000), (3, 256, 1000000), (1025, 2048, 100000)):
            sizes = rng.integers(lo, hi + 1, B)
            off = np.zeros(B + 1, dtype=np.int64); np.cumsum(sizes, out=off[1:])
            pts = torch.randn((int(off[-1]), 3), dtype=torch.float32, device=dev)
            d_off = torch.from_numpy(off).to(dev)
            for _ in range(2): eng.pca_batch(pts, d_off)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): eng.pca_batch(pts, d_off)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
            print(f"pca ragged B={B} sizes {lo}..{hi}: {dt:.3f} ms per call  {off[-1]*12/dt/1e6:.0f} GB/s", flush=True)
        # 2-D sets (contour points as a separate call): uniform and ragged
        pts2 = torch.randn((300000, 512, 2), dtype=torch.float32, device=dev)
        for _ in range(2): eng.pca_batch(pts2, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.pca_batch(pts2, None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        print(f"pca 2-D B=300000 N=512: {dt:.3f} ms per call  {pts2.numel()*4/dt/1e6:.0f} GB/s", flush=True)
        sizes = rng.integers(100, 2049, 200000)
        off = np.zeros(200001, dtype=np.int64); np.cumsum(sizes, out=off[1:])
        pts2 = torch.randn((int(off[-1]), 2), dtype=torch.float32, device=dev)
        d_off = torch.from_numpy(off).to(dev)
        for _ in range(2): eng.pca_batch(pts2, d_off)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.pca_batch(pts2, d_off)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        print(f"pca 2-D ragged B=200000 sizes 100..2048: {dt:.3f} ms per call  {off[-1]*8/dt/1e6:.0f} GB/s", flush=True)
    elif what == "pcas":   # uniform batches of small sets
        for (B, N) in [(1000000, 256), (1000000, 128), (1000000, 64), (1000000, 32), (2000000, 16)]:
            ms, gbps = time_pca(B, N, reps=8)
            print(f"pca B={B} N={N}: {ms:.3f} ms  {gbps:.0f} GB/s  {ms*1e6/B:.3f} ns/set", flush=True)
    elif what == "pcaq":   # one ragged batch of small sets (for counter runs)
        import numpy as np
        rng = np.random.default_rng(0)
        lo, hi, B = (int(v) for v in os.environ.get("PCAQ", "3,256,1000000").split(","))
        sizes = rng.integers(lo, hi + 1, B)
        off = np.zeros(B + 1, dtype=np.int64); np.cumsum(sizes, out=off[1:])
        pts = torch.randn((int(off[-1]), 3), dtype=torch.float32, device=dev)
        d_off = torch.from_numpy(off).to(dev)
        for _ in range(2): eng.pca_batch(pts, d_off)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): eng.pca_batch(pts, d_off)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
        print(f"pca ragged B={B} sizes {lo}..{hi}: {dt:.3f} ms per call  {off[-1]*12/dt/1e6:.0f} GB/s", flush=True)
    elif what == "sil1":
        ms, gbps = time_sil(20000, 224, 224, 224, reps=3)
        print(f"sil B=20000 224x224->224: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
    elif what == "pca1m":
        for (B, N) in [(1000000, 1024), (1000000, 256), (500000, 512)]:
            ms, gbps = time_pca(B, N, reps=12)
            print(f"pca B={B} N={N}: {ms:.3f} ms  {gbps:.0f} GB/s  ({os.environ.get('FX_LIB', 'libfx.so')})", flush=True)
    elif what == "sil":
        for (B,H,W,res) in [(20000,224,224,224),(100000,224,224,224),(20000,512,512,224)]:
            ms, gbps = time_sil(B,H,W,res)
            print(f"sil B={B} {H}x{W}->{res}: {ms:.3f} ms  {gbps:.0f} GB/s  {B/ms/1e3:.2f} M views/s", flush=True)
    else:
        for (B,N) in [(300000,1024),(100000,2048),(1000000,256)]:
            ms, gbps = time_pca(B,N)
            print(f"pca B={B} N={N}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
