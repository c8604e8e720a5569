"""Large-batch stress of the silhouette stage: many views per persistent block, two runs must agree bit for bit
and a sample must equal the CPU oracle."""
import importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import numpy as np, torch
import kbench
from oracle import cpu_ref
eng = kbench.eng
dev = torch.device("cuda", 0)
rng = np.random.default_rng(11)
for (B, H, W, res, kind) in ((20000, 224, 224, 224, "views"), (12000, 200, 152, 160, "noise"), (6000, 224, 224, 224, "rings"), (3000, 512, 512, 224, "views")):
    if kind == "views":
        imgs = kbench.gray_views(B, H, W, seed=7)
    elif kind == "noise":
        imgs = (torch.rand((B, H, W), device=dev) * 256).to(torch.uint8)
        imgs[::2] = ((torch.rand((B // 2 + B % 2, H, W), device=dev) < 0.45) * 200).to(torch.uint8)[: imgs[::2].shape[0]]
    else:
        base = torch.full((B, H, W), 235, dtype=torch.uint8, device=dev)
        for k in range(B):
            pass
        # rings and blobs of random sizes, drawn on the host for a few hundred patterns and tiled
        pats = np.full((300, H, W), 235, dtype=np.uint8)
        for p in pats:
            for _ in range(rng.integers(1, 4)):
                cy, cx = rng.integers(30, H - 30), rng.integers(30, W - 30)
                r1 = rng.integers(8, 60); r0 = rng.integers(0, r1)
                yy, xx = np.ogrid[:H, :W]
                d2 = (yy - cy) ** 2 + (xx - cx) ** 2
                p[(d2 <= r1 * r1) & (d2 >= r0 * r0)] = 30
        imgs = torch.from_numpy(pats).to(dev).repeat((B + 299) // 300, 1, 1)[:B].contiguous()
        imgs = (imgs.to(torch.int16) + torch.randint(-6, 7, imgs.shape, device=dev, dtype=torch.int16)).clamp_(0, 255).to(torch.uint8)
    for backend in ("opencv", "pil"):
        a = eng.silhouette_batch(imgs, res, 0.05, True, backend=backend)["mask"]
        b = eng.silhouette_batch(imgs, res, 0.05, True, backend=backend)["mask"]
        torch.cuda.synchronize()
        same = bool(torch.equal(a, b))
        idx = rng.choice(B, 24, replace=False)   # most of them from a block's later rounds
        bad = 0
        oracle = cpu_ref.silhouette_cv if backend == "opencv" else cpu_ref.silhouette
        for i in idx:
            want = oracle(imgs[int(i)].cpu().numpy(), res, 0.05, True)[0]
            bad += int(not np.array_equal(a[int(i)].cpu().numpy(), want))
        print(f"{kind} {backend} B={B} {H}x{W}->{res}: runs agree {same}, oracle mismatches {bad}/24", flush=True)
