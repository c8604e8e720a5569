import importlib, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
import kbench
eng = kbench.eng
imgs = kbench.gray_views(1, 224, 224)
for backend in ("opencv", "pil"):
    print("backend", backend, flush=True)
    for _ in range(2):
        eng.silhouette_batch(imgs, 224, 0.05, True, backend=backend)
        torch.cuda.synchronize()
