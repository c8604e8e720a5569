import importlib, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
import kbench
eng = kbench.eng
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
imgs = kbench.gray_views(B, 224, 224)
for backend in ("opencv",):
    print("backend", backend, "B", B, flush=True)
    for _ in range(2):
        eng.silhouette_batch(imgs, 224, 0.05, True, backend=backend)
        torch.cuda.synchronize()
