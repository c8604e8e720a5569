import importlib, sys, tempfile, shutil, time, cProfile, pstats
from pathlib import Path
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
eng = importlib.import_module("image-processing_amd.engine")
pca = importlib.import_module("image-processing_amd.extract_3d_pca_features")
dev = torch.device("cuda", 0)
C = 20000
root = Path(tempfile.mkdtemp(prefix="fx_prof_"))
pts = eng.synth_points(C, 1024, seed=7, device=dev).cpu().numpy()
sampled = root / "sampled"; sampled.mkdir()
def save(i): np.save(sampled / f"tree_{i:06d}.npy", pts[i])
with ThreadPoolExecutor(16) as p: list(p.map(save, range(C)))
args = ["--sampled_dir", str(sampled), "--out_dir", str(root / "desc"), "--reports_dir", str(root / "rep"), "--overwrite", "--workers", "16"]
pca.main(args)
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); pca.main(args); t = time.perf_counter() - t0; pr.disable()
print("main seconds", t)
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
shutil.rmtree(root, ignore_errors=True)
