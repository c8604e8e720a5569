"""Runs the PCA kernel a few times on 300k x 1024 points (for profiler passes)."""
import importlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
eng = importlib.import_module("image-processing_amd.engine")
pts = eng.synth_points(300000, 1024, seed=42, device=torch.device("cuda", 0))
for _ in range(3):
    eng.pca_batch(pts)
torch.cuda.synchronize()
