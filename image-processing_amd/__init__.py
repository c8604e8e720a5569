"""MI355X-native Hu-moment / PCA feature extraction.

The directory name contains a hyphen, so import it with
``importlib.import_module("image-processing_amd")``.  Sub-modules:

  extract_2d_hu, extract_3d_pca_features, merge_features
      the reference's function names over the HIP kernels
  engine      device-level batch API on torch tensors
  synth       host twin of the device shape generator
  dist        sharding and gathering across ranks
  _lib        ctypes binding of libfx.so
"""
__all__ = ["engine", "extract_2d_hu", "extract_3d_pca_features", "merge_features", "synth", "dist", "_lib"]
