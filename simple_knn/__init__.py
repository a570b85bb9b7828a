"""Import shim: `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:22 of the reference) resolves to the
HIP implementation in 4dgaussians_amd (csrc/knn.hip) when this repository is on PYTHONPATH."""
