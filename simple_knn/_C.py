"""Drop-in for the pybind module `simple_knn._C` of the un-vendored submodule submodules/simple-knn (.gitmodules:1-3):
the one function the reference calls, `distCUDA2(points) -> Tensor[N]` (scene/gaussian_model.py:148)."""
import importlib

_knn = importlib.import_module("4dgaussians_amd.knn")
distCUDA2 = _knn.distCUDA2
