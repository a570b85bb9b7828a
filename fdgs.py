"""`import fdgs` -> the package in ./4dgaussians_amd (whose name is not a Python identifier)."""
import importlib as _il
import sys as _sys

_sys.modules[__name__] = _il.import_module("4dgaussians_amd")
