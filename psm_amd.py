"""Import alias: ``import psm_amd`` == the package in
``solving-poisson-s-equation-through-dl-for-cfd-apllications_amd/`` (whose
directory name is not a valid Python identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("solving-poisson-s-equation-through-dl-for-cfd-apllications_amd")
sys.modules[__name__] = _pkg
