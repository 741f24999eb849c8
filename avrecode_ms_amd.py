"""Alias so that ``import avrecode_ms_amd`` finds the package directory ``avrecode-ms_amd/``
(named after the reference repository; a hyphen is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("avrecode-ms_amd")
