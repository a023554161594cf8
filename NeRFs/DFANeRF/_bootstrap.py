"""Puts the MI355X-native package (dfa-nerf_amd/dfanerf) on sys.path for the drop-in modules of this directory."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "dfa-nerf_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
