"""Drop-in for the reference's NeRFs/DFANeRF/load_audface.py."""
import _bootstrap  # noqa: F401
from dfanerf.load_audface import load_audface_data_split  # noqa: F401
