"""Drop-in for the reference's NeRFs/DFANeRF/decoder.py: same symbols, MI355X-native implementation."""
import _bootstrap  # noqa: F401
from dfanerf.decoder import Decoder, DeformationField_ori  # noqa: F401
