"""Drop-in for the reference's NeRFs/DFANeRF/run_nerf_helpers.py (the driver does `from run_nerf_helpers import *`)."""
import _bootstrap  # noqa: F401
from dfanerf.helpers import *  # noqa: F401,F403
from dfanerf.helpers import (AudioAttNet, AudioNet_W2L, Embedder, ExpressionEnc, get_embedder, get_rays,  # noqa: F401
                             get_rays_np, img2mse, mse2psnr, ndc_rays, sample_pdf, to8b)
