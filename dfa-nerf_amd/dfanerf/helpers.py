"""Host-side mirror of the reference's run_nerf_helpers.py (the symbols the driver imports with `*`).

Reference: /root/reference/NeRFs/DFANeRF/run_nerf_helpers.py.  get_rays / ndc_rays / sample_pdf dispatch to
the HIP kernels when their inputs live on the GPU; the dead AD-NeRF heritage of that file (NeRF, FaceNeRF,
AudioNet, DCT helpers) is not carried over.  The reference turns on
torch.autograd.set_detect_anomaly(True) at import (:5); this module deliberately does not."""
import numpy as np
import torch

from .nets import (AudioAttNet, AudioNet_W2L, Embedder, ExpressionEnc, get_embedder)  # noqa: F401


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device if isinstance(x, torch.Tensor) else None))


def to8b(x):
    """(255 * clip(x, 0, 1)) truncated to uint8.  numpy in -> numpy out (as the reference); a device tensor
    goes through the HIP kernel and stays on the device."""
    if isinstance(x, torch.Tensor) and x.is_cuda:
        from . import engine
        return engine.to8b(x)
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def _device_of(c2w):
    if isinstance(c2w, torch.Tensor) and c2w.is_cuda:
        return c2w.device
    return torch.device("cuda") if torch.cuda.is_available() else None


def get_rays(H, W, focal, c2w, cx=None, cy=None, stride=1):
    """rays_o, rays_d [H//stride, W//stride, 3]; ray (y, x) through pixel centre (x*?, y) like the reference
    (linspace(0, W-1, W//stride)).  Runs dfn_get_rays[_strided] on the GPU; stride != 1 is never used by the driver (golden G1c)."""
    dev = _device_of(c2w)
    if dev is None:
        raise RuntimeError("get_rays runs on the GPU (no CPU fallback)")
    from . import engine
    return engine.get_rays(int(H), int(W), float(focal), c2w, cx, cy, device=dev, stride=int(stride))


def get_rays_np(H, W, focal, c2w, cx=None, cy=None):
    """numpy variant kept for API parity (unused by the driver, run_nerf_helpers.py:468-481)."""
    cx = W * .5 if cx is None else cx
    cy = H * .5 if cy is None else cy
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - cx) / focal, -(j - cy) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    return np.broadcast_to(c2w[:3, -1], np.shape(rays_d)), rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    from . import engine
    if not rays_o.is_cuda:
        raise RuntimeError("ndc_rays runs on the GPU (no CPU fallback)")
    return engine.ndc_rays(H, W, focal, near, rays_o, rays_d)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """Inverse-CDF sampling on the GPU.  pytest=True feeds the reference's np.random.seed(0) numbers."""
    from . import engine
    if not bins.is_cuda:
        raise RuntimeError("sample_pdf runs on the GPU (no CPU fallback)")
    u = None
    if pytest:
        np.random.seed(0)
        shape = list(bins.shape[:-1]) + [N_samples]
        u = np.broadcast_to(np.linspace(0., 1., N_samples), shape) if det else np.random.rand(*shape)
        u = torch.as_tensor(np.ascontiguousarray(u), dtype=torch.float32, device=bins.device)
    lead = bins.shape[:-1]
    out = engine.sample_pdf(bins.reshape(-1, bins.shape[-1]), weights.reshape(-1, weights.shape[-1]), N_samples,
                            det=det, u=None if u is None else u.reshape(-1, N_samples))
    return out.reshape(*lead, N_samples)
