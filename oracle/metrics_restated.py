"""ORACLE (test infrastructure): literal restatement of the image-similarity metrics xvr takes from
``diffdrr.metrics`` -- the unfold-based patch NCC family -- and of xvr's own ``XrayTransforms``.

PARITY UNPINNED for the diffdrr part (absent dependency, /root/reference/uv.lock:955-977): restated from
the published definitions and anchored on the reference's call sites
  MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5])   src/xvr/registrar/base.py:119-121, src/xvr/model/loss.py:16
  GradientNormalizedCrossCorrelation2d(11, sigma)                  src/xvr/registrar/base.py:122
  DoubleGeodesicSE3(sdd)                                           src/xvr/model/loss.py:18, src/xvr/metrics/evaluator.py:15
``XrayTransforms`` IS in the reference tree and is restated line by line (src/xvr/utils/preprocess.py:5-66).
"""
import torch


def to_patches(x, patch_size):
    """[b,c,h,w] -> [b, c*h'*w', p, p]: every p x p patch becomes a channel."""
    x = x.unfold(2, patch_size, 1).unfold(3, patch_size, 1).contiguous()  # [b,c,h',w',p,p]
    b, c, hh, ww, p1, p2 = x.shape
    return x.reshape(b, c * hh * ww, p1, p2)


def ncc(x1, x2, patch_size=None, eps=1e-5):
    """Mean over channels (= patches) of the z-scored product; variance is biased, eps inside the sqrt."""
    if patch_size is not None:
        x1, x2 = to_patches(x1, patch_size), to_patches(x2, patch_size)
    _, c, h, w = x1.shape

    def norm(x):
        mu = x.mean(dim=[-1, -2], keepdim=True)
        var = x.var(dim=[-1, -2], keepdim=True, correction=0) + eps
        return (x - mu) / var.sqrt()

    return torch.einsum("b...,b...->b", norm(x1), norm(x2)) / (c * h * w)


def multiscale_ncc(x1, x2, patch_sizes=(None, 9), patch_weights=(0.5, 0.5), eps=1e-5):
    return sum(w * ncc(x1, x2, p, eps) for p, w in zip(patch_sizes, patch_weights))


def sobel(img):
    gx = torch.tensor([[1.0, 0.0, -1.0], [2.0, 0.0, -2.0], [1.0, 0.0, -1.0]])
    gy = torch.tensor([[1.0, 2.0, 1.0], [0.0, 0.0, 0.0], [-1.0, -2.0, -1.0]])
    k = torch.stack([gx, gy]).unsqueeze(1).to(img)
    return torch.nn.functional.conv2d(img, k, padding=1)


def gaussian_blur(img, kernel_size, sigma):
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, kernel_size, dtype=img.dtype, device=img.device)
    k1 = torch.exp(-0.5 * (x / sigma) ** 2)
    k1 = k1 / k1.sum()
    pad = kernel_size // 2
    img = torch.nn.functional.pad(img, (pad, pad, pad, pad), mode="reflect")
    c = img.shape[1]
    img = torch.nn.functional.conv2d(img, k1.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
    return torch.nn.functional.conv2d(img, k1.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)


def gradient_ncc(x1, x2, patch_size=None, sigma=0.0, eps=1e-5):
    if sigma and sigma > 0:
        x1, x2 = gaussian_blur(x1, 5, sigma), gaussian_blur(x2, 5, sigma)
    return ncc(sobel(x1), sobel(x2), patch_size, eps)


def equalize(x, n_bins=256, tau=0.01, eps=1e-10):
    """Differentiable (soft-histogram) histogram equalisation of [B, 1, H, W] images in [0, 1], src/xvr/utils/preprocess.py:34-66:
    Gaussian-kernel histogram over n_bins, its normalised CDF, every pixel mapped to the CDF averaged with its own bin weights.
    (one image at a time: the [pixels, bins] weight matrix is H * W * n_bins floats)"""
    B, _, H, W = x.shape
    bins = torch.linspace(0, 1, n_bins, device=x.device, dtype=x.dtype)[None, None]
    out = []
    for b in range(B):
        diff = x[b].reshape(1, -1, 1) - bins
        weights = (-diff.square() / (2 * tau**2)).exp()
        histogram = weights.sum(dim=1)
        histogram = histogram / (histogram.sum(dim=1, keepdim=True) + eps)
        cdf = torch.cumsum(histogram, dim=1)
        cdf_min = cdf[:, 0:1]
        cdf_normalized = (cdf - cdf_min) / (1 - cdf_min + eps)
        weights_norm = weights / (weights.sum(dim=-1, keepdim=True) + eps)
        out.append((weights_norm * cdf_normalized[:, None]).sum(dim=-1).view(1, 1, H, W))
    return torch.cat(out)


def xray_transforms(x, height, width=None, mean=0.15, std=0.1, equalize_=False, per_image=False):
    """Standardize (global min-max over the whole tensor) -> [Equalize] -> Resize((h, w)) -> Normalize(mean, std)
    (src/xvr/utils/preprocess.py:5-31; torchvision's Resize on tensors is antialiased bilinear).  per_image: every image by its
    own min / max (this package's batched multi-start; the same thing for the single image the reference ever passes)."""
    width = height if width is None else width
    if per_image:
        lo, hi = x.amin(dim=(1, 2, 3), keepdim=True), x.amax(dim=(1, 2, 3), keepdim=True)
        x = (x - lo) / (hi - lo + 1e-6)
    else:
        x = (x - x.min()) / (x.max() - x.min() + 1e-6)
    if equalize_:
        x = equalize(x)
    if x.shape[-2:] != (height, width):
        x = torch.nn.functional.interpolate(x, size=(height, width), mode="bilinear", antialias=True, align_corners=False)
    return (x - mean) / std
