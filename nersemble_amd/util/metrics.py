"""Image metrics of the evaluation path -- what the reference's model builds in ``populate_modules``
(``nersemble_instant_ngp.py:158-160``: ``PeakSignalNoiseRatio(data_range=1.0)``, ``structural_similarity_index_measure``,
``LearnedPerceptualImagePatchSimilarity(normalize=True)``) and calls in ``get_image_metrics_and_images`` (``:446-449,
:488-491``) on ``[1, C, H, W]`` images.

torchmetrics / lpips are third-party packages that are not installed here and not vendored with the reference, so these
are restatements of their published algorithms (torchmetrics >= 1.0 functional SSIM; LPIPS v0.1 with the AlexNet
trunk): PARITY UNPINNED against those packages.  ``tests/test_eval_metrics.py`` checks PSNR / SSIM against an
independent float64 numpy / scipy computation and known answers.

One-off evaluation-time transforms on whole images: plain torch ops on whatever device the image lives on, no kernel of
their own.
"""
import math
import warnings
from typing import Optional, Sequence, Union

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class PeakSignalNoiseRatio(nn.Module):
    """``10 log10(data_range^2 / mean((preds - target)^2))`` over all elements (torchmetrics' default reduction)."""

    def __init__(self, data_range: float = 1.0):
        super().__init__()
        self.data_range = float(data_range)

    def forward(self, preds: Tensor, target: Tensor) -> Tensor:
        mse = ((preds - target) ** 2).mean()
        return 10.0 * torch.log10(self.data_range ** 2 / mse)


def _gaussian_window(sigma: float, dtype, device) -> Tensor:
    """1-D normalised Gaussian with torchmetrics' support: ``2 int(3.5 sigma + 0.5) + 1`` taps (11 for sigma 1.5)."""
    size = int(3.5 * sigma + 0.5) * 2 + 1
    dist = torch.arange((1 - size) / 2, (1 + size) / 2, 1, dtype=dtype, device=device)
    g = torch.exp(-((dist / sigma) ** 2) / 2)
    return g / g.sum()


def structural_similarity_index_measure(preds: Tensor, target: Tensor, sigma: Union[float, Sequence[float]] = 1.5,
                                        data_range: Optional[float] = None, k1: float = 0.01, k2: float = 0.03
                                        ) -> Tensor:
    """SSIM with a Gaussian window (torchmetrics defaults: sigma 1.5 -> 11 x 11 window, k1 0.01, k2 0.03,
    ``data_range=None`` -> the larger of the two images' value ranges, 'elementwise_mean' reduction).

    ``preds`` / ``target``: ``[B, C, H, W]``.  Local moments are Gaussian-weighted means of p, t, p^2, t^2 and pt; the
    index map ``(2 mu_p mu_t + c1)(2 cov + c2) / ((mu_p^2 + mu_t^2 + c1)(var_p + var_t + c2))`` is averaged over the
    pixels whose window lies fully inside the image (torchmetrics reflect-pads, convolves, then crops the padded
    border again -- which is the same set), then over the batch.
    """
    if preds.shape != target.shape or preds.dim() != 4:
        raise ValueError("expected two [B, C, H, W] images of the same shape")
    sig = (float(sigma), float(sigma)) if not isinstance(sigma, Sequence) else tuple(float(s) for s in sigma)
    if data_range is None:
        data_range = torch.maximum(preds.max() - preds.min(), target.max() - target.min())
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    gy = _gaussian_window(sig[0], preds.dtype, preds.device)
    gx = _gaussian_window(sig[1], preds.dtype, preds.device)
    if preds.shape[-2] < gy.numel() or preds.shape[-1] < gx.numel():
        raise ValueError("image smaller than the SSIM window")
    b, c = preds.shape[:2]
    stack = torch.cat([preds, target, preds * preds, target * target, preds * target])        # [5B, C, H, W]
    kernel = (gy[:, None] * gx[None, :]).expand(c, 1, -1, -1)
    moments = F.conv2d(stack, kernel, groups=c)                                                # 'valid'
    mu_p, mu_t, e_pp, e_tt, e_pt = moments.split(b)
    var_p = (e_pp - mu_p * mu_p).clamp(min=0.0)
    var_t = (e_tt - mu_t * mu_t).clamp(min=0.0)
    cov = e_pt - mu_p * mu_t
    index = ((2 * mu_p * mu_t + c1) * (2 * cov + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (var_p + var_t + c2))
    return index.reshape(b, -1).mean(-1).mean()


class LearnedPerceptualImagePatchSimilarity(nn.Module):
    """LPIPS v0.1, AlexNet trunk (the torchmetrics default ``net_type='alex'``), ``normalize=True`` = inputs in [0,1].

    distance = sum over the 5 ReLU stages of  mean_xy( w_l . (f_l(p)/|f_l(p)| - f_l(t)/|f_l(t)|)^2 )
    with f_l the AlexNet activations of the (shifted / scaled) images and w_l the learned non-negative 1x1 weights.

    The trunk and the linear heads are LEARNED weights (ImageNet AlexNet + the LPIPS calibration) that ship with the
    ``lpips`` / ``torchvision`` packages; neither is available offline.  ``weights`` = path of a state dict with
    torchvision's ``features.{0,3,6,8,10}.{weight,bias}`` and lpips' ``lin{0..4}.model.1.weight``; without it the
    metric answers NaN (once with a warning) instead of a made-up number.
    """
    _CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1),
              (10, 256, 256, 3, 1, 1))

    def __init__(self, normalize: bool = True, weights: Optional[str] = None):
        super().__init__()
        self.normalize = normalize
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1))
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1))
        self.convs, self.lins = nn.ModuleList(), nn.ModuleList()      # built when weights arrive
        self.loaded = False
        self._warned = False
        if weights is not None:
            self.load_weights(torch.load(weights, map_location="cpu"))

    def load_weights(self, state) -> None:
        self.convs = nn.ModuleList([nn.Conv2d(i, o, k, s, p) for _, i, o, k, s, p in self._CONVS])
        self.lins = nn.ModuleList([nn.Conv2d(o, 1, 1, bias=False) for _, _, o, _, _, _ in self._CONVS])
        for p in self.parameters():
            p.requires_grad_(False)
        with torch.no_grad():
            for conv, lin, (idx, *_), k in zip(self.convs, self.lins, self._CONVS, range(5)):
                conv.weight.copy_(state[f"features.{idx}.weight"])
                conv.bias.copy_(state[f"features.{idx}.bias"])
                lin.weight.copy_(state[f"lin{k}.model.1.weight"])
        self.loaded = True

    def _stages(self, x: Tensor):
        x = (x - self.shift) / self.scale
        for k, conv in enumerate(self.convs):
            if k in (1, 2):
                x = F.max_pool2d(x, 3, 2)
            x = F.relu(conv(x))
            yield x

    def forward(self, img1: Tensor, img2: Tensor) -> Tensor:
        if not self.loaded:
            if not self._warned:
                warnings.warn("LPIPS weights are not available offline: reporting NaN (see util/metrics.py)")
                self._warned = True
            return torch.full((), math.nan, device=img1.device)
        if self.shift.device != img1.device:
            self.to(img1.device)
        if self.normalize:
            img1, img2 = 2 * img1 - 1, 2 * img2 - 1
        total = 0.0
        for fa, fb, lin in zip(self._stages(img1), self._stages(img2), self.lins):
            na = fa / (fa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            nb = fb / (fb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            total = total + lin((na - nb) ** 2).mean(dim=(2, 3))
        return total.mean()
