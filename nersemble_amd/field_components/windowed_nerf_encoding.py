"""WindowedNeRFEncoding -- mirror of the reference's field_components/windowed_nerf_encoding.py:10-92 (on top
of nerfstudio's NeRFEncoding): 2*pi-scaled input, frequencies 2^linspace(min,max,n), sin / cos via a pi/2
phase shift, cosine-eased window per frequency band, optionally appends the 2*pi-SCALED input."""
from typing import Optional

import torch
from torch import nn, Tensor


class WindowedNeRFEncoding(nn.Module):
    def __init__(self, in_dim: int, num_frequencies: int, min_freq_exp: float, max_freq_exp: float,
                 include_input: bool = False) -> None:
        super().__init__()
        self.in_dim = in_dim
        self.num_frequencies = num_frequencies
        self.min_freq = min_freq_exp
        self.max_freq = max_freq_exp
        self.include_input = include_input

    def get_out_dim(self) -> int:
        out = self.in_dim * self.num_frequencies * 2
        return out + self.in_dim if self.include_input else out

    def posenc_window(self, windows_param) -> Tensor:
        bands = torch.linspace(self.min_freq, self.max_freq, self.num_frequencies)
        x = torch.clamp(windows_param - bands, 0, 1)
        return 0.5 * (1 - torch.cos(torch.pi * x))

    def forward(self, in_tensor: Tensor, covs: Optional[Tensor] = None, windows_param: Optional[float] = None) -> Tensor:
        if covs is not None:
            raise NotImplementedError("integrated positional encoding is not used by NeRSemble")
        x = 2 * torch.pi * in_tensor
        freqs = 2 ** torch.linspace(self.min_freq, self.max_freq, self.num_frequencies, device=x.device)
        scaled = (x[..., None] * freqs).reshape(*x.shape[:-1], -1)            # [..., in_dim * n_freq]
        enc = torch.sin(torch.cat([scaled, scaled + torch.pi / 2.0], dim=-1))
        if windows_param is not None:
            window = self.posenc_window(windows_param).to(x.device)[None, :].repeat(x.shape[-1], 1).reshape(-1).repeat(2)
            enc = window * enc
        if self.include_input:
            enc = torch.cat([enc, x], dim=-1)
        return enc
