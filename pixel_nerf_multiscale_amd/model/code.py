import math

import torch
from torch.autograd.profiler import record_function

from ..util import as_conf


class PositionalEncoding(torch.nn.Module):
    """NeRF positional encoding with the reference's layout and buffers (reference model/code.py:6-56):
    out = [x, sin(f0 x), sin(f0 x + pi/2), sin(f1 x), ...], f_k = freq_factor * 2^k.  Inside the render
    path the encoding is computed in the HIP kernels; this module carries the hyper-parameters, the
    state-dict buffers (_freqs, _phases) and a torch forward for standalone use."""

    def __init__(self, num_freqs=6, d_in=3, freq_factor=math.pi, include_input=True):
        super().__init__()
        self.num_freqs, self.d_in, self.freq_factor, self.include_input = num_freqs, d_in, float(freq_factor), include_input
        self.freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)
        self.d_out = num_freqs * 2 * d_in + (d_in if include_input else 0)
        self.register_buffer("_freqs", self.freqs.repeat_interleave(2).view(1, -1, 1))
        ph = torch.zeros(2 * num_freqs)
        ph[1::2] = math.pi * 0.5
        self.register_buffer("_phases", ph.view(1, -1, 1))

    def forward(self, x):
        """PositionalEncoding.forward under the reference's profiler label (code.py:40)."""
        with record_function("positional_enc"):
            return self._forward_impl(x)

    def _forward_impl(self, x):
        if x.numel() == 0:
            return x.new_empty(x.shape[0], self.d_out)
        e = torch.sin(self._phases + x[:, None, :] * self._freqs).flatten(1)
        return torch.cat((x, e), dim=-1) if self.include_input else e

    @classmethod
    def from_conf(cls, conf, d_in=3):
        conf = as_conf(conf)
        return cls(conf.get_int("num_freqs", 6), d_in, conf.get_float("freq_factor", math.pi),
                   conf.get_bool("include_input", True))
