import torch
from torch import nn

from ..util import as_conf


class ResnetBlockFC(nn.Module):
    """Pre-activation residual FC block: x + fc_1(relu(fc_0(relu(x)))) (reference resnetfc.py:10-62)."""

    def __init__(self, size_in, size_out=None, size_h=None, beta=0.0):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        if beta > 0 or size_in != size_out:
            raise NotImplementedError("softplus / shortcut blocks are not used by any shipped config")
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        nn.init.zeros_(self.fc_0.bias)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.zeros_(self.fc_1.bias)
        nn.init.zeros_(self.fc_1.weight)
        self.shortcut = None

    def forward(self, x):
        return x + self.fc_1(torch.relu(self.fc_0(torch.relu(x))))


class ResnetFC(nn.Module):
    """Parameter container with the reference's state-dict layout (lin_in, lin_z.{b}, blocks.{b}.fc_{0,1},
    lin_out; reference resnetfc.py:65-250).  The render path reads the parameters straight into the HIP
    kernels (see _native.pnr_mlp); the module itself is never evaluated in PyTorch."""

    def __init__(self, d_in, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, beta=0.0,
                 combine_layer=1000, combine_type="average", use_spade=False):
        super().__init__()
        if isinstance(d_latent, (list, tuple)):
            d_latent = sum(int(v) for v in d_latent)
        if use_spade or beta > 0:
            raise NotImplementedError("use_spade / softplus are not supported (no shipped config uses them)")
        self.d_in, self.d_out, self.n_blocks = int(d_in), int(d_out), int(n_blocks)
        self.d_latent, self.d_hidden = int(d_latent), int(d_hidden)
        self.combine_layer, self.combine_type, self.use_spade = int(combine_layer), combine_type, False
        if self.d_in > 0:
            self.lin_in = nn.Linear(self.d_in, self.d_hidden)
            nn.init.zeros_(self.lin_in.bias)
            nn.init.kaiming_normal_(self.lin_in.weight, a=0, mode="fan_in")
        self.lin_out = nn.Linear(self.d_hidden, self.d_out)
        nn.init.zeros_(self.lin_out.bias)
        nn.init.kaiming_normal_(self.lin_out.weight, a=0, mode="fan_in")
        self.blocks = nn.ModuleList([ResnetBlockFC(self.d_hidden) for _ in range(self.n_blocks)])
        if self.d_latent != 0:
            n_lin_z = min(self.combine_layer, self.n_blocks)
            self.lin_z = nn.ModuleList([nn.Linear(self.d_latent, self.d_hidden) for _ in range(n_lin_z)])
            for m in self.lin_z:
                nn.init.zeros_(m.bias)
                nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in")
        self.activation = nn.ReLU()

    def forward(self, zx, combine_inner_dims=(1,), combine_index=None, dim_size=None):
        raise RuntimeError(
            "ResnetFC is evaluated inside the HIP point kernel (PixelNeRFNet.forward / NeRFRenderer -> "
            "libpnr_hip pnr_point_mlp / pnr_render); there is no PyTorch evaluation path in this package")

    @classmethod
    def from_conf(cls, conf, d_in, **kwargs):
        conf = as_conf(conf)
        return cls(d_in, n_blocks=conf.get_int("n_blocks", 5), d_hidden=conf.get_int("d_hidden", 128),
                   beta=conf.get_float("beta", 0.0), combine_layer=conf.get_int("combine_layer", 1000),
                   combine_type=conf.get_string("combine_type", "average"),
                   use_spade=conf.get_bool("use_spade", False), **kwargs)
