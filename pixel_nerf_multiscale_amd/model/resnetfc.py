import torch
from torch.autograd.profiler import record_function
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
    kernels (see _native.pnr_mlp); the module itself is never evaluated in PyTorch — forward() is a native stage
    call on assembled rows (pnr_resnetfc_forward)."""

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

    def native_struct(self):
        """pnr_mlp over this module's parameter storage (fp32 weights only; the packed MFMA stream is owned by
        PixelNeRFNet.mlp_struct).  Returns (struct, keepalive)."""
        from .. import _native as N
        m = N.pnr_mlp()
        m.d_in, m.d_latent, m.d_hidden, m.d_out = self.d_in, self.d_latent, self.d_hidden, self.d_out
        m.n_blocks, m.combine_layer, m.combine_type = self.n_blocks, self.combine_layer, N.COMBINE[self.combine_type]
        keep = []

        def P(t):
            t = N.f32c(t.detach())
            keep.append(t)
            return N.ptr(t)

        m.lin_in_w, m.lin_in_b = P(self.lin_in.weight), P(self.lin_in.bias)
        m.lin_out_w, m.lin_out_b = P(self.lin_out.weight), P(self.lin_out.bias)
        for b, blk in enumerate(self.blocks):
            m.fc0_w[b], m.fc0_b[b] = P(blk.fc_0.weight), P(blk.fc_0.bias)
            m.fc1_w[b], m.fc1_b[b] = P(blk.fc_1.weight), P(blk.fc_1.bias)
        if self.d_latent:
            for b, lz in enumerate(self.lin_z):
                m.lin_z_w[b], m.lin_z_b[b] = P(lz.weight), P(lz.bias)
        return m, keep

    def forward(self, zx, combine_inner_dims=(1,), combine_index=None, dim_size=None):
        """ResnetFC.forward under the reference's profiler label (resnetfc.py:180)."""
        with record_function("resnetfc_infer"):
            return self._forward_impl(zx, combine_inner_dims, combine_index, dim_size)

    def _forward_impl(self, zx, combine_inner_dims=(1,), combine_index=None, dim_size=None):
        """zx (..., d_latent + d_in), latent first -> (..., d_out) with the `combine_inner_dims` reduction of the reference
        (resnetfc.py:173-236, util.combine_interleaved): a native stage call (pnr_resnetfc_forward, fp32 kernels) — the
        module is not evaluated in PyTorch.  Inference only; the differentiable network is PixelNeRFNet.forward in
        training mode (render/autograd.py), which is also where the fused MFMA kernel lives."""
        import ctypes as C
        from .. import _native as N
        if combine_index is not None:
            raise NotImplementedError("combine_index / dim_size (scatter reduction) are not used by the render path")
        if not zx.is_cuda:
            raise RuntimeError("ResnetFC runs on the HIP device only (no CPU / PyTorch evaluation path in this package)")
        if torch.is_grad_enabled() and (zx.requires_grad or any(p.requires_grad for p in self.parameters())) and self.training:
            raise RuntimeError("ResnetFC.forward is an inference stage call; train through PixelNeRFNet.forward / NeRFRenderer")
        if zx.shape[-1] != self.d_latent + self.d_in:
            raise ValueError(f"zx has {zx.shape[-1]} features, expected d_latent + d_in = {self.d_latent + self.d_in}")
        dims = tuple(int(d) for d in combine_inner_dims)
        ns = dims[0]
        inner_pts = 1
        for d in dims[1:]:
            inner_pts *= d
        z2 = N.f32c(zx.detach()).reshape(-1, zx.shape[-1])
        rows = z2.shape[0]
        if self.combine_layer >= self.n_blocks:
            ns, inner_pts = 1, 1                      # the reduction sits behind the last block: never reached (resnetfc.py:214)
        if rows % (ns * inner_pts) != 0:
            raise ValueError(f"{rows} rows do not divide into combine_inner_dims {dims}")
        outer = rows // (ns * inner_pts)
        if ns == 1:                                    # no reduction: one flat batch of rows
            outer, inner_pts = 1, rows
        m, keep = self.native_struct()
        dev = z2.device
        out = torch.empty(outer * inner_pts, self.d_out, device=dev)
        ws = torch.empty(N.lib.pnr_resnetfc_workspace_bytes(C.byref(m), ns), dtype=torch.uint8, device=dev)
        N.check(N.lib.pnr_resnetfc_forward(C.byref(m), N.ptr(z2), outer, ns, inner_pts, N.ptr(out), ws.data_ptr(), ws.numel(),
                                           N.current_stream(dev)), "pnr_resnetfc_forward")
        if ns == 1:
            return out.reshape(*zx.shape[:-1], self.d_out)
        return out.reshape(outer, *dims[1:], self.d_out) if len(dims) > 1 else out

    @classmethod
    def from_conf(cls, conf, d_in, **kwargs):
        conf = as_conf(conf)
        return cls(d_in, n_blocks=conf.get_int("n_blocks", 5), d_hidden=conf.get_int("d_hidden", 128),
                   beta=conf.get_float("beta", 0.0), combine_layer=conf.get_int("combine_layer", 1000),
                   combine_type=conf.get_string("combine_type", "average"),
                   use_spade=conf.get_bool("use_spade", False), **kwargs)
