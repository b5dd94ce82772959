"""Torch fp32 CPU reference of the build's U-Net (``unet_spec``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

PARITY UNPINNED for the network body: the reference builds its network with
``cfg_to_bism_model`` (skoots/lib/utils.py:17-107) from the third-party package
``bism`` (requirements.txt:1, ``bism>=0.0.6``, no upper pin), which is absent
from this image and from the reference tree.  What the reference fixes is the
I/O contract, restated here:

  (B, 1, X, Y, Z) float  ->  (B, 5, X, Y, Z)
  channels [0:3] = vectors in [-1, 1] (tanh), [-2] = skeleton probability
  (sigmoid), [-1] = semantic probability (sigmoid)
  (skoots/lib/eval.py:145-147, skoots/train/engine.py:461-463).

The body is the build's own dense-conv U-Net, shaped by the reference's model
config (skoots/config.py:20-34): DIMS=[32,64,128,64,32], DEPTHS=[2,2,2,2,2],
IN_CHANNELS=1, OUT_CHANNELS=32, ACTIVATION "silu" (registered at
skoots/lib/utils.py:37-42).  Every block is Conv3d -> GroupNorm(8) -> SiLU:

  enc0  : depths[0] x conv3(-> 32)            full resolution
  down0 : conv2 stride 2 (32 -> 64)
  enc1  : depths[1] x conv3(64 -> 64)         1/2
  down1 : conv2 stride 2 (64 -> 128)
  mid   : depths[2] x conv3(128 -> 128)       1/4
  red1  : conv1 (128 -> 64); nearest x2; cat[skip1, up] -> depths[3] x conv3(-> 64)
  red0  : conv1 (64 -> 32);  nearest x2; cat[skip0, up] -> depths[4] x conv3(-> 32)
  heads : conv1 (32 -> 5): tanh on [0:3], sigmoid on [3], [4]

The HIP runner (skoots_amd/unet.py) executes exactly this graph; parity
tolerance for its outputs is 1e-3 absolute (BASELINE.json north_star).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

GN_GROUPS = 8
GN_EPS = 1e-5


class ConvGNAct(nn.Module):
    def __init__(self, cin: int, cout: int, k: int, stride: int = 1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, k, stride=stride, padding=(k - 1) // 2 if stride == 1 else 0)
        self.norm = nn.GroupNorm(GN_GROUPS, cout, eps=GN_EPS)

    def forward(self, x):
        return F.silu(self.norm(self.conv(x)))


class UNetSpec(nn.Module):
    def __init__(self, in_channels: int = 1, dims: Sequence[int] = (32, 64, 128, 64, 32),
                 depths: Sequence[int] = (2, 2, 2, 2, 2)):
        super().__init__()
        d0, d1, d2, d3, d4 = dims
        assert d3 == d1 and d4 == d0, "decoder widths mirror the encoder"
        self.dims, self.depths = tuple(dims), tuple(depths)

        def stack(cin, cout, n):
            return nn.ModuleList([ConvGNAct(cin if i == 0 else cout, cout, 3) for i in range(n)])

        self.enc0 = stack(in_channels, d0, depths[0])
        self.down0 = ConvGNAct(d0, d1, 2, stride=2)
        self.enc1 = stack(d1, d1, depths[1])
        self.down1 = ConvGNAct(d1, d2, 2, stride=2)
        self.mid = stack(d2, d2, depths[2])
        self.red1 = ConvGNAct(d2, d3, 1)
        self.dec1 = stack(d1 + d3, d3, depths[3])
        self.red0 = ConvGNAct(d3, d4, 1)
        self.dec0 = stack(d0 + d4, d4, depths[4])
        self.heads = nn.Conv3d(d4, 5, 1)

    def forward(self, x):
        """Extents that are not multiples of 4 (a volume thinner than the tile) are zero-padded at the
        high end up to the next multiple of 4 and the output is cropped back; how the reference's own
        network (bism, absent) treats such crops is not in the tree, so this is the build's definition."""
        shape = x.shape[2:]
        pad = [(-s) % 4 for s in shape]
        if any(pad):
            x = F.pad(x, (0, pad[2], 0, pad[1], 0, pad[0]))
            return self._forward(x)[:, :, :shape[0], :shape[1], :shape[2]]
        return self._forward(x)

    def _forward(self, x):
        for m in self.enc0:
            x = m(x)
        s0 = x
        x = self.down0(x)
        for m in self.enc1:
            x = m(x)
        s1 = x
        x = self.down1(x)
        for m in self.mid:
            x = m(x)
        x = F.interpolate(self.red1(x), size=s1.shape[2:], mode="nearest")
        x = torch.cat([s1, x], dim=1)
        for m in self.dec1:
            x = m(x)
        x = F.interpolate(self.red0(x), size=s0.shape[2:], mode="nearest")
        x = torch.cat([s0, x], dim=1)
        for m in self.dec0:
            x = m(x)
        y = self.heads(x)
        return torch.cat([torch.tanh(y[:, 0:3]), torch.sigmoid(y[:, 3:5])], dim=1)


def _q16(t):
    return t.half().float()


def forward_fp16_storage(model: "UNetSpec", x):
    """The same graph with the HIP path's STORAGE precision restated in fp32 torch ops:
    MFMA layers use fp16 weights and fp16 activations with fp32 accumulation; a conv's
    raw output is stored as fp16, GroupNorm statistics come from the fp32 accumulators,
    the affine + SiLU result is stored as fp16; the stem and the heads keep fp32
    weights; the 5 output channels are stored as fp16.  Used to separate kernel
    correctness (HIP vs this, tight tolerance) from the cost of fp16 storage itself
    (this vs the fp32 forward)."""

    def block(m: ConvGNAct, t, fp32_weights=False):
        w = m.conv.weight if fp32_weights else _q16(m.conv.weight)
        y = F.conv3d(t, w, m.conv.bias, stride=m.conv.stride, padding=m.conv.padding)
        B, C = y.shape[:2]
        g = y.double().reshape(B, GN_GROUPS, -1)
        mu = g.mean(-1, keepdim=True)
        var = (g * g).mean(-1, keepdim=True) - mu * mu
        rstd = (1.0 / (var.clamp_min(0) + GN_EPS).sqrt()).float()
        gam = m.norm.weight.reshape(1, GN_GROUPS, -1)
        a = (gam * rstd).reshape(1, C, 1, 1, 1)
        b = (m.norm.bias.reshape(1, GN_GROUPS, -1) - mu.float() * gam * rstd).reshape(1, C, 1, 1, 1)
        return _q16(F.silu(a * _q16(y) + b))

    shape = x.shape[2:]
    pad = [(-s) % 4 for s in shape]
    if any(pad):
        x = F.pad(x, (0, pad[2], 0, pad[1], 0, pad[0]))
        return forward_fp16_storage(model, x)[:, :, :shape[0], :shape[1], :shape[2]]
    t = x
    for i, m in enumerate(model.enc0):
        t = block(m, t, fp32_weights=(i == 0))
    s0 = t
    t = block(model.down0, t)
    for m in model.enc1:
        t = block(m, t)
    s1 = t
    t = block(model.down1, t)
    for m in model.mid:
        t = block(m, t)
    t = F.interpolate(block(model.red1, t), size=s1.shape[2:], mode="nearest")
    t = torch.cat([s1, t], dim=1)
    for m in model.dec1:
        t = block(m, t)
    t = F.interpolate(block(model.red0, t), size=s0.shape[2:], mode="nearest")
    t = torch.cat([s0, t], dim=1)
    for m in model.dec0:
        t = block(m, t)
    y = model.heads(t)
    return _q16(torch.cat([torch.tanh(y[:, 0:3]), torch.sigmoid(y[:, 3:5])], dim=1))


def flops_per_voxel(dims=(32, 64, 128, 64, 32), depths=(2, 2, 2, 2, 2), in_channels=1) -> float:
    """Algorithmic conv FLOPs (2*Cin*Cout*k^3 per output voxel) per full-resolution voxel."""
    d0, d1, d2, d3, d4 = dims
    f = 0.0

    def stack(cin, cout, n, s):
        return sum(2.0 * (cin if i == 0 else cout) * cout * 27 / s for i in range(n))

    f += stack(in_channels, d0, depths[0], 1)
    f += 2.0 * d0 * d1 * 8 / 8
    f += stack(d1, d1, depths[1], 8)
    f += 2.0 * d1 * d2 * 8 / 64
    f += stack(d2, d2, depths[2], 64)
    f += 2.0 * d2 * d3 / 64
    f += stack(d1 + d3, d3, depths[3], 8)
    f += 2.0 * d3 * d4 / 8
    f += stack(d0 + d4, d4, depths[4], 1)
    f += 2.0 * d4 * 5
    return f


def build(seed: int = 101196, **kw) -> UNetSpec:
    """Random-init network under the seed the reference trains with (train/engine.py:53)."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    m = UNetSpec(**kw).eval()
    # GroupNorm affine away from (1, 0) so parity tests exercise gamma/beta.
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.GroupNorm):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.3, 0.3)
    torch.random.set_rng_state(g)
    return m
