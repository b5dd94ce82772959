#!/usr/bin/env python3
"""Numerics study for DESIGN.md section 10, item 3 (CPU, torch): can the tolerance-meeting precision run as ONE fp16 product
plus fp8 CORRECTION products instead of split's three fp16 products?

    operand  = hi (fp16) + lo8 / s_lo        lo8 = e4m3(s_lo (v - hi)),   s_lo a per-tensor power of two
    product  = w_hi x_hi  (fp16 MFMA)  +  w_lo8 x8 / s  +  w8 x_lo8 / s   (fp8 MFMA, fp32 accumulate; x8 = e4m3(s_x x), w8 likewise)

against the fp32 forward of oracle/unet_spec.py on one tile; the modes "fp16" (storage of the fast path) and "split" (hi + lo fp16,
lo*lo dropped) are evaluated the same way for comparison.  Prints max-abs / rms of the 5 output channels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import unet_spec as S  # noqa: E402

F8 = torch.float8_e4m3fn


def pow2_scale(t, target=256.0):
    m = t.abs().max().item()
    if m == 0:
        return 1.0
    import math
    return 2.0 ** math.floor(math.log2(target / m))


def q8(t):
    s = pow2_scale(t)
    return (t * s).to(F8).float() / s


def qbits(t, mant):
    """round to `mant` explicit mantissa bits, unlimited exponent range (what a block-scaled fp6 / fp4 could hold at best)"""
    m, e = torch.frexp(t)
    return torch.ldexp(torch.round(m * 2 ** (mant + 1)) / 2 ** (mant + 1), e)


def qmx(t, fmt):
    """OCP MX block format along the channel axis (dim 1 of activations, dim 1 = cin of weights): blocks of 32 share a
    power-of-two scale, elements are fp6 e2m3 / fp4 e2m1 -- what v_mfma_scale_f32_16x16x128_f8f6f4 multiplies at 2x the
    fp8 rate.  The block scale puts the block's largest magnitude into the top binade of the element format."""
    grid = {"fp6": [m / 8 for m in range(8)] + [(1 + m / 8) * 2 ** e for e in range(3) for m in range(8)],
            "fp4": [0, 0.5, 1, 1.5, 2, 3, 4, 6]}[fmt]
    g = torch.tensor(sorted(set(grid)), dtype=torch.float32)
    top = 4.0   # both formats' largest binade starts at 4
    shp = t.shape
    C = shp[1]
    pad = (-C) % 32
    tt = F.pad(t, (0, 0) * (t.dim() - 2) + (0, pad)) if pad else t
    tt = tt.reshape(shp[0], (C + pad) // 32, 32, -1)
    mx = tt.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
    sc = torch.exp2(torch.floor(torch.log2(mx)) - 2)      # max / sc in [4, 8)
    u = (tt / sc).clamp(-g[-1], g[-1])
    idx = torch.bucketize(u.abs(), (g[1:] + g[:-1]) / 2)
    q = torch.sign(u) * g[idx] * sc
    q = q.reshape(shp[0], C + pad, *shp[2:])[:, :C]
    return q


def parts(v, mode):
    """-> list of (tensor, kind) decompositions used below"""
    hi = v.half().float()
    if mode == "fp16":
        return hi, None, None
    res = v - hi
    if mode == "split":
        return hi, res.half().float(), hi          # lo16, "full" operand for the cross terms = hi
    if mode == "hi16+lo8":
        return hi, q8(res), q8(v)                  # lo8, and the fp8 copy of the operand for the other cross term
    if mode in ("hi16+mxfp6", "hi16+mxfp4"):
        return hi, qmx(res, mode[7:]), qmx(v, mode[7:])
    if mode.startswith("hi16+m"):                  # hi16+mK: corrections with K mantissa bits (fp6 e2m3: 3, fp4 e2m1: 1)
        k = int(mode[6:])
        return hi, qbits(res, k), qbits(v, k)
    raise ValueError(mode)


def conv(x, w, bias, mode, **kw):
    xh, xl, xf = parts(x, mode)
    wh, wl, wf = parts(w, mode)
    y = F.conv3d(xh, wh, bias, **kw)
    if mode != "fp16":
        y = y + F.conv3d(xf, wl, None, **kw) + F.conv3d(xl, wf, None, **kw)
    return y


def forward(model, x, mode):
    def store(v):   # what a tensor written to HBM holds
        if mode == "fp16":
            return v.half().float()
        hi = v.half().float()
        if mode == "split":
            lo = (v - hi).half().float()
        elif mode == "hi16+lo8":
            lo = q8(v - hi)
        elif mode.startswith("hi16+mx"):
            lo = qmx(v - hi, mode[7:])
        else:
            lo = qbits(v - hi, int(mode[6:]))
        return hi + lo

    def block(m, t):
        y = conv(t, m.conv.weight, m.conv.bias, mode, stride=m.conv.stride, padding=m.conv.padding)
        B, C = y.shape[:2]
        g = y.double().reshape(B, S.GN_GROUPS, -1)
        mu = g.mean(-1, keepdim=True)
        var = (g * g).mean(-1, keepdim=True) - mu * mu
        rstd = (1.0 / (var.clamp_min(0) + S.GN_EPS).sqrt()).float()
        gam = m.norm.weight.reshape(1, S.GN_GROUPS, -1)
        a = (gam * rstd).reshape(1, C, 1, 1, 1)
        b = (m.norm.bias.reshape(1, S.GN_GROUPS, -1) - mu.float() * gam * rstd).reshape(1, C, 1, 1, 1)
        return store(F.silu(a * store(y) + b))

    t = x
    for i, m in enumerate(model.enc0):
        if i == 0:   # the stem keeps exact weights (hi + lo split in every mode)
            y = F.conv3d(t.half().float(), m.conv.weight, m.conv.bias, padding=1)
            B, C = y.shape[:2]
            t = store(F.silu(F.group_norm(store(y), S.GN_GROUPS, m.norm.weight, m.norm.bias, S.GN_EPS)))
        else:
            t = block(m, t)
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
    return torch.cat([torch.tanh(y[:, 0:3]), torch.sigmoid(y[:, 3:5])], dim=1).half().float()


def main():
    torch.set_num_threads(8)
    model = S.build()
    shape = tuple(int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else (96, 96, 20)))
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 1) + shape, generator=g).to(torch.float16)
    x = ((img - 127.5) / 73.9).float()
    with torch.no_grad():
        want = model(x)
        for mode in ("fp16", "split", "hi16+lo8", "hi16+mxfp6", "hi16+mxfp4", "hi16+m2", "hi16+m1"):
            got = forward(model, x, mode)
            e = (got - want).abs()
            print(f"{mode:10s} tile {shape}: max-abs {e.max().item():.3e}  rms {e.pow(2).mean().sqrt().item():.3e}", flush=True)


if __name__ == "__main__":
    main()
