"""GPU parity tests of the folded-upsample decoder conv (csrc/conv3d_up.hip, sk_conv3d_upfold): the same function as
sk_conv3d over [skip, upsampled x] -- the first conv of each decoder level of the network (oracle/unet_spec.py; the
reference builds it at skoots/lib/utils.py:17-107).  Integer-valued operands make every product, sum and folded weight
exact, so there the two kernels and torch must agree BIT FOR BIT; on random data the folded weights are rounded to fp16
after the sum instead of before it, which moves the result within the fp16 rounding of the weights (tolerance below)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _cf(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


# (B, out spatial, c_skip, c_up[, cout = 32])
SHAPES = [
    (1, (8, 12, 20), 32, 32),      # Zl = 10: 3 rows per workgroup, exact
    (2, (12, 14, 20), 32, 32),     # Yl = 7: ragged last workgroup, batch 2
    (1, (16, 22, 10), 32, 32),     # Zl = 5: 6 rows per workgroup (the half-resolution level of a 300x300x20 tile)
    (1, (4, 8, 6), 32, 32),        # Zl = 3: 10 rows, 30 of 32 columns
    (1, (20, 10, 24), 32, 32),     # Zl = 12: 2 rows, 24 of 32 columns; two x-chunks
    (1, (8, 6, 40), 32, 32),       # Zl = 20: one row per workgroup, 20 of 32 columns
    (1, (8, 8, 20), 64, 32),       # two skip chunks
    (1, (8, 8, 12), 32, 64),       # two upsampled chunks
    (1, (44, 30, 20), 32, 32),     # several steps per x-chunk, ring reuse
    (1, (14, 22, 10), 64, 64, 64),  # the level-1 decoder conv: cout 64 = two launches; x extent not a multiple of 4
    (2, (6, 12, 20), 32, 32),      # x extent 6: a ragged second step
    (2, (8, 300, 20), 32, 32),     # the production (y, z) plane: 50 workgroups per x-chunk, weight rows in LDS
    (1, (8, 150, 10), 64, 64, 64),  # ... and the level-1 plane
]


def _make(B, osp, c_skip, c_up, integer, seed, cout=32):
    gen = torch.Generator().manual_seed(seed)
    lo = tuple(s // 2 for s in osp)
    if integer:
        skip = torch.randint(-3, 4, (B, c_skip) + osp, generator=gen).half()
        up = torch.randint(-3, 4, (B, c_up) + lo, generator=gen).half()
        w = torch.randint(-2, 3, (cout, c_skip + c_up, 3, 3, 3), generator=gen).float()
        b = torch.randint(-4, 5, (cout,), generator=gen).float()
    else:
        skip = torch.randn((B, c_skip) + osp, generator=gen).half()
        up = torch.randn((B, c_up) + lo, generator=gen).half()
        w = torch.randn((cout, c_skip + c_up, 3, 3, 3), generator=gen) / ((c_skip + c_up) * 27) ** 0.5
        b = torch.randn(cout, generator=gen) * 0.1
    return skip, up, w, b


def _torch(skip, up, w, b):
    x = torch.cat([skip.float(), F.interpolate(up.float(), scale_factor=2, mode="nearest")], dim=1)
    return F.conv3d(x, w, b, padding=1)


def test_upfold_geometry_support():
    from skoots_amd import _ffi
    assert _ffi.lib.sk_conv3d_upfold_num_blocks(300, 300, 20, 32) > 0      # production tile, level 0
    assert _ffi.lib.sk_conv3d_upfold_num_blocks(150, 150, 10, 64) > 0       # ... level 1 (two launches of 32 channels)
    assert _ffi.lib.sk_conv3d_upfold_num_blocks(300, 300, 20, 128) < 0      # cout 128: not built
    assert _ffi.lib.sk_conv3d_upfold_num_blocks(512, 512, 128, 32) < 0      # Zl = 64 > 32
    assert _ffi.lib.sk_conv3d_upfold_num_blocks(8, 6, 64, 32) < 0           # Zl = 32: the four sub-planes pass 192 positions
    assert _ffi.lib.sk_conv3d_upfold_num_blocks(9, 12, 20, 32) < 0          # odd x extent
    assert _ffi.lib.sk_conv3d_upfold_num_blocks(8, 12, 19, 32) < 0


@pytest.mark.parametrize("shape", SHAPES)
def test_upfold_exact_on_integers(shape):
    """Integer operands: folded kernel == direct kernel == torch fp32, bit for bit, and equal GroupNorm sums."""
    from skoots_amd import unet as U
    B, osp, c_skip, c_up = shape[:4]
    cout = shape[4] if len(shape) > 4 else 32
    skip, up, w, b = _make(B, osp, c_skip, c_up, True, 7 + osp[1], cout)
    want = _torch(skip, up, w, b)
    assert want.abs().max() < 2048      # representable in fp16 exactly
    s_d, u_d = _cl(skip).to(DEV), _cl(up).to(DEV)
    got, partial = U.conv3d_upfold(s_d, u_d, U.pack_conv_weight_upfold(w, c_skip, DEV), b.to(DEV), cout)
    got = _cf(got.cpu().float())
    assert torch.equal(got, want)
    zeros = torch.zeros(4096, dtype=torch.uint8, device=DEV)
    ref, rpartial = U.conv3d([(s_d, 0), (u_d, 1)], U.pack_conv_weight(w, DEV), b.to(DEV), cout, 3, osp, zeros)
    assert torch.equal(_cf(ref.cpu().float()), want)
    ps, rs = partial.sum(dim=1).cpu(), rpartial.sum(dim=1).cpu()
    assert torch.equal(ps[..., 0], rs[..., 0])                   # integer sums below 2^24: order-free
    assert torch.allclose(ps[..., 1], rs[..., 1], rtol=1e-6)     # sums of squares pass 2^24: fp32 rounding by order


@pytest.mark.parametrize("shape", SHAPES[:5] + SHAPES[-2:])
def test_upfold_vs_torch_random(shape):
    """Random operands against torch fp32 on the fp16-rounded inputs and the UNROUNDED weights: the kernel's weights are
    fp16 (a folded one is the fp16 of a sum of up to 8 taps), accumulation fp32, output fp16."""
    from skoots_amd import unet as U
    B, osp, c_skip, c_up = shape[:4]
    cout = shape[4] if len(shape) > 4 else 32
    skip, up, w, b = _make(B, osp, c_skip, c_up, False, 11 + osp[0], cout)
    want = _torch(skip, up, w, b)
    got, partial = U.conv3d_upfold(_cl(skip).to(DEV), _cl(up).to(DEV), U.pack_conv_weight_upfold(w, c_skip, DEV), b.to(DEV), cout)
    got = _cf(got.cpu().float())
    err = (got - want).abs().max().item()
    assert err <= 3e-3 * max(1.0, want.abs().max().item()), err
    p = partial.sum(dim=1).cpu()
    wq = got.reshape(B, cout // 4, 4, -1)      # statistics are those of the stored fp16 values
    assert torch.allclose(p[..., 0], wq.sum(dim=(2, 3)), rtol=1e-3, atol=2e-2 * wq.shape[-1] ** 0.5)
    assert torch.allclose(p[..., 1], (wq ** 2).sum(dim=(2, 3)), rtol=2e-3)


def test_network_with_and_without_fold():
    """The whole network with the folded decoder conv against the same network on the direct kernel: the difference
    stays inside the fp16 mode's documented distance from the fp32 oracle."""
    from skoots_amd import unet as U
    model = U.smoke_model(DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    vol = torch.randint(0, 256, (64, 64, 24), generator=g, device=DEV, dtype=torch.uint8).to(torch.float16)
    origins = [(0, 0, 0), (4, 0, 4)]
    a = model.forward_tiles(vol, origins, (60, 64, 20), 127.5, 73.9).clone()
    model.fold_upsample = False
    b = model.forward_tiles(vol, origins, (60, 64, 20), 127.5, 73.9).clone()
    model.fold_upsample = True
    assert (a.float() - b.float()).abs().max().item() <= 5e-3
    assert (a.float() - b.float()).pow(2).mean().sqrt().item() <= 5e-4


def test_network_with_one_pass_stem():
    """HipUNet.stem_single_pass (round 4 A/B switch): the stem conv runs once, stores the raw fp16 result with its statistics
    (sk_conv3d_stem_raw) and enc0.1 activates it in LDS.  Against the default two-pass stem (statistics, then the conv again
    with the activation in its epilogue) the raw tensor is rounded to fp16 once more: the outputs differ by fp16 rounding
    noise, inside the mode's distance from the fp32 oracle."""
    from skoots_amd import unet as U
    model = U.smoke_model(DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    vol = torch.randint(0, 256, (64, 64, 24), generator=g, device=DEV, dtype=torch.uint8).to(torch.float16)
    origins = [(0, 0, 0), (4, 0, 4)]
    a = model.forward_tiles(vol, origins, (60, 64, 20), 127.5, 73.9).clone()
    model.stem_single_pass = True
    b = model.forward_tiles(vol, origins, (60, 64, 20), 127.5, 73.9).clone()
    model.stem_single_pass = False
    assert not torch.equal(a, b)   # the switch took another path
    assert (a.float() - b.float()).abs().max().item() <= 5e-3
    assert (a.float() - b.float()).pow(2).mean().sqrt().item() <= 5e-4


@pytest.mark.parametrize("shape", [(1, (8, 12, 20), 32, 32, 32), (2, (12, 14, 20), 32, 32, 32), (1, (14, 22, 10), 64, 64, 64),
                                   (1, (8, 8, 12), 32, 64, 32)])
def test_upfold_split_vs_float64(shape):
    """precision "split" (hi + lo fp16 pairs, three MFMA phases per logical chunk): against a float64 conv of the SAME
    pair-representable operands -- the bound test_conv_split_vs_torch holds sk_conv3d_split to (2e-5 of the range)."""
    from skoots_amd import unet as U
    B, osp, c_skip, c_up, cout = shape
    gen = torch.Generator().manual_seed(31 + osp[1])
    lo = tuple(v // 2 for v in osp)
    skip = U.join_pair(U.split_pair(_cl(torch.randn((B, c_skip) + osp, generator=gen))))      # what the pair holds exactly
    up = U.join_pair(U.split_pair(_cl(torch.randn((B, c_up) + lo, generator=gen))))
    w = torch.randn((cout, c_skip + c_up, 3, 3, 3), generator=gen) / ((c_skip + c_up) * 27) ** 0.5
    w = U.join_pair(U.split_pair(w.unsqueeze(-1))).squeeze(-1)
    b = torch.randn(cout, generator=gen) * 0.1
    x = torch.cat([_cf(skip).double(), F.interpolate(_cf(up).double(), scale_factor=2, mode="nearest")], dim=1)
    want = F.conv3d(x, w.double(), b.double(), padding=1)
    got, partial = U.conv3d_upfold(U.split_pair(skip).to(DEV), U.split_pair(up).to(DEV),
                                   U.pack_conv_weight_upfold(w, c_skip, DEV, split=True), b.to(DEV), cout, split=True)
    got = _cf(U.join_pair(got.cpu())).double()
    err = (got - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    p = partial.sum(dim=1).cpu().double()
    wq = want.reshape(B, cout // 4, 4, -1)
    assert torch.allclose(p[..., 0], wq.sum(dim=(2, 3)), rtol=1e-4, atol=1e-3 * wq.shape[-1] ** 0.5)
    assert torch.allclose(p[..., 1], (wq ** 2).sum(dim=(2, 3)), rtol=1e-4)
