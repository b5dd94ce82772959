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


def _e4m3(t):
    return t.to(torch.float8_e4m3fn)


@pytest.mark.parametrize("shape", [(1, (8, 12, 20), 32, 32, 32), (2, (12, 14, 20), 32, 32, 32), (1, (14, 22, 10), 64, 64, 64),
                                   (1, (8, 8, 12), 32, 64, 32)])
def test_upfold_mix8_fp8_phases_exact(shape):
    """sk_conv3d_upfold_mix8, the two block-scaled fp8 phases alone (hi halves = 0): every (cout, cin) pair has ONE non-zero
    tap w = s (1 + m / 8) + j 2^-14, so a folded weight is that tap or zero and all four fp8 images (x8, lo8 of both sources,
    2^(b+11) w_lo, 2^b w) are exact small integers: the result 2^-(b+15) (conv(x8, 2^(b+11) w_lo) + conv(lo8, 2^b w)) over
    cat([skip, upsample(up)]) is exact in fp32.  Pins the tap-row pairing of the skip chunks on the de-interleaved planes, the
    (ty; tz pair) K blocks of the upsampled chunks per parity class, both lane maps and the zero halos of the 8-bit bytes."""
    from skoots_amd import unet as U
    B, osp, c_skip, c_up, cout = shape
    gen = torch.Generator().manual_seed(5 + osp[1])
    lo = tuple(v // 2 for v in osp)
    cin = c_skip + c_up
    base = (1 + torch.randint(0, 8, (cout, cin), generator=gen).float() / 8) * (torch.randint(0, 2, (cout, cin), generator=gen) * 2 - 1).float()
    j = torch.randint(-3, 4, (cout, cin), generator=gen).float()
    tap = torch.randint(0, 27, (cout, cin), generator=gen)
    onehot = torch.nn.functional.one_hot(tap, 27).float().reshape(cout, cin, 3, 3, 3)
    w = onehot * (base + j * 2.0 ** -14)[..., None, None, None]
    wp, b = U.pack_conv_weight_upfold_mix8(w, c_skip, DEV)
    assert b == 6
    parts = {}
    for name, c, sp in (("skip", c_skip, osp), ("up", c_up, lo)):
        x8 = torch.randint(-3, 4, (B,) + sp + (c,), generator=gen).float()
        l8 = torch.randint(-3, 4, (B,) + sp + (c,), generator=gen).float()
        parts[name] = (x8, l8, U.mix8_line(torch.zeros((B,) + sp + (c,), dtype=torch.float16), _e4m3(x8), _e4m3(l8)).to(DEV))
    got, _ = U.conv3d_upfold_mix8(parts["skip"][2], parts["up"][2], wp, b, torch.zeros(cout, device=DEV), cout)
    torch.cuda.synchronize()

    def cat(k):
        return torch.cat([_cf(parts["skip"][k]).double(), F.interpolate(_cf(parts["up"][k]).double(), scale_factor=2, mode="nearest")], dim=1)
    wlo = (onehot * (j * 2.0 ** -14 * 2.0 ** (b + 11))[..., None, None, None]).double()
    wv = (onehot * (base * 2.0 ** b)[..., None, None, None]).double()
    want = (F.conv3d(cat(0), wlo, padding=1) + F.conv3d(cat(1), wv, padding=1)) * 2.0 ** -(b + 15)
    assert torch.equal(_cf(U.join_pair(got.cpu()).double()), want)


@pytest.mark.parametrize("shape", [(1, (8, 12, 20), 32, 32, 32), (2, (12, 14, 20), 32, 32, 32), (1, (14, 22, 10), 64, 64, 64),
                                   (1, (8, 8, 12), 32, 64, 32)])
def test_upfold_mix8_vs_float64(shape):
    """sk_conv3d_upfold_mix8 on realistic operands against a float64 conv of the same activations (what a split pair holds) and
    fp32 weights, next to sk_conv3d_upfold_split: the fp8 corrections leave <= 1/10 of the error of the uncorrected fp16 product
    (sk_conv3d_mix8's bound, tests/test_hip_unet.py); integer weights and activations (no lo parts) come out exactly."""
    from skoots_amd import unet as U
    B, osp, c_skip, c_up, cout = shape
    gen = torch.Generator().manual_seed(31 + osp[1])
    lo = tuple(v // 2 for v in osp)
    skip = U.join_pair(U.split_pair(torch.nn.functional.silu(torch.randn((B,) + osp + (c_skip,), generator=gen) * 1.5)))
    up = U.join_pair(U.split_pair(torch.nn.functional.silu(torch.randn((B,) + lo + (c_up,), generator=gen) * 1.5)))
    w = torch.randn((cout, c_skip + c_up, 3, 3, 3), generator=gen) / ((c_skip + c_up) * 27) ** 0.5
    bias = torch.randn(cout, generator=gen) * 0.1
    x = torch.cat([_cf(skip).double(), F.interpolate(_cf(up).double(), scale_factor=2, mode="nearest")], dim=1)
    want = F.conv3d(x, w.double(), bias.double(), padding=1)
    scale = max(1.0, want.abs().max().item())
    wp, b = U.pack_conv_weight_upfold_mix8(w, c_skip, DEV)
    got, partial = U.conv3d_upfold_mix8(U.mix8_of(skip).to(DEV), U.mix8_of(up).to(DEV), wp, b, bias.to(DEV), cout)
    sp, sp_partial = U.conv3d_upfold(U.split_pair(skip).to(DEV), U.split_pair(up).to(DEV),
                                     U.pack_conv_weight_upfold(w, c_skip, DEV, split=True), bias.to(DEV), cout, split=True)
    torch.cuda.synchronize()
    e_mix = (_cf(U.join_pair(got.cpu())).double() - want).abs().max().item()
    e_split = (_cf(U.join_pair(sp.cpu())).double() - want).abs().max().item()
    e_f16 = (F.conv3d(x.half().float(), w.half().float(), bias, padding=1).double() - want).abs().max().item()
    print(f"upfold mix8 {shape}: max-abs split {e_split:.2e} mix8 {e_mix:.2e} fp16 operands {e_f16:.2e} (scale {scale:.2f})")
    assert e_mix <= 3e-5 * scale and e_mix <= e_f16 / 10
    assert torch.allclose(partial.sum(1), sp_partial.sum(1), rtol=1e-4, atol=3e-2)
    # integers: only the fp16 phases contribute, exactly
    xi = torch.randint(-3, 4, (B,) + osp + (c_skip,), generator=gen).float()
    ui = torch.randint(-3, 4, (B,) + lo + (c_up,), generator=gen).float()
    wi = torch.randint(-1, 2, (cout, c_skip + c_up, 3, 3, 3), generator=gen).float()
    wpi, bi = U.pack_conv_weight_upfold_mix8(wi, c_skip, DEV)
    goti, _ = U.conv3d_upfold_mix8(U.mix8_of(xi).to(DEV), U.mix8_of(ui).to(DEV), wpi, bi, torch.zeros(cout, device=DEV), cout)
    wanti = F.conv3d(torch.cat([_cf(xi), F.interpolate(_cf(ui), scale_factor=2, mode="nearest")], dim=1), wi, padding=1)
    assert torch.equal(_cf(U.join_pair(goti.cpu())), wanti)
