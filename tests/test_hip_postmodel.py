"""GPU parity tests for stages a5-a11 (gate/dilate/scatter, CCL, follow, assign,
renumber): HIP path (through the C ABI, via the skoots_amd.lib mirror) against the
golden fixtures made from the reference's functions and against the CPU oracle.
Everything here is integer / byte / index work (or fp32 arithmetic feeding integer
indices): the bar is bit-exact."""
import numpy as np
import pytest
import torch

from oracle import pipeline as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sk():
    import skoots_amd.lib.eval as E
    import skoots_amd.lib.flood_fill as F
    import skoots_amd.lib.morphology as M
    import skoots_amd.lib.skeleton as S
    import skoots_amd.lib.vector_to_embedding as V
    from skoots_amd import _ffi

    class NS:
        pass
    ns = NS()
    ns.E, ns.F, ns.M, ns.S, ns.V, ns.ffi = E, F, M, S, V, _ffi
    return ns


# ----------------------------------------------------------------------------- follow
def test_follow_golden(sk, golden):
    g = golden("follow.npz")
    for i in range(int(g["n"])):
        v = torch.from_numpy(g[f"vector_{i}"]).to(DEV)
        emb = sk.V.vector_to_embedding(torch.tensor(g[f"scale_{i}"]), v, N=int(g[f"n_{i}"]),
                                       decay=float(g[f"decay_{i}"]))
        assert np.array_equal(emb.cpu().numpy().view(np.uint32), g[f"embed_{i}"].view(np.uint32)), i


def test_follow_kat(sk, golden):
    g = golden("kat.npz")
    emb = sk.V.vector_to_embedding(torch.tensor((1, 1, 1)), torch.from_numpy(g["vector"]).to(DEV), N=2)
    assert emb[0, :, 5, 5, 5].tolist() == [6.0, 6.0, 6.0]
    assert np.array_equal(emb.cpu().numpy(), g["embed"])


@pytest.mark.parametrize("shape,scale,n,decay,dtype", [
    ((64, 48, 20), (60, 60, 12), 10, 1.0, torch.float16),
    ((100, 90, 33), (60, 60, 12), 10, 1.0, torch.float16),
    ((100, 90, 33), (20, 20, 4), 7, 0.9, torch.float32),
    ((1, 7, 300), (5, 5, 50), 10, 1.0, torch.float16),     # ragged: degenerate axis
    ((500, 500, 50), (60, 60, 12), 10, 1.0, torch.float16),  # the reference's full crop size
])
def test_follow_vs_oracle(sk, shape, scale, n, decay, dtype):
    gen = torch.Generator().manual_seed(sum(shape))
    v = (torch.rand((1, 3) + shape, generator=gen) * 2 - 1)
    v = (v * (torch.rand((1, 1) + shape, generator=gen) > 0.3)).to(dtype)
    want = O.vector_to_embedding(torch.tensor(scale), v, N=n, decay=decay)
    got = sk.V.vector_to_embedding(torch.tensor(scale), v.to(DEV), N=n, decay=decay).cpu()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_gather_golden(sk, golden):
    g = golden("gather.npz")
    for i in range(int(g["n"])):
        emb = torch.from_numpy(g[f"embed_{i}"]).clone()
        emb += torch.tensor(g[f"origin_{i}"]).view(1, 3, 1, 1, 1)
        got = sk.S.index_skeleton_by_embed(torch.from_numpy(g[f"labels_{i}"]).to(DEV), emb.to(DEV))
        assert got.dtype == torch.int32
        assert np.array_equal(got.cpu().numpy(), g[f"out_{i}"]), i


# ----------------------------------------------------------------------------- dilation
def test_gate_dilate_scatter_golden(sk, golden):
    g = golden("dilate.npz")
    for i in range(int(g["n"])):
        out = torch.from_numpy(g[f"out_{i}"])[0]  # (5, w, h, d)
        _, w, h, d = out.shape
        for ov in ((4, 4, 2), (1, 3, 1), (3, 1, 2)):
            st = sk.E.VolumeState((w + 3, h + 2, d + 1), DEV, keep_planar_vectors=True)
            org = (2, 1, 1)
            st.scatter_tile(out.to(DEV).contiguous(), org, ov)
            sl = tuple(slice(o, s - o) for o, s in zip(ov, (w, h, d)))
            dst = tuple(slice(org[k] + ov[k], org[k] + (w, h, d)[k] - ov[k]) for k in range(3))
            skel = st.skeleton.cpu().numpy()
            assert np.array_equal(skel[dst], g[f"skel_{i}"][0, 0][sl]), (i, ov)
            vp = st.vec_planar.cpu().numpy()
            want = g[f"vec_{i}"][0][(slice(None),) + sl]
            assert np.array_equal(vp[(slice(None),) + dst].view(np.uint16), want.view(np.uint16)), (i, ov)
            v4 = st.vectors_planar().cpu().numpy()
            assert np.array_equal(v4.view(np.uint16), vp.view(np.uint16))
            # nothing outside the interior is touched
            mask = np.ones(skel.shape, bool)
            mask[dst] = False
            assert not skel[mask].any() and not vp[:, mask].any()


def test_max_filter_library_functions(sk):
    gen = torch.Generator().manual_seed(9)
    x = torch.rand((1, 1, 21, 17, 9), generator=gen)
    assert torch.equal(sk.M.binary_dilation(x.to(DEV)).cpu(), O.binary_dilation(x))
    assert torch.equal(sk.M.binary_dilation_2d(x.to(DEV)).cpu(), O.binary_dilation_2d(x))


# ----------------------------------------------------------------------------- flood fill
def test_flood_golden(sk, golden):
    g = golden("flood.npz")
    for i in range(int(g["n"])):
        inp = torch.from_numpy(g[f"in_{i}"].astype(np.int16)).to(DEV)
        got = sk.F.efficient_flood_fill(inp)
        assert got.dtype == torch.int32
        assert np.array_equal(got.cpu().numpy(), g[f"out_{i}"].astype(np.int32)), f"case {i}"


@pytest.mark.parametrize("shape,fill", [((64, 64, 64), 0.25), ((33, 1010, 7), 0.3),
                                        ((120, 90, 230), 0.2), ((1, 1, 5), 0.5),
                                        ((256, 256, 64), 0.3)])
def test_flood_vs_oracle_random(sk, shape, fill):
    gen = torch.Generator().manual_seed(shape[0] * 7 + shape[2])
    v = (torch.rand(shape, generator=gen) < fill).to(torch.int16)
    want = O.efficient_flood_fill(v.clone().unsqueeze(0)).numpy().astype(np.int32)
    got = sk.F.efficient_flood_fill(v.unsqueeze(0).to(DEV)).cpu().numpy()
    if want.max() < 32767 and np.array_equal(got, want):
        return
    # dense random fields can trip the reference's sum/product seam heuristic or wrap int16:
    # then only the TRUE partition is required
    ref = O.true_ccl_partition(v.numpy())
    pairs = np.unique(np.stack([got.ravel().astype(np.int64), ref.ravel().astype(np.int64)]), axis=1)
    assert len(np.unique(pairs[0])) == pairs.shape[1] == len(np.unique(pairs[1]))


@pytest.mark.parametrize("shape,fill", [((24, 20, 64), 0.02), ((24, 20, 64), 0.35), ((40, 33, 128), 0.10), ((3, 5, 16), 0.6)])
def test_ccl_mask_driven_path_equals_generic_path_and_scipy(shape, fill):
    """sk_ccl_crop has two implementations: the mask-driven one (16 voxels per thread, parent array touched at
    foreground voxels only; crops whose rows are 16-byte aligned) and the one-thread-per-voxel one.  The same mask through
    both -- the second time embedded at z0 = 8 in a wider volume, which un-aligns its rows -- must give the same labels,
    and those are scipy.ndimage.label's numbering (components by first voxel in C order, flood_fill.py:135) + the id
    offset of flood_all (:45-47, first id 3)."""
    import scipy.ndimage
    from skoots_amd import _ffi
    X, Y, Z = shape
    gen = torch.Generator().manual_seed(Z + X)
    m = (torch.rand(shape, generator=gen) < fill).to(torch.uint8)
    if Z > 16:   # runs across a 16-voxel chunk boundary
        m[:, :, 15:17] |= (torch.rand((X, Y, 2), generator=gen) < 0.5).to(torch.uint8)
    want, k = scipy.ndimage.label(m.numpy())
    want = np.where(want > 0, want + 2, 0).astype(np.int32)
    dev = torch.device(DEV)
    st = _ffi.stream_ptr(dev)

    def run(vol, z0, d):
        Xv, Yv, Zv = vol.shape
        labels = torch.full((Xv, Yv, Zv), -7, dtype=torch.int32, device=dev)
        ws_bytes = _ffi.lib.sk_ccl_workspace_bytes(Xv * Yv * d)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        state = torch.tensor([1, 0, 0, 0], dtype=torch.int32, device=dev)
        _ffi.check(_ffi.lib.sk_ccl_crop(_ffi.ptr(vol), _ffi.ptr(labels), Xv, Yv, Zv, 0, 0, z0, Xv, Yv, d, _ffi.ptr(ws), ws_bytes,
                                        _ffi.ptr(state), st))
        torch.cuda.synchronize()
        return labels[:, :, z0:z0 + d].cpu().numpy(), state.cpu().tolist()

    fast, st_fast = run(m.to(dev).contiguous(), 0, Z)
    wide = torch.zeros((X, Y, Z + 16), dtype=torch.uint8)
    wide[:, :, 8:8 + Z] = m
    slow, st_slow = run(wide.to(dev).contiguous(), 8, Z)
    assert np.array_equal(fast, want) and np.array_equal(slow, want)
    assert st_fast == st_slow and st_fast[1] == k


# ----------------------------------------------------------------------------- whole post-model path
def _inject_from(out_vol_dev):
    def inject(_, origin, eff):
        x, y, z = origin
        return out_vol_dev[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]].contiguous()
    return inject


def test_postmodel_golden(sk, golden):
    g = golden("postmodel.npz")
    out_vol = torch.from_numpy(g["out"]).to(DEV)
    X, Y, Z = out_vol.shape[1:]
    image = torch.zeros((X, Y, Z), dtype=torch.float16, device=DEV)
    res = sk.E.eval_volume(image, None, g["scale"].tolist(), mean=0.0, std=1.0,
                           inject=_inject_from(out_vol), keep_planar_vectors=True)
    st = res["state"]
    assert np.array_equal(st.vec_planar.cpu().numpy().view(np.uint16), g["vectors"].view(np.uint16))
    assert np.array_equal(st.vectors_planar().cpu().numpy().view(np.uint16), g["vectors"].view(np.uint16))
    assert np.array_equal(st.skeleton.cpu().numpy(), g["skeleton"][0])
    assert np.array_equal(st.labels.cpu().numpy(), g["labels"].astype(np.int32))
    want_final, _ = O.renumber(g["instance_raw"])
    assert np.array_equal(res["instance_mask"].cpu().numpy(), want_final)
    assert res["n_instances"] == 5


def test_assign_raw_matches_reference_before_renumber(sk, golden):
    g = golden("postmodel.npz")
    X, Y, Z = g["labels"].shape
    st = sk.E.VolumeState((X, Y, Z), DEV)
    v = torch.from_numpy(g["vectors"]).to(DEV)
    sk.ffi.check(sk.ffi.lib.sk_vec_interleave(sk.ffi.ptr(v), sk.ffi.ptr(st.vec4), X * Y * Z,
                                              sk.ffi.stream_ptr()))
    for dt in (torch.int16, torch.int32):
        st.instance = None
        inst = st.assign(g["scale"].tolist(), labels=torch.from_numpy(g["labels"]).to(DEV).to(dt))
        assert np.array_equal(inst.cpu().numpy(), g["instance_raw"].astype(np.int32))


def test_assign_on_a_crop_above_2_pow_24_voxels(sk):
    """A stage-3 crop of more than 2^24 voxels (the reference's 500x500x50 has 12.5 M): the crop-local flat index is then
    NOT exact in fp32 (vector_to_embedding.py:121-127 ravels in fp32), so a voxel with a zero vector can read the vector
    of a NEIGHBOUR whose index rounds to the same float and hop away -- which the two-voxel kernel's shortcut ("zero
    vector: the label at its own position") would miss.  sk_follow_assign must take the one-voxel kernel there
    (round 4: `p.nvox <= 2^24` in the fast path's condition) and equal the oracle bit for bit.  Sparse field, vectors
    only in the high-index corner of the volume where the rounding happens."""
    # 17.0 M voxels > 2^24 = 16.8 M; one crop = the whole volume.  X*Y*Z = 2 (mod 4): the reference clamps the flat index
    # to float(X*Y*Z - 1) (vector_to_embedding.py:125), which must round DOWN -- with X*Y*Z = 0 (mod 4) it rounds up to
    # X*Y*Z and the reference's own `take` raises IndexError at the last voxel, i.e. such crops cannot run there at all
    X, Y, Z = 263, 259, 250
    assert X * Y * Z > 1 << 24 and (X * Y * Z) % 4 == 2
    gen = torch.Generator().manual_seed(11)
    vec = torch.zeros((3, X, Y, Z), dtype=torch.float16)
    blk = (torch.rand((3, 12, 40, Z), generator=gen) * 2 - 1) * 0.2
    keep = torch.rand((1, 12, 40, Z), generator=gen) > 0.5   # half of the voxels keep a vector, their z neighbours none
    vec[:, X - 12:, Y - 40:, :] = (blk * keep).half()
    labels = torch.zeros((X, Y, Z), dtype=torch.int32)
    labels[X - 40:, Y - 80:, :] = torch.randint(1, 500, (40, 80, Z), generator=gen, dtype=torch.int32)
    scale = (60, 60, 12)
    emb = O.vector_to_embedding(torch.tensor(scale), vec.unsqueeze(0), N=O.FOLLOW_N)
    want = O.index_skeleton_by_embed(labels.unsqueeze(0).unsqueeze(0), emb)[0, 0]
    # the case the guard exists for does occur in this field: a zero-vector voxel whose answer is not its own label
    zero = (vec == 0).all(dim=0)
    assert int((zero & (want != labels))[X - 12:, Y - 40:].sum()) > 0
    st = sk.E.VolumeState((X, Y, Z), DEV)
    v = vec.to(DEV)
    sk.ffi.check(sk.ffi.lib.sk_vec_interleave(sk.ffi.ptr(v), sk.ffi.ptr(st.vec4), X * Y * Z, sk.ffi.stream_ptr()))
    inst = st.assign(scale, crop=(X, Y, Z), overlap=(0, 0, 0), labels=labels.to(DEV))
    assert torch.equal(inst.cpu(), want.to(torch.int32))


@pytest.mark.parametrize("shape", [(160, 144, 40), (530, 140, 58), (128, 128, 32)])
def test_postmodel_vs_oracle_blobs(sk, shape):
    """Seeded blob field (SURVEY.md 8d workload generator) through stages 1-tail..renumber."""
    from tests.workload import blob_field
    out_vol, k = blob_field(shape, seed=shape[0], n_blobs=40, rmax=(9, 9, 3))
    image = torch.zeros((1,) + shape, dtype=torch.float16)

    def inject_cpu(_, origin, eff):
        x, y, z = origin
        return out_vol[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]].unsqueeze(0)

    want = O.eval_volume(image, lambda c: None, (60, 60, 12), mean=0.0, std=1.0, inject=inject_cpu)
    res = sk.E.eval_volume(image.to(DEV), None, (60, 60, 12), mean=0.0, std=1.0,
                           inject=_inject_from(out_vol.to(DEV)), keep_planar_vectors=True)
    st = res["state"]
    assert np.array_equal(st.vec_planar.cpu().numpy().view(np.uint16), want["vectors"].view(np.uint16))
    assert np.array_equal(st.skeleton.cpu().numpy(), want["skeleton"][0])
    assert np.array_equal(st.labels.cpu().numpy(), want["labels"].astype(np.int32))
    assert np.array_equal(res["instance_mask"].cpu().numpy(), want["instance_mask"])


def test_renumber_vs_oracle(sk):
    gen = torch.Generator().manual_seed(4)
    for shape, hi in (((37, 29, 11), 50), ((128, 64, 33), 5000), ((5,), 3)):
        lab = torch.randint(0, hi, shape, generator=gen).to(torch.int32)
        lab[torch.rand(shape, generator=gen) < 0.5] = 0
        want, _ = O.renumber(lab.numpy())
        st = sk.E.VolumeState((1, 1, 1), DEV)
        st.instance = lab.to(DEV).clone()
        k = st.renumber()
        assert np.array_equal(st.instance.cpu().numpy(), want)
        assert k == want.max()


def test_renumber_idempotent_and_sorted_first_appearance(sk):
    gen = torch.Generator().manual_seed(11)
    lab = torch.randint(0, 200000, (4_000_000,), generator=gen).to(torch.int32)
    st = sk.E.VolumeState((1, 1, 1), DEV)
    st.instance = lab.to(DEV).clone()
    k = st.renumber()
    once = st.instance.clone()
    # first appearances are 1, 2, 3, ... in memory order
    flat = once.cpu().numpy()
    nz = flat[flat > 0]
    _, first = np.unique(nz, return_index=True)
    assert np.array_equal(np.argsort(first), np.arange(k))
    st.renumber()
    assert torch.equal(st.instance, once)


# ----------------------------------------------------------------------------- C-ABI error behaviour
def test_abi_errors(sk):
    lib, ffi = sk.ffi.lib, sk.ffi
    t = torch.zeros(64, dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError, match="N must be"):
        ffi.check(lib.sk_vector_to_embedding(ffi.ptr(t), 0, ffi.ptr(t), 2, 2, 2,
                                             ffi.float_array([1.0] * 3), 0, None))
    with pytest.raises(ValueError, match="write box"):
        import ctypes
        i3 = ctypes.c_int32 * 3
        ffi.check(lib.sk_gate_dilate_scatter(ffi.ptr(t), 0, 1, (ctypes.c_int64 * 1)(0), 64, 16, 4, 4, 4, 4,
                                             i3(0, 0, 0), i3(2, 2, 2), i3(2, 2, 2), None, None,
                                             ffi.ptr(t), 8, 8, 8, 0.8, 0.8, None))
    with pytest.raises(ValueError, match="outside volume"):
        ffi.check(lib.sk_ccl_crop(ffi.ptr(t), ffi.ptr(t), 4, 4, 4, 2, 0, 0, 4, 4, 4, ffi.ptr(t), 1 << 20,
                                  ffi.ptr(t), None))
