"""Pin the CPU oracle (oracle/pipeline.py) to the fixtures produced by the reference's
own functions (tests/golden/make_golden.py).  Bit-exact everywhere."""
import numpy as np
import torch

from oracle import pipeline as O


def test_g1_tiling(golden):
    g = golden("tiling.npz")
    for i in range(int(g["n"])):
        shape = g[f"shape_{i}"].tolist()
        origins, eff = O.crop_origins(shape[1:], g[f"crop_{i}"].tolist(), g[f"overlap_{i}"].tolist())
        assert eff == g[f"eff_{i}"].tolist()
        assert np.array_equal(np.array(origins, dtype=np.int32), g[f"origins_{i}"])
        assert O.get_total_num_crops(shape, g[f"crop_{i}"].tolist(), g[f"overlap_{i}"].tolist()) == len(origins)


def test_g1_known_counts():
    # tile / crop counts computed with the reference's generator
    for shape, n1, n3 in (((128, 128, 32), 100, 50), ((512, 512, 128), 117, 16),
                          ((1024, 1024, 256), 936, 63), ((2048, 2048, 512), 6292, 468)):
        assert len(O.crop_origins(shape, [300, 300, 20], (50, 50, 5))[0]) == n1
        assert len(O.crop_origins(shape, [500, 500, 50], (50, 50, 5))[0]) == n3


def test_g2_follow(golden):
    g = golden("follow.npz")
    for i in range(int(g["n"])):
        emb = O.vector_to_embedding(torch.tensor(g[f"scale_{i}"]), torch.from_numpy(g[f"vector_{i}"]),
                                    N=int(g[f"n_{i}"]), decay=float(g[f"decay_{i}"]))
        assert emb.dtype == torch.float32
        assert np.array_equal(emb.numpy(), g[f"embed_{i}"]), f"case {i}"


def test_g3_gather(golden):
    g = golden("gather.npz")
    for i in range(int(g["n"])):
        emb = torch.from_numpy(g[f"embed_{i}"]).clone()
        emb += torch.tensor(g[f"origin_{i}"]).view(1, 3, 1, 1, 1)
        got = O.index_skeleton_by_embed(torch.from_numpy(g[f"labels_{i}"]), emb)
        assert got.dtype == torch.int32
        assert np.array_equal(got.numpy(), g[f"out_{i}"]), f"case {i}"


def test_g4_gate_dilate(golden):
    g = golden("dilate.npz")
    for i in range(int(g["n"])):
        out = torch.from_numpy(g[f"out_{i}"])
        vec, sk = O.gate_dilate(out)
        assert np.array_equal(vec.half().numpy().view(np.uint16), g[f"vec_{i}"].view(np.uint16))
        assert np.array_equal(sk.numpy(), g[f"skelmap_{i}"])
        assert np.array_equal(sk.gt(0.8).numpy().astype(np.uint8), g[f"skel_{i}"])


def test_g5_flood(golden):
    g = golden("flood.npz")
    for i in range(int(g["n"])):
        inp = torch.from_numpy(g[f"in_{i}"].astype(np.int16))
        got = O.efficient_flood_fill(inp)
        assert got.dtype == torch.int16
        assert np.array_equal(got.numpy(), g[f"out_{i}"]), f"case {i}"


def test_g5_first_label_is_3(golden):
    # first crop labels start at 3
    v = torch.zeros((1, 5, 5, 5), dtype=torch.int16)
    v[0, 1, 1, 1] = 1
    v[0, 3, 3, 3] = 1
    assert O.efficient_flood_fill(v).unique().tolist() == [0, 3, 4]


def test_g6_postmodel(golden):
    g = golden("postmodel.npz")
    out_vol = torch.from_numpy(g["out"])
    X, Y, Z = out_vol.shape[1:]
    image = torch.zeros((1, X, Y, Z), dtype=torch.float16)

    def inject(_, origin, eff):
        x, y, z = origin
        return out_vol[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]].unsqueeze(0)

    vectors, skeleton = O.stage1(image, lambda c: None, 0.0, 1.0, inject=inject)
    assert np.array_equal(vectors.view(np.uint16), g["vectors"].view(np.uint16))
    assert np.array_equal(skeleton, g["skeleton"])
    res = O.post_model(vectors, skeleton, g["scale"].tolist())
    assert np.array_equal(res["labels"], g["labels"])
    assert np.array_equal(res["instance_raw"], g["instance_raw"])
    # analytic property of the blob field: five blobs -> five instances
    assert res["instance_mask"].max() == 5
    assert len(np.unique(res["instance_mask"])) == 6


def test_g7_kat(golden):
    g = golden("kat.npz")
    emb = O.vector_to_embedding(torch.tensor((1, 1, 1)), torch.from_numpy(g["vector"]), N=2)
    assert emb[0, :, 5, 5, 5].tolist() == [6.0, 6.0, 6.0]  # vector_to_embedding.py:221-232
    assert np.array_equal(emb.numpy(), g["embed"])
    graph = {1: [2, 3], 2: [1], 3: [1, 5, 4], 4: [5], 5: [3], 6: [7], 7: [6, 8, 9], 8: [7], 9: [7]}
    assert O.connected_components(graph) == [[1, 2, 3, 5, 4], [6, 7, 8, 9]]  # flood_fill.py:264-277


def test_renumber_definition():
    a = np.array([[0, 7, 7], [3, 0, 7], [9, 3, -2]], dtype=np.int16)
    out, m = O.renumber(a)
    assert out.tolist() == [[0, 1, 1], [2, 0, 1], [3, 2, 4]]
    assert m == {7: 1, 3: 2, 9: 3, -2: 4, 0: 0}


def test_flood_equals_true_ccl_when_no_quirk():
    gen = torch.Generator().manual_seed(3)
    v = (torch.rand((1, 30, 30, 210), generator=gen) > 0.75).to(torch.int16)
    got = O.efficient_flood_fill(v.clone()).numpy()
    ref = O.true_ccl_partition(v[0].numpy())
    # same partition: label pairs are in bijection
    pairs = np.unique(np.stack([got.ravel().astype(np.int64), ref.ravel().astype(np.int64)]), axis=1)
    assert len(np.unique(pairs[0])) == pairs.shape[1] == len(np.unique(pairs[1]))


def test_16bit_storage_emulation_of_the_training_step_is_the_plain_step_at_fp32():
    """oracle.train_step_16bit_storage (the checker of the HIP bf16 step) with dtype float32 rounds nothing: it must
    reproduce oracle.train_step -- the same graph, hand-written GroupNorm included -- and with bfloat16 it must stay
    within bf16's error of it."""
    import copy

    import torch
    from oracle import train_step as O
    from oracle import unet_spec
    ref = unet_spec.build().train()
    twins = [copy.deepcopy(ref) for _ in range(2)]
    g = torch.Generator().manual_seed(40)
    B, X, Y, Z = 1, 16, 12, 8
    images = torch.randn((B, 1, X, Y, Z), generator=g)
    masks = (torch.rand((B, 1, X, Y, Z), generator=g) > 0.5).float()
    skele = (torch.rand((B, 1, X, Y, Z), generator=g) > 0.9).float()
    baked = torch.rand((B, 3, X, Y, Z), generator=g) * 16
    sigma, scale = torch.tensor([20.0, 20.0, 20.0]), torch.tensor((60, 60, 12))
    want = O.train_step(ref, O.make_optimizer(ref), images, masks, skele, baked, sigma, scale)
    grads = {k: p.grad.clone() for k, p in ref.named_parameters()}
    for twin, dtype, tol_l, tol_g in ((twins[0], torch.float32, 1e-6, 1e-4), (twins[1], torch.bfloat16, 5e-3, 0.15)):
        got = O.train_step_16bit_storage(twin, O.make_optimizer(twin), images, masks, skele, baked, sigma, scale, dtype=dtype)
        assert (got - want).abs().max().item() <= tol_l
        for k, p in twin.named_parameters():
            assert (p.grad - grads[k]).abs().max() <= tol_g * grads[k].abs().max(), (dtype, k)
