#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own functions.

Run only in the build container (the reference never travels to the GPU box):

    cd /tmp && PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 PYTORCH_JIT=0 \
        python /root/repo/tests/golden/make_golden.py

The reference modules on the hot path are imported unmodified.  Absent third-party
modules that are only touched by module-level ``import`` statements (numba,
skimage, bism, yacs -- SURVEY.md Appendix A) are registered as empty placeholders;
none of them is reached by the functions called here.  ``PYTORCH_JIT=0`` is
needed because the scripted ``binary_dilation`` does not compile on torch 2.10.

Fixtures are data only (inputs + outputs of reference calls), stored as .npz.
  G1 tiling.npz          crops()/get_total_num_crops() origins            cropper.py:8-144
  G2 follow.npz          vector_to_embedding()                            vector_to_embedding.py:79-174
  G3 gather.npz          index_skeleton_by_embed()                        skeleton.py:656-695
  G4 dilate.npz          binary_dilation / binary_dilation_2d chain       morphology.py:155-199, eval.py:145-176
  G5 flood.npz           efficient_flood_fill()                           flood_fill.py:13-261
  G6 postmodel.npz       eval.py:145-284 composed from reference functions on an injected field
  G7 kat.npz             the two __main__ known answers
  G8 loss.npz            config-5 loss terms and their input gradient (tversky + baked_embed_to_prob)
  G9 bake.npz            bake_skeleton (CPU path) + average_baked_skeletons              skeleton.py:18-48,370-528
  G10 validate.npz       mask_iou / accuracies_from_iou / f1_score / get_segmentation_errors   validate/lib.py:170-438
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _ident(*a, **k):
    return a[0] if a and callable(a[0]) else (lambda f: f)


_stub("numba", njit=_ident, prange=range)
_sk = _stub("skimage")
_sk.morphology = _stub("skimage.morphology")
_stub("bism")
for _s in ("backends", "modules", "models", "models.spatial_embedding"):
    _stub("bism." + _s)
sys.modules["bism.models.spatial_embedding"].SpatialEmbedding = object
_y = _stub("yacs")
_y.config = _stub("yacs.config", CfgNode=dict)

from skoots.lib.cropper import crops, get_total_num_crops  # noqa: E402
from skoots.lib.flood_fill import connected_components, efficient_flood_fill  # noqa: E402
from skoots.lib.morphology import binary_dilation, binary_dilation_2d  # noqa: E402
from skoots.lib.skeleton import index_skeleton_by_embed  # noqa: E402
from skoots.lib.vector_to_embedding import vector_to_embedding  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


class _ShapeOnly:
    """crops() only slices and reads .shape; avoid allocating C4-sized arrays."""

    def __init__(self, shape):
        self.shape = shape

    def __getitem__(self, idx):
        return torch.zeros(1)


# ----------------------------------------------------------------------------- G1
def g1():
    cases = [
        ((1, 128, 128, 32), [300, 300, 20], (50, 50, 5)),
        ((1, 512, 512, 128), [300, 300, 20], (50, 50, 5)),
        ((1, 1024, 1024, 256), [300, 300, 20], (50, 50, 5)),
        ((1, 2048, 2048, 512), [300, 300, 20], (50, 50, 5)),
        ((1, 301, 299, 21), [300, 300, 20], (50, 50, 5)),
        ((1, 640, 333, 47), [300, 300, 20], (50, 50, 5)),
        ((3, 128, 128, 32), [500, 500, 50], (50, 50, 5)),
        ((3, 512, 512, 128), [500, 500, 50], (50, 50, 5)),
        ((3, 1024, 1024, 256), [500, 500, 50], (50, 50, 5)),
        ((3, 2048, 2048, 512), [500, 500, 50], (50, 50, 5)),
        ((3, 777, 500, 51), [500, 500, 50], (50, 50, 5)),
        ((1, 1024, 1024, 256), [1000, 1000, 200], (0, 0, 0)),
        ((1, 2048, 2048, 512), [1000, 1000, 200], (0, 0, 0)),
        ((1, 1002, 4, 4), [1000, 1000, 200], (0, 0, 0)),
    ]
    out = {}
    for i, (shape, crop, ov) in enumerate(cases):
        c1 = list(crop)
        total = get_total_num_crops(shape, c1, ov)
        c2 = list(crop)
        origins = [o for _, o in crops(_ShapeOnly(shape), c2, ov)]
        assert len(origins) == total and c1 == c2
        out[f"shape_{i}"] = np.array(shape)
        out[f"crop_{i}"] = np.array(crop)
        out[f"overlap_{i}"] = np.array(ov)
        out[f"eff_{i}"] = np.array(c2)
        out[f"origins_{i}"] = np.array(origins, dtype=np.int32)
    out["n"] = np.array(len(cases))
    save("tiling.npz", **out)


# ----------------------------------------------------------------------------- G2 / G3
def _field(gen, shape, kind):
    """fp16 vector fields in [-1, 1]; 'ties' lands embeddings on .5 boundaries,
    'edge' pushes hard against the crop faces (clamp-to-k wrap quirk)."""
    w, h, d = shape
    v = (torch.rand((1, 3, w, h, d), generator=gen) * 2 - 1)
    if kind == "ties":
        v = (torch.randint(-8, 9, (1, 3, w, h, d), generator=gen).float() / 8)
    elif kind == "edge":
        v = torch.where(torch.rand((1, 3, w, h, d), generator=gen) > 0.5, torch.ones(()), v)
    elif kind == "neg":
        v = torch.where(torch.rand((1, 3, w, h, d), generator=gen) > 0.5, -torch.ones(()), v)
    elif kind == "sparse":
        v = v * (torch.rand((1, 1, w, h, d), generator=gen) > 0.6)
    return v.to(torch.float16)


def g2_g3():
    gen = torch.Generator().manual_seed(1234)
    cases = [
        ((24, 20, 12), (60, 60, 12), 1, 1.0, "rand"),
        ((24, 20, 12), (60, 60, 12), 2, 1.0, "rand"),
        ((24, 20, 12), (60, 60, 12), 10, 1.0, "rand"),
        ((24, 20, 12), (60, 60, 12), 10, 0.95, "rand"),
        ((32, 32, 16), (3, 3, 1), 10, 1.0, "rand"),
        ((32, 32, 16), (4, 4, 2), 10, 1.0, "ties"),
        ((32, 32, 16), (8, 8, 4), 5, 0.95, "ties"),
        ((17, 9, 5), (60, 60, 12), 10, 1.0, "edge"),
        ((17, 9, 5), (60, 60, 12), 10, 1.0, "neg"),
        ((20, 31, 7), (6, 6, 2), 10, 1.0, "sparse"),
        ((8, 8, 8), (1, 1, 1), 3, 0.5, "edge"),
    ]
    out2, out3 = {}, {}
    for i, (shape, scale, n, decay, kind) in enumerate(cases):
        v = _field(gen, shape, kind)
        emb = vector_to_embedding(torch.tensor(scale), v, N=n, decay=decay)
        out2[f"vector_{i}"] = v.numpy()
        out2[f"scale_{i}"] = np.array(scale)
        out2[f"n_{i}"] = np.array(n)
        out2[f"decay_{i}"] = np.array(decay)
        out2[f"embed_{i}"] = emb.numpy()
        # G3: gather from a label volume larger than the crop, with a crop origin offset
        origin = (3 + i, 2, 1)
        vol = tuple(s + 9 for s in shape)
        labels = torch.randint(0, 3000, (1, 1) + vol, generator=gen).to(torch.int16)
        e2 = emb.clone()
        e2 += torch.tensor(origin).view(1, 3, 1, 1, 1)
        got = index_skeleton_by_embed(labels, e2)
        out3[f"labels_{i}"] = labels.numpy()
        out3[f"origin_{i}"] = np.array(origin)
        out3[f"embed_{i}"] = emb.numpy()
        out3[f"out_{i}"] = got.numpy()
    out2["n"] = out3["n"] = np.array(len(cases))
    save("follow.npz", **out2)
    save("gather.npz", **out3)


# ----------------------------------------------------------------------------- G4
def _ref_gate_dilate(out):
    """The reference's per-tile post-ops (eval.py:145-157) expressed through its
    own morphology functions: gate by prob>0.8, one 3-D and two 2-D dilations."""
    prob, skel, vec = out[:, [-1]], out[:, [-2]].float(), out[:, 0:3]
    gate = prob.gt(0.8)
    skel = binary_dilation(skel * gate)
    skel = binary_dilation_2d(binary_dilation_2d(skel))
    return vec * gate, skel


def _interior(origin, size, margin):
    return tuple(slice(o + m, o + s - m) for o, s, m in zip(origin, size, margin))


def g4():
    gen = torch.Generator().manual_seed(77)
    out = {}
    k = 0
    for dtype in (torch.float32, torch.float16):
        for shape in ((20, 18, 8), (9, 33, 5)):
            o = torch.rand((1, 5) + shape, generator=gen)
            o[:, 0:3] = o[:, 0:3] * 2 - 1
            # sprinkle exact-threshold values
            o[:, 3:5][torch.rand((1, 2) + shape, generator=gen) > 0.97] = 0.8
            o = o.to(dtype)
            vec, sk = _ref_gate_dilate(o)
            out[f"out_{k}"] = o.numpy()
            out[f"vec_{k}"] = vec.half().numpy()
            out[f"skel_{k}"] = sk.gt(0.8).numpy().astype(np.uint8)
            out[f"skelmap_{k}"] = sk.numpy()
            k += 1
    out["n"] = np.array(k)
    save("dilate.npz", **out)


# ----------------------------------------------------------------------------- G5
def _blobs(gen, shape, n, rmax):
    vol = torch.zeros(shape, dtype=torch.int16)
    for _ in range(n):
        c = [int(torch.randint(0, s, (1,), generator=gen)) for s in shape]
        r = [int(torch.randint(1, m + 1, (1,), generator=gen)) for m in rmax]
        sl = tuple(slice(max(0, ci - ri), min(s, ci + ri + 1)) for ci, ri, s in zip(c, r, shape))
        vol[sl] = 1
    return vol


def g5():
    gen = torch.Generator().manual_seed(5)
    out = {}
    cases = []
    # (a) single-crop volumes
    cases.append(_blobs(gen, (40, 36, 20), 25, (2, 2, 1)).unsqueeze(0))
    cases.append((torch.rand((1, 24, 24, 12), generator=gen) > 0.7).to(torch.int16))
    # (b) z seam at 200 (SURVEY: bar crossing the seam gets one label)
    s = torch.zeros((1, 6, 6, 210), dtype=torch.int16)
    s[0, 1, 1, 190:205] = 1
    s[0, 3, 3, 0:5] = 1
    s[0, 4, 4, 203:208] = 1
    s[0, 5, 0:6, 195] = 1
    cases.append(s)
    v = _blobs(gen, (8, 8, 230), 30, (1, 1, 6)).unsqueeze(0)
    cases.append(v)
    # (c) x / y seams via volumes just over 1000
    v = _blobs(gen, (1010, 6, 5), 60, (9, 1, 1)).unsqueeze(0)
    cases.append(v)
    v = _blobs(gen, (5, 1013, 6), 60, (1, 9, 1)).unsqueeze(0)
    cases.append(v)
    # (d) two seam axes at once (x+z, y+z)
    v = _blobs(gen, (1003, 5, 204), 70, (6, 1, 5)).unsqueeze(0)
    cases.append(v)
    v = _blobs(gen, (4, 1003, 203), 70, (1, 6, 5)).unsqueeze(0)
    cases.append(v)
    # (e) empty volume and full volume
    cases.append(torch.zeros((1, 10, 10, 10), dtype=torch.int16))
    cases.append(torch.ones((1, 7, 5, 3), dtype=torch.int16))
    for i, c in enumerate(cases):
        inp = c.clone()
        res = efficient_flood_fill(c.clone())
        out[f"in_{i}"] = inp.numpy().astype(np.uint8)
        out[f"out_{i}"] = res.numpy()
    out["n"] = np.array(len(cases))
    save("flood.npz", **out)


# ----------------------------------------------------------------------------- G6
def g6():
    """eval.py:126-284 composed from reference functions on an injected blob field
    (the network is replaced by a synthetic `out`; everything else is reference code)."""
    gen = torch.Generator().manual_seed(2024)
    X, Y, Z = 140, 132, 34
    scale = (60, 60, 12)
    # blob field: prob=1 inside ellipsoids, vec points at the centre, skeleton = small ball
    xs, ys, zs = torch.meshgrid(torch.arange(X), torch.arange(Y), torch.arange(Z), indexing="ij")
    out_vol = torch.zeros((5, X, Y, Z))
    centres = [(60, 58, 12), (75, 70, 20), (62, 76, 9), (55, 66, 22), (70, 56, 16)]
    radii = [(7, 6, 4), (6, 7, 3), (5, 5, 3), (4, 6, 3), (6, 4, 4)]
    for (cx, cy, cz), (rx, ry, rz) in zip(centres, radii):
        inside = ((xs - cx) / rx) ** 2 + ((ys - cy) / ry) ** 2 + ((zs - cz) / rz) ** 2 <= 1.0
        out_vol[4][inside] = 0.95
        out_vol[0][inside] = ((cx - xs) / scale[0]).clamp(-1, 1)[inside]
        out_vol[1][inside] = ((cy - ys) / scale[1]).clamp(-1, 1)[inside]
        out_vol[2][inside] = ((cz - zs) / scale[2]).clamp(-1, 1)[inside]
        core = (xs - cx) ** 2 + (ys - cy) ** 2 + (zs - cz) ** 2 <= 1
        out_vol[3][core] = 0.9
    out_vol += (torch.rand(out_vol.shape, generator=gen) - 0.5) * 0.02  # noise
    out_vol = out_vol.to(torch.float16)

    vectors = np.zeros((3, X, Y, Z), dtype=np.float16)
    skeleton = np.zeros((1, X, Y, Z), dtype=np.uint8)
    cropsize = [300, 300, 20]
    overlap = [50, 50, 5]
    image = torch.zeros((1, X, Y, Z), dtype=torch.float16)
    for _, org in crops(image, cropsize, overlap, device="cpu"):
        win = tuple(slice(o, o + c) for o, c in zip(org, cropsize))
        vec, skel = _ref_gate_dilate(out_vol[(slice(None),) + win].unsqueeze(0))
        dst = _interior(org, cropsize, overlap)
        src = _interior((0, 0, 0), cropsize, overlap)
        vectors[(slice(None),) + dst] = vec[0][(slice(None),) + src].half().numpy()  # eval.py:175
        skeleton[(slice(None),) + dst] = skel[0][(slice(None),) + src].gt(0.8).numpy()  # eval.py:176

    labels = efficient_flood_fill(torch.from_numpy(skeleton[...]).to(torch.int16))
    instance_mask = torch.zeros_like(labels, dtype=torch.int16)
    lab5 = labels.unsqueeze(0).unsqueeze(0)
    cropsize = [500, 500, 50]
    overlap = (50, 50, 5)
    for field, org in crops(vectors, crop_size=cropsize, overlap=overlap):
        emb = vector_to_embedding(scale=torch.tensor(scale), vector=field, N=10)  # eval.py:271-273
        emb += torch.tensor(org).view(1, 3, 1, 1, 1)  # eval.py:274-276
        got = index_skeleton_by_embed(skeleton=lab5, embed=emb).squeeze()  # eval.py:277-279
        instance_mask[_interior(org, cropsize, overlap)] = got[_interior((0, 0, 0), cropsize, overlap)]
    save("postmodel.npz", out=out_vol.numpy(), scale=np.array(scale), vectors=vectors,
         skeleton=skeleton, labels=labels.numpy(), instance_raw=instance_mask.numpy())


# ----------------------------------------------------------------------------- G7
def g7():
    vector = torch.ones((1, 3, 10, 10, 10)).float()
    vector[:, :, 5, 5, 5] = -1
    vector[:, [0, 1, 2], 4, 4, 4] = torch.tensor((2.0, 2.0, 2.0))
    o = vector_to_embedding(torch.tensor((1, 1, 1)), vector, N=2)
    graph = {1: [2, 3], 2: [1], 3: [1, 5, 4], 4: [5], 5: [3], 6: [7], 7: [6, 8, 9], 8: [7], 9: [7]}
    cc = connected_components(graph)
    assert o[0, :, 5, 5, 5].tolist() == [6.0, 6.0, 6.0]
    assert cc == [[1, 2, 3, 5, 4], [6, 7, 8, 9]]
    save("kat.npz", vector=vector.numpy(), embed=o.numpy(),
         cc0=np.array(cc[0]), cc1=np.array(cc[1]))


# ----------------------------------------------------------------------------- G8 (config 5: loss)
def g8():
    """Training-loss pieces of the config-5 step (train/engine.py:461-496): values and input
    gradients of the reference's tversky (train/loss.py:95-212) and baked_embed_to_prob
    (lib/embedding_to_prob.py:5-51) composed exactly as the engine composes them."""
    from skoots.lib.embedding_to_prob import baked_embed_to_prob
    from skoots.train.loss import tversky
    gen = torch.Generator().manual_seed(88)
    B, X, Y, Z = 2, 12, 10, 8
    out = torch.rand((B, 5, X, Y, Z), generator=gen)
    out[:, 0:3] = out[:, 0:3] * 2 - 1
    out = out.detach().requires_grad_(True)
    masks = (torch.rand((B, 1, X, Y, Z), generator=gen) > 0.6).float() * torch.randint(1, 5, (B, 1, X, Y, Z), generator=gen)
    skele = (torch.rand((B, 1, X, Y, Z), generator=gen) > 0.85).float()
    baked = torch.rand((B, 3, X, Y, Z), generator=gen) * torch.tensor([X, Y, Z]).view(1, 3, 1, 1, 1)
    scale = torch.tensor((60, 60, 12))
    sigma = torch.tensor([20.0, 20.0, 20.0])
    l_embed, l_prob, l_skel = tversky(0.25, 0.75, 1e-8), tversky(0.5, 0.5, 1e-8), tversky(0.5, 1.5, 1e-8)
    prob, vec, sk = out[:, [-1]], out[:, 0:3], out[:, [-2]]
    emb = vector_to_embedding(scale, vec)
    pe = baked_embed_to_prob(emb, baked, sigma)
    le = l_embed(pe, masks.gt(0).float())
    lp = l_prob(prob, masks.gt(0).float())
    ls = l_skel(sk, skele.gt(0).float())
    loss = 1.0 * le + 1.0 * lp + 1.0 * ls
    loss.backward()
    save("loss.npz", out=out.detach().numpy(), masks=masks.numpy(), skele=skele.numpy(), baked=baked.numpy(),
         scale=scale.numpy(), sigma=sigma.numpy(), embed_prob=pe.detach().numpy(),
         losses=np.array([le.item(), lp.item(), ls.item(), loss.item()]), grad=out.grad.numpy())


# ----------------------------------------------------------------------------- G9 (next row N3: target baking)
def g9():
    """Training-target baking: the reference's CPU path of bake_skeleton (lib/skeleton.py:370-445,
    448-528) and average_baked_skeletons (lib/skeleton.py:18-48) on a small instance mask."""
    from skoots.lib.skeleton import average_baked_skeletons, bake_skeleton
    gen = torch.Generator().manual_seed(99)
    X, Y, Z = 26, 22, 12
    masks = torch.zeros((X, Y, Z), dtype=torch.int64)
    masks[2:12, 3:14, 1:9] = 1
    masks[13:24, 2:11, 2:11] = 4
    masks[14:25, 12:21, 0:6] = 7
    masks[3:6, 16:20, 9:12] = 9          # instance with a single skeleton point
    skeletons = {}
    for i in (1, 4, 7, 9):
        nz = masks.eq(i).nonzero()
        k = 1 if i == 9 else 6
        pick = torch.randperm(nz.shape[0], generator=gen)[:k]
        skeletons[i] = nz[pick].float()
    anis = (1.0, 1.0, 3.0)
    raw = bake_skeleton(masks.clone(), skeletons, anis, average=False, device="cpu")
    avg = bake_skeleton(masks.clone(), skeletons, anis, average=True, device="cpu")
    ids = np.array(sorted(skeletons), dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum([skeletons[int(i)].shape[0] for i in ids])]).astype(np.int32)
    pts = np.concatenate([skeletons[int(i)].numpy() for i in ids]).astype(np.float32)
    save("bake.npz", masks=masks.numpy().astype(np.int32), ids=ids, offsets=offs, points=pts,
         anisotropy=np.array(anis, dtype=np.float32), baked=raw.numpy(), baked_avg=avg.numpy())


# ----------------------------------------------------------------------------- G10 (next row N4: validation metrics)
def g10():
    """Instance-level validation metrics: mask_iou, accuracies_from_iou, f1_score and
    get_segmentation_errors of skoots/validate/lib.py:170-232,358-438 on two small label volumes."""
    if "skimage.io" not in sys.modules:
        sys.modules["skimage.io"] = types.ModuleType("skimage.io")
    from skoots.validate.lib import accuracies_from_iou, f1_score, get_segmentation_errors, mask_iou
    gen = torch.Generator().manual_seed(123)
    X, Y, Z = 30, 28, 10
    gt = torch.zeros((1, X, Y, Z), dtype=torch.int32)
    pred = torch.zeros((1, X, Y, Z), dtype=torch.int32)
    gt[0, 2:10, 2:12, 1:8] = 3
    gt[0, 12:22, 3:12, 0:9] = 5
    gt[0, 3:14, 15:26, 2:9] = 8
    gt[0, 18:28, 16:27, 1:6] = 11          # missed by the prediction
    pred[0, 3:11, 2:12, 1:8] = 2           # good match of 3
    pred[0, 12:17, 3:12, 0:9] = 4          # gt 5 split in two
    pred[0, 17:22, 3:12, 0:9] = 6
    pred[0, 3:14, 15:26, 2:5] = 9          # partial match of 8
    pred[0, 24:29, 1:6, 6:10] = 12         # false positive
    pred[0, 2:6, 13:17, 0:3] = 13          # straddles gt 8 and background
    noise = torch.rand((1, X, Y, Z), generator=gen) > 0.97
    pred[noise] = 0
    iou = mask_iou(gt, pred)
    acc = {str(t): accuracies_from_iou(iou, t) for t in (0.1, 0.5, 0.75)}
    over, under = get_segmentation_errors(gt, pred)
    tp, fp, fn = acc["0.5"]
    save("validate.npz", gt=gt.numpy(), pred=pred.numpy(), iou=iou.numpy(),
         acc=np.array([acc[k] for k in ("0.1", "0.5", "0.75")], dtype=np.float64),
         f1=np.array([f1_score(tp, fp, fn)]), seg_errors=np.array([over, under]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    g1(); g2_g3(); g4(); g5(); g6(); g7(); g8(); g9(); g10()
