"""CPU restatement of the SKOOTS eval hot path (everything except the network body).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function cites the
reference file:line it restates (paths relative to the reference tree root).
Parity pinning: each function here is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the
reference's own functions (``tests/golden/make_golden.py``).  Exceptions are
``renumber`` (third-party fastremap, absent -> "parity unpinned") and the network
body (third-party bism, absent -> see ``oracle/unet_spec.py``).

All arithmetic is done with the same torch CPU ops / dtypes the reference uses so
that float rounding (fp32 products, half-to-even ``round``) is bit-identical.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from scipy import ndimage
from torch import Tensor

# --------------------------------------------------------------------------- #
# Tiling  (skoots/lib/cropper.py:8-55 and 58-144)
# --------------------------------------------------------------------------- #


def effective_crop(spatial: Sequence[int], crop: Sequence[int]) -> List[int]:
    """cropper.py:13-16 / 81-84 -- a crop edge never exceeds the image edge.

    (The reference mutates the caller's list in place; callers of the oracle get
    the clamped list back instead.)
    """
    return [int(c) if c < s else int(s) for c, s in zip(crop, spatial)]


def crop_origins(
    spatial: Sequence[int], crop: Sequence[int], overlap: Sequence[int]
) -> Tuple[List[Tuple[int, int, int]], List[int]]:
    """Origins emitted by ``crops`` in generator order (cropper.py:97-144).

    Loop variables advance by ``crop - 2*overlap`` while ``< dim``; the emitted
    origin is clamped to ``dim - crop``.  Order: x outer, y, z inner.
    Returns (origins, effective crop size).
    """
    eff = effective_crop(spatial, crop)
    for c, o in zip(eff, overlap):
        assert c - 2 * o > 0, "overlap must be smaller than half the crop"
    axes = []
    for dim, c, o in zip(spatial, eff, overlap):
        pos, v = [], 0
        while v < dim:
            pos.append(v if v + c <= dim else dim - c)
            v += c - 2 * o
        axes.append(pos)
    out = [(x, y, z) for x in axes[0] for y in axes[1] for z in axes[2]]
    return out, eff


def get_total_num_crops(image_shape, crop_size, overlap) -> int:
    """cropper.py:8-55 (image_shape is (C, X, Y, Z))."""
    return len(crop_origins(list(image_shape)[1:], crop_size, overlap)[0])


# --------------------------------------------------------------------------- #
# Offset following  (skoots/lib/vector_to_embedding.py:79-132, 135-174)
# --------------------------------------------------------------------------- #


def vector_to_embedding(scale: Tensor, vector: Tensor, N: int = 1, decay: float = 1.0) -> Tensor:
    """phi = index + v*s, then N-1 follow steps (vector_to_embedding.py:79-132).

    ``vector`` is (1, 3, W, H, D) (fp16 or fp32); returns fp32.
    """
    assert vector.ndim == 5 and vector.shape[0] == 1 and vector.shape[1] == 3
    _, _, w, h, d = vector.shape
    num = scale.float().reshape(1, 3, 1, 1, 1)
    grid = torch.stack(
        torch.meshgrid(
            torch.arange(w, dtype=torch.float32),
            torch.arange(h, dtype=torch.float32),
            torch.arange(d, dtype=torch.float32),
            indexing="ij",
        )
    ).unsqueeze(0)
    emb = grid + vector * num  # :104-105 (fp32 by promotion)
    strength = 1.0
    for _ in range(N - 1):
        strength *= decay  # :113 python double
        step = vector * (strength * num)  # :115 scalar cast to fp32, then fp32 products
        q = emb.round()  # :117 half-to-even
        for axis, bound in enumerate((w, h, d)):
            q[:, axis] = q[:, axis].clamp(0, bound)  # :119 inclusive of `bound` (quirk)
        # :122-126 ravel in fp32, left-to-right
        flat = (q[:, [0]] * h * d) + (q[:, [1]] * d) + q[:, [2]]
        flat = flat.clamp(0, w * h * d - 1).long()  # :127
        for c in range(3):
            emb[:, [c]] = emb[:, [c]] + step[:, [c]].take(flat)  # :129-130
    return emb


# --------------------------------------------------------------------------- #
# Label gather  (skoots/lib/skeleton.py:656-695)
# --------------------------------------------------------------------------- #


def index_skeleton_by_embed(skeleton: Tensor, embed: Tensor) -> Tensor:
    """round -> clamp to the label volume -> gather (skeleton.py:675-695)."""
    assert skeleton.ndim == 5 and embed.ndim == 5
    _, c, x, y, z = embed.shape
    e = embed.reshape(c, -1).round()
    xi = e[0].clamp(0, skeleton.shape[2] - 1).long()
    yi = e[1].clamp(0, skeleton.shape[3] - 1).long()
    zi = e[2].clamp(0, skeleton.shape[4] - 1).long()
    return skeleton[0, 0][xi, yi, zi].to(torch.int32).reshape(1, 1, x, y, z)


# --------------------------------------------------------------------------- #
# Morphology  (skoots/lib/morphology.py:155-175, 178-199)
# --------------------------------------------------------------------------- #


def _max_filter_zero_pad(image: Tensor, radius: Tuple[int, int, int]) -> Tensor:
    """max over a box window with zero padding.

    morphology.py:170-175 builds the window with one-hot conv3d kernels
    (zero padded) and takes the channel max; that is a max over the window
    where out-of-bounds taps contribute 0.0.
    """
    assert image.ndim == 5
    rx, ry, rz = radius
    x = torch.nn.functional.pad(image, (rz, rz, ry, ry, rx, rx), value=0.0)
    w, h, d = image.shape[2:]
    out = None
    for dx in range(2 * rx + 1):
        for dy in range(2 * ry + 1):
            for dz in range(2 * rz + 1):
                win = x[:, :, dx : dx + w, dy : dy + h, dz : dz + d]
                out = win.clone() if out is None else torch.maximum(out, win)
    return out


def binary_dilation(image: Tensor) -> Tensor:
    """3x3x3 max filter, zero padded (morphology.py:155-175)."""
    return _max_filter_zero_pad(image, (1, 1, 1))


def binary_dilation_2d(image: Tensor) -> Tensor:
    """3x3x1 max filter, zero padded (morphology.py:178-199)."""
    return _max_filter_zero_pad(image, (1, 1, 0))


# --------------------------------------------------------------------------- #
# Skeleton labelling  (skoots/lib/flood_fill.py:13-261)
# --------------------------------------------------------------------------- #

FLOOD_CROP = (1000, 1000, 200)  # flood_fill.py:28


def _i16(a) -> np.ndarray:
    return np.asarray(a).astype(np.int16)


def get_adjacent_labels(p0: np.ndarray, p1: np.ndarray) -> List[Tuple[int, int]]:
    """flood_fill.py:237-261 -- sum/product membership heuristic, int16 arithmetic."""
    p0 = _i16(p0)
    p1 = _i16(p1)
    with np.errstate(over="ignore"):
        sums = set(np.unique((p0 + p1).astype(np.int16)).tolist())
        prods = set(np.unique((p0 * p1).astype(np.int16)).tolist())
    pairs = []
    for a in np.unique(p0):
        for b in np.unique(p1):
            if a == 0 or b == 0:
                continue
            with np.errstate(over="ignore"):
                s = int(np.int16(a) + np.int16(b))
                p = int(np.int16(a) * np.int16(b))
            if s in sums and p in prods:
                pairs.append((int(a), int(b)))
    return pairs


def connected_components(graph: Dict[int, List[int]]) -> List[List[int]]:
    """flood_fill.py:143-174 -- depth-first, nodes in dict order, edges in list order.

    (Iterative, but visits in exactly the recursive pre-order of the reference.)
    """
    seen = {n: False for n in graph}
    comps: List[List[int]] = []
    for root in graph:
        if seen[root]:
            continue
        comp: List[int] = []
        stack: List[Tuple[int, int]] = []
        seen[root] = True
        comp.append(root)
        stack.append((root, 0))
        while stack:
            node, i = stack.pop()
            nbrs = graph[node]
            while i < len(nbrs) and seen[nbrs[i]]:
                i += 1
            if i < len(nbrs):
                stack.append((node, i + 1))
                nxt = nbrs[i]
                seen[nxt] = True
                comp.append(nxt)
                stack.append((nxt, 0))
        comps.append(comp)
    return comps


def efficient_flood_fill(skeleton: Tensor) -> Tensor:
    """flood_fill.py:13-122.

    ``skeleton``: (1, X, Y, Z) or (X, Y, Z) int16, modified in place like the
    reference; returns the (X, Y, Z) labelled volume (labels not sequential).
    Quirks kept: the running id resets after an empty crop (flood_fill.py:138-140
    returns ``mask.max()`` == 0), int16 storage wraps, the seam test is the
    sum/product heuristic.
    """
    assert skeleton.dtype == torch.int16
    vol = skeleton.unsqueeze(0) if skeleton.ndim == 3 else skeleton
    arr = vol[0].numpy()  # shares memory
    origins, (cw, ch, cd) = crop_origins(arr.shape, list(FLOOD_CROP), (0, 0, 0))

    running = 1
    seams: List[List[int]] = [[], [], []]
    for (x, y, z) in origins:
        for axis, v in enumerate((x, y, z)):
            if v not in seams[axis]:
                seams[axis].append(v)
        fg = arr[x : x + cw, y : y + ch, z : z + cd] > 0
        lab, _ = ndimage.label(fg)  # :135 default 6-connectivity
        lab = lab.astype(np.int16).astype(np.int32) + fg.astype(np.int32) * np.int32(running + 1)
        running = int(lab.max())  # :138-140 (0 if the crop is empty)
        arr[x : x + cw, y : y + ch, z : z + cd] = lab.astype(np.int16)  # int16 store wraps

    collisions: List[Tuple[int, int]] = []
    for v in seams[0]:
        if v > 0:
            collisions.extend(get_adjacent_labels(arr[v], arr[v - 1]))
    for v in seams[1]:
        if v > 0:
            collisions.extend(get_adjacent_labels(arr[:, v], arr[:, v - 1]))
    for v in seams[2]:
        if v > 0:
            collisions.extend(get_adjacent_labels(arr[:, :, v], arr[:, :, v - 1]))

    graph: Dict[int, List[int]] = {}
    for a, b in collisions:  # :82-91
        graph.setdefault(a, []).append(b)
        graph.setdefault(b, []).append(a)

    lut: Dict[int, int] = {}
    for comp in connected_components(graph):  # :101-105 last id represents the component
        keep = comp[-1]
        for other in comp[:-1]:
            if other not in lut:  # _in_place_replace: first hit wins (:198-203)
                lut[other] = keep
    if lut:
        flat = arr.reshape(-1)
        table = np.arange(-32768, 32768, dtype=np.int16)
        for k, v in lut.items():
            table[k + 32768] = np.int16(v)
        flat[:] = table[flat.astype(np.int32) + 32768]
    return vol.squeeze(0)


def true_ccl_partition(binary: np.ndarray) -> np.ndarray:
    """Plain 6-connected labelling of the whole volume (scipy), for property tests."""
    return ndimage.label(binary > 0)[0]


# --------------------------------------------------------------------------- #
# Renumber  (fastremap.renumber call at skoots/lib/eval.py:304-306)
# --------------------------------------------------------------------------- #


def renumber(labels: np.ndarray) -> Tuple[np.ndarray, Dict[int, int]]:
    """Relabel to 1..K by first appearance in C memory order, 0 preserved.

    PARITY UNPINNED: ``fastremap`` (third-party, Cython, version unpinned in the
    reference's setup.py) is absent from this image.  This follows its published
    behaviour (``renumber(arr, start=1, preserve_zero=True)``: ids handed out in
    order of first appearance while scanning the array in memory order).  The
    reference holds no test or golden output for it.
    """
    flat = np.ascontiguousarray(labels).reshape(-1)
    uniq, first = np.unique(flat, return_index=True)
    keep = uniq != 0
    uniq, first = uniq[keep], first[keep]
    order = np.argsort(first, kind="stable")
    mapping = {int(u): i + 1 for i, u in enumerate(uniq[order])}
    lo = int(flat.min()) if flat.size else 0
    hi = int(flat.max()) if flat.size else 0
    table = np.zeros(hi - lo + 1, dtype=np.int64)
    for k, v in mapping.items():
        table[k - lo] = v
    out = table[flat.astype(np.int64) - lo].reshape(labels.shape)
    mapping[0] = 0
    return out.astype(np.int32), mapping


# --------------------------------------------------------------------------- #
# Driver stages  (skoots/lib/eval.py:126-306) on in-memory tensors
# --------------------------------------------------------------------------- #

TILE = (300, 300, 20)  # eval.py:126
TILE_OVERLAP = (50, 50, 5)  # eval.py:127
ASSIGN_CROP = (500, 500, 50)  # eval.py:248
ASSIGN_OVERLAP = (50, 50, 5)  # eval.py:249
PROB_THR = 0.8  # eval.py:149-150
SKEL_THR = 0.8  # eval.py:176
FOLLOW_N = 10  # eval.py:272


def gate_dilate(out: Tensor) -> Tuple[Tensor, Tensor]:
    """eval.py:145-157 on one tile's network output (1, 5, w, h, d).

    Returns (vec (1,3,w,h,d) in ``out``'s dtype, skeleton_map (1,1,w,h,d) fp32).
    """
    prob = out[:, [-1]]
    skel = out[:, [-2]].float()
    vec = out[:, 0:3]
    vec = vec * prob.gt(PROB_THR)
    skel = skel * prob.gt(PROB_THR)
    skel = binary_dilation(skel)
    skel = binary_dilation_2d(binary_dilation_2d(skel))
    return vec, skel


def scatter_tile(vectors: np.ndarray, skeleton: np.ndarray, vec: Tensor, skel: Tensor,
                 origin, eff, overlap=TILE_OVERLAP) -> None:
    """eval.py:160-176 -- interior crop written into the volume arrays."""
    (x, y, z), (cw, ch, cd), (ox, oy, oz) = origin, eff, overlap
    dst = (slice(x + ox, x + cw - ox), slice(y + oy, y + ch - oy), slice(z + oz, z + cd - oz))
    src = (slice(ox, -ox), slice(oy, -oy), slice(oz, -oz))
    vectors[(slice(None),) + dst] = vec[0][(slice(None),) + src].half().numpy()
    skeleton[(slice(None),) + dst] = skel[0][(slice(None),) + src].gt(SKEL_THR).numpy()


def stage1(image: Tensor, model, mean, std, tile=TILE, overlap=TILE_OVERLAP,
           inject=None, budget_s: float = None, progress: dict = None) -> Tuple[np.ndarray, np.ndarray]:
    """eval.py:126-176.  ``image`` (1, X, Y, Z); ``model`` maps (1,1,w,h,d) fp32 ->
    (1,5,w,h,d).  ``inject(out, origin, eff)`` may replace the network output
    (used to feed synthetic fields to the post-model stages).
    """
    _, X, Y, Z = image.shape
    vectors = np.zeros((3, X, Y, Z), dtype=np.float16)  # eval.py:103
    skeleton = np.zeros((1, X, Y, Z), dtype=np.uint8)  # eval.py:102
    origins, eff = crop_origins((X, Y, Z), list(tile), overlap)
    import time as _time
    t0 = _time.perf_counter()
    for k, (x, y, z) in enumerate(origins):
        if budget_s is not None and k > 0 and _time.perf_counter() - t0 > budget_s:
            break  # bench.py's bounded CPU-baseline sample: the caller extrapolates
        if progress is not None:
            progress.update(done=k + 1, total=len(origins), seconds=_time.perf_counter() - t0)
        crop = image[:, x : x + eff[0], y : y + eff[1], z : z + eff[2]].unsqueeze(0)
        crop = crop.sub(mean).div(std)  # eval.py:139
        out = model(crop.float())
        if inject is not None:
            out = inject(out, (x, y, z), eff)
        vec, skel = gate_dilate(out)
        scatter_tile(vectors, skeleton, vec, skel, (x, y, z), eff, overlap)
        if progress is not None:
            progress.update(done=k + 1, total=len(origins), seconds=_time.perf_counter() - t0)
    return vectors, skeleton


def stage2(skeleton: np.ndarray) -> Tensor:
    """eval.py:223 -- (1,X,Y,Z) u8 -> (X,Y,Z) int16 labels."""
    return efficient_flood_fill(torch.from_numpy(skeleton.copy()).to(torch.int16))


def stage3(vectors: np.ndarray, labels: Tensor, scale, n: int = FOLLOW_N, decay: float = 1.0,
           crop=ASSIGN_CROP, overlap=ASSIGN_OVERLAP) -> Tensor:
    """eval.py:245-284 -- follow + assign per crop, interior written (last writer wins)."""
    _, X, Y, Z = vectors.shape
    inst = torch.zeros((X, Y, Z), dtype=torch.int16)
    lab5 = labels.unsqueeze(0).unsqueeze(0)
    origins, eff = crop_origins((X, Y, Z), list(crop), overlap)
    scale_t = torch.as_tensor(scale)
    for (x, y, z) in origins:
        v = torch.from_numpy(vectors[:, x : x + eff[0], y : y + eff[1], z : z + eff[2]]).unsqueeze(0)
        emb = vector_to_embedding(scale_t, v, N=n, decay=decay)
        emb += torch.tensor((x, y, z)).view(1, 3, 1, 1, 1)  # eval.py:274-276
        got = index_skeleton_by_embed(lab5, emb)[0, 0]
        ox, oy, oz = overlap
        inst[x + ox : x + eff[0] - ox, y + oy : y + eff[1] - oy, z + oz : z + eff[2] - oz] = \
            got[ox:-ox, oy:-oy, oz:-oz]
    return inst


def post_model(vectors: np.ndarray, skeleton: np.ndarray, scale, n: int = FOLLOW_N,
               decay: float = 1.0):
    """Stages 2-3 + renumber on stage-1 outputs.  Returns dict of intermediates."""
    labels = stage2(skeleton)
    inst = stage3(vectors, labels, scale, n=n, decay=decay)
    final, mapping = renumber(inst.numpy())
    return {"labels": labels.numpy(), "instance_raw": inst.numpy(), "instance_mask": final}


def eval_volume(image: Tensor, model, scale, mean=None, std=None, inject=None,
                n: int = FOLLOW_N):
    """Whole pipeline eval.py:126-306 on an in-memory (1,X,Y,Z) volume."""
    img16 = image.to(torch.float16)  # eval.py:80
    mean = img16.mean() if mean is None else mean  # eval.py:87
    std = img16.std() if std is None else std  # eval.py:88
    vectors, skeleton = stage1(img16, model, mean, std, inject=inject)
    res = post_model(vectors, skeleton, scale, n=n)
    res.update(vectors=vectors, skeleton=skeleton)
    return res
