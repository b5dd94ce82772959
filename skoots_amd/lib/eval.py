"""``skoots.lib.eval.eval`` on the MI355X (reference: skoots/lib/eval.py:33-320).

The reference streams 300x300x20 tiles host->GPU->host through zarr arrays and runs
flood fill, offset following and assignment on the CPU.  Here the whole volume stays
resident in HBM (a 2048x2048x512 volume needs ~35 GB of the 288 GB): the image, the
vectors (interleaved fp16x4), the skeleton mask, the labels and the instance mask
never leave the device between stages, tiles are read in place, clamped duplicate
tiles are skipped and stage 3 evaluates every voxel exactly once.

Stage boundaries (and the timed region of ``<image>_skoots_benchmark.txt``) are the
reference's: stage 1 = tiles -> network -> gate/dilate/scatter (eval.py:126-176),
stage 2 = skeleton labelling (:223), stage 3 = follow + assign (:245-284), then
renumber (:304-306).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import time
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _ffi
from ..profile import ASSIGN_BYTES_PER_VOXEL, GATE_BYTES_PER_VOXEL, maybe_span
from . import cropper
from .flood_fill import label_skeleton
from .vector_to_embedding import step_scales

TILE = (300, 300, 20)          # eval.py:126
TILE_OVERLAP = (50, 50, 5)     # eval.py:127
ASSIGN_CROP = (500, 500, 50)   # eval.py:248
ASSIGN_OVERLAP = (50, 50, 5)   # eval.py:249
PROB_THR = 0.8                 # eval.py:149-150
SKEL_THR = 0.8                 # eval.py:176
FOLLOW_N = 10                  # eval.py:272


def thresholds_for(dtype: torch.dtype) -> Tuple[float, float]:
    """``tensor.gt(0.8)`` compares in the tensor's dtype: the python scalar is cast to
    fp16 for an fp16 network output (autocast path, eval.py:142-150) and to fp32
    otherwise; ``skeleton_map`` is always fp32 (eval.py:146)."""
    prob = float(torch.tensor(PROB_THR, dtype=dtype).float())
    skel = float(torch.tensor(SKEL_THR, dtype=torch.float32))
    return prob, skel


class VolumeState:
    """HBM-resident arrays of one volume, or of one rank's z-window of it (slab + halo).

    ``shape`` is the GLOBAL (X, Y, Z); ``window`` = (w_lo, w_hi) are the planes the local
    arrays cover (the whole volume on one GPU)."""

    def __init__(self, shape: Sequence[int], device, window: Optional[Tuple[int, int]] = None,
                 keep_planar_vectors: bool = False):
        X, Y, Z = (int(s) for s in shape)
        self.shape = (X, Y, Z)
        self.window = (0, Z) if window is None else (int(window[0]), int(window[1]))
        zl = self.window[1] - self.window[0]
        self.local_shape = (X, Y, zl)
        self.device = torch.device(device)
        # vectors: interleaved (X,Y,Zl,4) fp16 -- zero = the never-written frame (eval.py:103)
        self.vec4 = torch.zeros((X, Y, zl, 4), dtype=torch.float16, device=device)
        self.skeleton = torch.zeros((X, Y, zl), dtype=torch.uint8, device=device)  # eval.py:102
        self.vec_planar = (torch.zeros((3, X, Y, zl), dtype=torch.float16, device=device)
                           if keep_planar_vectors else None)
        self.labels: Optional[Tensor] = None
        self.instance: Optional[Tensor] = None
        self.profile = None   # optional skoots_amd.profile.KernelProfile (bench.py: per-stage HBM rooflines)

    # -- stage 1 tail ------------------------------------------------------------------
    def scatter_tile(self, out5: Tensor, origin: Sequence[int], overlap=TILE_OVERLAP,
                     owners: Optional[Sequence[np.ndarray]] = None) -> None:
        """One-tile form of :meth:`scatter_tiles`; ``out5`` is (5, w, h, d)."""
        self.scatter_tiles([out5], [origin], overlap, owners)

    def scatter_tiles(self, tiles: Sequence[Tensor], origins: Sequence[Sequence[int]], overlap=TILE_OVERLAP,
                      owners: Optional[Sequence[np.ndarray]] = None) -> None:
        """Gate + dilate + threshold + interior scatter (eval.py:145-176) of a batch of tile outputs in
        one launch.  ``tiles``: (5, w, h, d) fp16/fp32 tensors that are views of ONE storage with
        identical strides and a contiguous z axis (e.g. ``out5[b]`` of a (B,5,w,h,d) batch, or windows
        of a (5,X,Y,Z) array); ``origins`` in GLOBAL coordinates.

        ``owners`` (per-axis ``cropper.owner_table``): restrict each write to the part of the interior
        the tile writes LAST in the reference's order, so that tiles may be scattered in any order
        (or concurrently on several streams) with the same result."""
        t0 = tiles[0]
        _ffi.require_gpu(t0, "out5") if t0.is_contiguous() else None
        if not t0.is_cuda:
            raise RuntimeError("out5 must live on the MI355X; skoots_amd has no CPU fallback")
        assert t0.ndim == 4 and t0.shape[0] == 5 and t0.stride(3) == 1
        _, w, h, d = t0.shape
        X, Y, zl = self.local_shape
        esz = t0.element_size()
        base = min(t.data_ptr() for t in tiles)
        offs, orgs, los, his = [], [], [], []
        for t, origin in zip(tiles, origins):
            assert t.shape == t0.shape and t.stride() == t0.stride() and t.dtype == t0.dtype
            lo = [int(o) for o in overlap]
            hi = [int(s - o) for s, o in zip((w, h, d), overlap)]
            if owners is not None:
                skip = False
                for ax in range(3):
                    own = np.nonzero(owners[ax] == origin[ax])[0]
                    if own.size == 0:
                        skip = True  # every voxel of this tile's interior is overwritten by later tiles
                        break
                    lo[ax], hi[ax] = int(own[0]) - origin[ax], int(own[-1]) + 1 - origin[ax]
                if skip:
                    continue
            offs.append((t.data_ptr() - base) // esz)
            orgs += [int(origin[0]), int(origin[1]), int(origin[2]) - self.window[0]]
            los += lo
            his += hi
        n = len(offs)
        if n == 0:
            return
        pthr, sthr = thresholds_for(t0.dtype)
        for k in range(0, n, 16):
            m = min(16, n - k)
            written = sum((his[3 * j] - los[3 * j]) * (his[3 * j + 1] - los[3 * j + 1]) * (his[3 * j + 2] - los[3 * j + 2])
                          for j in range(k, k + m))
            with maybe_span(self.profile, "gate_dilate_scatter", self.device, GATE_BYTES_PER_VOXEL * written):
                _ffi.check(_ffi.lib.sk_gate_dilate_scatter(
                    C.c_void_p(base), _ffi.dtype_code(t0), m, (C.c_int64 * m)(*offs[k:k + m]),
                    t0.stride(0), t0.stride(1), t0.stride(2), w, h, d,
                    (C.c_int32 * (3 * m))(*orgs[3 * k:3 * (k + m)]), (C.c_int32 * (3 * m))(*los[3 * k:3 * (k + m)]),
                    (C.c_int32 * (3 * m))(*his[3 * k:3 * (k + m)]), _ffi.ptr(self.vec4), _ffi.ptr(self.vec_planar),
                    _ffi.ptr(self.skeleton), X, Y, zl, pthr, sthr, _ffi.stream_ptr(self.device)))
        self._keep = tiles  # the views must outlive the launch

    # -- stage 2 -----------------------------------------------------------------------
    def label(self) -> Tensor:
        assert self.window == (0, self.shape[2]), "label() is the single-GPU path"
        self.labels = label_skeleton(self.skeleton)
        return self.labels

    # -- stage 3 -----------------------------------------------------------------------
    def assign(self, scale, n: int = FOLLOW_N, decay: float = 1.0, crop=ASSIGN_CROP,
               overlap=ASSIGN_OVERLAP, labels: Optional[Tensor] = None,
               z_range: Optional[Tuple[int, int]] = None) -> Tensor:
        """Follow + assign for the planes ``z_range`` (eval.py:245-284) against the FULL
        (X, Y, Z) label volume; returns ``instance`` int32 (X, Y, planes)."""
        labels = self.labels if labels is None else labels
        assert labels is not None, "run label() first"
        X, Y, Z = self.shape
        assert tuple(labels.shape) == (X, Y, Z), "labels must be the full-volume label array"
        eff = cropper.clamp_crop_(list(crop), (X, Y, Z))
        tables = [cropper.owner_table(dm, c, o) for dm, c, o in zip((X, Y, Z), eff, overlap)]
        z_lo, z_hi = (0, Z) if z_range is None else z_range
        oz = tables[2][z_lo:z_hi]
        oz = oz[oz >= 0]
        if oz.size:
            assert self.window[0] <= oz.min() and oz.max() + eff[2] <= self.window[1], (
                "a stage-3 crop reaches outside the rank's window: increase the halo")
        own = [torch.from_numpy(t).to(self.device) for t in tables]
        self.instance = torch.zeros((X, Y, z_hi - z_lo), dtype=torch.int32, device=self.device)
        sc = step_scales(scale, n, decay)
        with maybe_span(self.profile, "follow_assign", self.device, ASSIGN_BYTES_PER_VOXEL * X * Y * (z_hi - z_lo)):
            _ffi.check(_ffi.lib.sk_follow_assign(
                _ffi.ptr(self.vec4), _ffi.ptr(labels), _ffi.dtype_code(labels), _ffi.ptr(self.instance),
                X, Y, Z, self.window[0], self.window[1], _ffi.ptr(own[0]), _ffi.ptr(own[1]), _ffi.ptr(own[2]),
                eff[0], eff[1], eff[2], _ffi.float_array(sc), n, z_lo, z_hi, _ffi.stream_ptr(self.device)))
        self._own = own  # keep the tables alive until the stream has consumed them
        return self.instance

    # -- renumber ----------------------------------------------------------------------
    def renumber(self, max_label: Optional[int] = None) -> int:
        """In-place relabel of ``instance`` to 1..K by first appearance (eval.py:304-306)."""
        inst = self.instance
        assert inst is not None
        if max_label is None:
            max_label = int(inst.max().item())
        n = inst.numel()
        ws_bytes = _ffi.lib.sk_renumber_workspace_bytes(n, max_label)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        k = torch.zeros(1, dtype=torch.int32, device=self.device)
        _ffi.check(_ffi.lib.sk_renumber(_ffi.ptr(inst), n, max_label, _ffi.ptr(ws), ws_bytes,
                                        _ffi.ptr(k), _ffi.stream_ptr(self.device)))
        return int(k.item())

    def vectors_planar(self) -> Tensor:
        """(3, X, Y, Zl) fp16, the layout of the reference's ``vectors`` array."""
        X, Y, Z = self.local_shape
        out = torch.empty((3, X, Y, Z), dtype=torch.float16, device=self.device)
        _ffi.check(_ffi.lib.sk_vec_deinterleave(_ffi.ptr(self.vec4), _ffi.ptr(out), X * Y * Z,
                                                _ffi.stream_ptr(self.device)))
        return out


def tile_grid(shape: Sequence[int], tile=TILE, overlap=TILE_OVERLAP):
    """Distinct stage-1 tile origins in reference order + the effective tile size."""
    eff = list(tile)
    origins = cropper.distinct_origins(shape, eff, overlap)
    return origins, eff


def eval_volume(image: Tensor, model, scale, mean=None, std=None, n: int = FOLLOW_N,
                tile=TILE, tile_overlap=TILE_OVERLAP, tile_batch: Optional[int] = None,
                inject: Optional[Callable] = None, keep_planar_vectors: bool = False,
                timings: Optional[Dict[str, float]] = None) -> Dict[str, Tensor]:
    """In-memory variant of :func:`eval` on one GPU: (1, X, Y, Z) or (X, Y, Z) image on the
    device -> instance mask, vectors, skeleton, labels (all HBM-resident tensors).

    ``model`` is a :class:`skoots_amd.unet.HipUNet`; ``inject(out5, origin, eff) -> out5``
    may replace a tile's network output (parity tests and bench.py's assignment workload
    use it).  The multi-GPU form is :class:`skoots_amd.parallel.ShardedVolume`.
    ``tile_batch``: tiles per network launch; None = as many as the tile count and the free device memory allow,
    at most 64 (:func:`skoots_amd.parallel.pick_tile_batch`; a tile's output does not depend on its batch).
    """
    from ..parallel import ShardedVolume
    img = image.squeeze(0) if image.ndim == 4 else image
    _ffi.require_gpu(img, "image")
    img16 = img if img.dtype == torch.float16 else img.to(torch.float16)  # eval.py:80
    if mean is None or std is None:
        # eval.py:87-88 fallback: statistics of the fp16 image, evaluated by the same torch
        # CPU reduction the reference uses (outside the timed region).
        cpu = img16.cpu()
        mean = float(cpu.mean()) if mean is None else mean
        std = float(cpu.std()) if std is None else std
    sv = ShardedVolume(tuple(img16.shape), 0, 1, img16.device)
    res = sv.run(img16.contiguous(), model, scale, float(mean), float(std), n=n, tile=tile,
                 tile_overlap=tile_overlap, tile_batch=tile_batch, inject=inject,
                 keep_planar_vectors=keep_planar_vectors)
    if timings is not None:
        timings.update(sv.timings)
    return res


# ----------------------------------------------------------------------------------------
# File-level entry point: same signature and side effects as skoots.lib.eval.eval
# ----------------------------------------------------------------------------------------
def _read_image(path: str) -> np.ndarray:
    """[Z, X, Y(, C)] array from a multi-page TIFF (Pillow) or a .npy file (eval.py:61)."""
    if path.endswith(".npy"):
        return np.load(path)
    from PIL import Image
    pages = []
    with Image.open(path) as im:
        for i in range(getattr(im, "n_frames", 1)):
            im.seek(i)
            pages.append(np.array(im))
    return np.stack(pages, axis=0)


def _write_mask_tif(path: str, mask_zxy: np.ndarray) -> None:
    """(Z, X, Y) integer stack -> multi-page TIFF, zlib/deflate compressed (eval.py:309-310)."""
    from PIL import Image
    arr = mask_zxy.astype(np.uint16) if mask_zxy.max() < 65536 else mask_zxy.astype(np.int32)
    pages = [Image.fromarray(p) for p in arr]
    pages[0].save(path, save_all=True, append_images=pages[1:], compression="tiff_adobe_deflate")


def _cfg_get(cfg, section: str, key: str, default=None):
    sec = cfg[section] if isinstance(cfg, dict) else getattr(cfg, section)
    return sec.get(key, default) if isinstance(sec, dict) else getattr(sec, key, default)


@torch.inference_mode()
def eval(image_path: str, checkpoint_path: str, used_cached_data: bool = False, precision: str = "fp16") -> None:
    """Evaluates SKOOTS on an arbitrary image (drop-in for ``skoots.lib.eval.eval``).

    ``precision`` (not in the reference's signature; keyword with the reference's behaviour as default): "fp16" = what the
    reference's fp16 autocast does (eval.py:142); "split" = fp16 hi + lo operand pairs, network outputs within 1e-3 of an
    fp32 forward at about a third of the speed; "mix8" = "split" with the 3x3x3 convs' correction products on the block-scaled fp8
    matrix instruction (same tolerance, ~20 % faster); "fp32" = exact-fp32 matrix instructions (:class:`skoots_amd.unet.HipUNet`).

    Writes next to the image, with the reference's names: ``<base>_skoots_skeleton`` (1,X,Y,Z) u1
    and ``<base>_skoots_vectors`` (3,X,Y,Z) f2 as ``.zarr`` directory stores (zarr v2 layout, uncompressed,
    written by ``zarr_store``: the zarr package itself is not in this image),
    ``<base>_skoots_benchmark.txt`` and ``<base>_instance_mask.tif`` (Z,X,Y), and prints DONE.

    ``checkpoint_path``: ``torch.save``d dict with ``cfg`` (dict/attribute config holding
    ``SKOOTS.VECTOR_SCALING`` and ``MODEL.*``), ``model_state_dict`` and optionally
    ``dataset_mean`` / ``dataset_std`` (eval.py:51-55, 87-88, 99, 117-118).
    """
    from .. import unet
    start = time.time()
    logging.info(f"Loading model file: {checkpoint_path}")
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    if "cfg" not in checkpoint:
        raise RuntimeError("Attempting to evaluate skoots on a legacy model file.")  # eval.py:55
    cfg = checkpoint["cfg"]
    if not torch.cuda.is_available():
        raise RuntimeError("skoots_amd.eval needs an MI355X: there is no CPU path")
    device = torch.device("cuda", torch.cuda.current_device())
    base = os.path.splitext(image_path)[0]

    logging.info(f"Loading image from file: {image_path}")
    image = _read_image(image_path)  # [Z, X, Y(, C)]
    image = image[..., np.newaxis] if image.ndim == 3 else image
    image = image.transpose(-1, 1, 2, 0)
    image = image[[2], ...] if image.shape[0] > 3 else image  # eval.py:64 -> [C=1, X, Y, Z]
    c, x, y, z = image.shape
    logging.info(f"Loaded an image with shape: {(c, x, y, z)}, dtype: {image.dtype}")
    img16 = torch.from_numpy(np.ascontiguousarray(image[0]).astype(np.float32)).to(torch.float16)  # eval.py:80
    mean = checkpoint["dataset_mean"] if "dataset_mean" in checkpoint else img16.mean()  # eval.py:87
    std = checkpoint["dataset_std"] if "dataset_std" in checkpoint else img16.std()  # eval.py:88
    scale = [int(v) for v in _cfg_get(cfg, "SKOOTS", "VECTOR_SCALING")]  # eval.py:99

    logging.info("Constructing SKOOTS model")
    model = unet.cfg_to_model(cfg, device, checkpoint["model_state_dict"], precision=precision)
    from . import zarr_store
    skel_path, vec_path = base + "_skoots_skeleton.zarr", base + "_skoots_vectors.zarr"  # eval.py:102-103

    benchmark_start = time.time()
    dev_img = img16.to(device)
    if used_cached_data and os.path.exists(skel_path) and os.path.exists(vec_path):  # eval.py:105 (os.exists bug fixed)
        from ..parallel import ShardedVolume  # noqa: F401
        state = VolumeState((x, y, z), device)
        state.skeleton.copy_(torch.from_numpy(zarr_store.load(skel_path)[0]).to(device))
        vp = torch.from_numpy(zarr_store.load(vec_path)).to(device)
        _ffi.check(_ffi.lib.sk_vec_interleave(_ffi.ptr(vp), _ffi.ptr(state.vec4), x * y * z,
                                              _ffi.stream_ptr(device)))
        state.label()
        state.assign(scale)
        state.renumber()
        inst, vectors, skeleton = state.instance, vp, state.skeleton
    else:
        res = eval_volume(dev_img, model, scale, mean=float(mean), std=float(std))
        inst, skeleton = res["instance_mask"], res["skeleton"]
        vectors = res["state"].vectors_planar()
    torch.cuda.synchronize(device)
    dt = time.time() - benchmark_start

    zarr_store.save(skel_path, skeleton.cpu().numpy()[np.newaxis])
    zarr_store.save(vec_path, vectors.cpu().numpy())
    logging.info("writing benchmark information")
    with open(base + "_skoots_benchmark.txt", "w") as f:  # eval.py:286-295
        f.write("SKOOTS Segmentation Benchmark:\n")
        f.write("------------------------------\n")
        f.write(f"Time: {dt} seconds\n")
        f.write(f"Memory (current/max): ({torch.cuda.memory_allocated(device)}, "
                f"{torch.cuda.max_memory_allocated(device)})\n\n")
    print("DONE")
    logging.info(f"saving to tif file at {base}_instance_mask.tif")
    _write_mask_tif(base + "_instance_mask.tif", inst.cpu().numpy().transpose(2, 0, 1))
    elapsed = time.time() - start
    logging.info(f"DONE: Process took {elapsed} seconds, {elapsed / 60} minutes, {elapsed / (60 ** 2)}, hours")
