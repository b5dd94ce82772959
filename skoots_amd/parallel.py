"""Z-sharded execution of the eval hot path: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the MI355X node; "gloo" for the CPU / single-GPU
rehearsals in tests/).

The reference is single device (skoots/lib/eval.py:57).  The path shards naturally:

  stage 1  tiles are independent.  Rank r owns the planes [z_lo, z_hi) of the volume; every
           tile of the global 300x300x20 grid (unchanged) runs on exactly one rank: the tiles,
           z-major, are cut into equal runs, so a z position that straddles two slabs is split
           in (x, y) between them; what a rank's tiles write into a neighbour's slab is sent
           point to point, packed (input halo comes with the image).
  stage 2  each rank labels its slab; the label planes either side of every slab
           boundary are exchanged, the (few) seam equivalences are all-gathered and
           every rank applies the same union to its slab.  The slabs are then
           all-gathered into the full label volume: a 10-step follow can end ~165 planes
           away from where it started, more than two 64-plane slabs.
  stage 3  vectors are exchanged with the neighbours (window = slab +- halo) and
           sk_follow_assign runs on the rank's planes against the full label volume.
  renumber first-appearance positions are all-reduced (MIN) so that every rank derives
           the same 1..K numbering.

xGMI is point to point (7 links x ~153 GB/s per GPU): the halo traffic rides one link
per neighbour, the all-gather uses all of them; both are small next to stage 1.
With world == 1 this is exactly the single-GPU pipeline of ``skoots_amd.lib.eval``.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor

from .lib import cropper

HALO = 48  # planes kept either side of the slab: >= 45 (stage-3 crop reach) and the tile depth reach


def slab_bounds(Z: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal plane ranges; rank r owns [lo, hi)."""
    return [(Z * r // world, Z * (r + 1) // world) for r in range(world)]


def window_of(slab: Tuple[int, int], Z: int, world: int, halo: int = HALO) -> Tuple[int, int]:
    if world == 1:
        return (0, Z)
    return (max(0, slab[0] - halo), min(Z, slab[1] + halo))


def _owned_range(own: np.ndarray, origin: int) -> Tuple[int, int]:
    idx = np.nonzero(own == origin)[0]
    return (int(idx[0]), int(idx[-1]) + 1) if idx.size else (0, 0)


def tile_plan(shape: Sequence[int], tile: Sequence[int], overlap: Sequence[int], world: int,
              halo: int = HALO) -> Tuple[List[List[Tuple[int, int, int]]], List[int]]:
    """Which rank evaluates which tile of the reference's global grid (cropper.py:97-144), plus the
    effective tile size.  Every distinct tile runs on exactly ONE rank.  With one rank: all of them in
    the reference's order.  Otherwise the tiles, sorted z-major, are cut into ``world`` equal contiguous
    runs -- a z position of the grid that straddles two slabs is split between the two ranks in (x, y),
    so every rank evaluates T/world tiles (+-1) whatever Z/world is (whole z positions per rank would
    leave 7 : 6 imbalances at 2048x2048x512 on 8 ranks).  What a rank's tiles write into another
    rank's slab travels point to point afterwards (:func:`exchange_blocks`).  Raises ``ValueError`` if a
    run does not fit the rank's window (slab +- halo): slabs too thin for the halo."""
    eff = list(tile)
    origins = cropper.distinct_origins(shape, eff, overlap)
    if world == 1:
        return [origins], eff
    slabs = slab_bounds(shape[2], world)
    windows = [window_of(s, shape[2], world, halo) for s in slabs]
    order = sorted(origins, key=lambda o: (o[2], o[0], o[1]))
    T = len(order)
    plan = [order[T * r // world:T * (r + 1) // world] for r in range(world)]
    for r, (p_, w) in enumerate(zip(plan, windows)):
        for o in p_:
            if not (w[0] <= o[2] and o[2] + eff[2] <= w[1]):
                raise ValueError(f"tile at z={o[2]} does not fit rank {r}'s window {w} (slab +- {halo} planes): "
                                 f"slabs of {shape[2]}/{world} planes are too thin for this halo")
    return plan, eff


def block_plan(shape: Sequence[int], eff: Sequence[int], overlap: Sequence[int],
               plan: List[List[Tuple[int, int, int]]]) -> List[Tuple[int, int, Tuple[int, ...]]]:
    """(src, dst, (x0, x1, y0, y1, z0, z1)) for every box of voxels that a tile evaluated on rank
    ``src`` writes (as their last writer, eval.py:160-176) into the slab of another rank ``dst``;
    global coordinates, deterministic order (the same list on every rank)."""
    world = len(plan)
    slabs = slab_bounds(shape[2], world)
    own = [cropper.owner_table(dm, c, o) for dm, c, o in zip(shape, eff, overlap)]
    rng = [{int(o): _owned_range(t, int(o)) for o in np.unique(t) if o >= 0} for t in own]
    out = []
    for src, tiles in enumerate(plan):
        for (ox, oy, oz) in tiles:
            (x0, x1), (y0, y1), (z0, z1) = rng[0].get(ox, (0, 0)), rng[1].get(oy, (0, 0)), rng[2].get(oz, (0, 0))
            if x0 >= x1 or y0 >= y1:
                continue
            for dst, (lo, hi) in enumerate(slabs):
                a, b = max(z0, lo), min(z1, hi)
                if dst != src and a < b:
                    out.append((src, dst, (x0, x1, y0, y1, a, b)))
    return out


def halo_plan(slabs: List[Tuple[int, int]], windows: List[Tuple[int, int]], rank: int):
    """(sends, recvs): lists of (peer, z_lo, z_hi) plane ranges (global z).  Rank r sends
    the part of its slab that falls into a peer's window and receives the parts of its
    own window owned by peers."""
    sends, recvs = [], []
    my_slab, my_win = slabs[rank], windows[rank]
    for q in range(len(slabs)):
        if q == rank:
            continue
        lo, hi = max(my_slab[0], windows[q][0]), min(my_slab[1], windows[q][1])
        if lo < hi:
            sends.append((q, lo, hi))
        lo, hi = max(slabs[q][0], my_win[0]), min(slabs[q][1], my_win[1])
        if lo < hi:
            recvs.append((q, lo, hi))
    return sends, recvs


class Comm:
    """Thin wrapper: RCCL moves device tensors directly; gloo stages through the host.

    Every call is accounted under a ``what`` tag (bytes this rank SENDS, time on the calling stream:
    HIP events for RCCL, wall clock for the host-staged gloo rehearsal) -- ``stats()`` is what
    bench.py prints for N > 1, so that a scaling curve can be read per exchange step."""

    def __init__(self, rank: int, world: int, force_device: bool = False):
        """``force_device``: run the collectives through ``torch.distributed`` even with ONE rank (they are identities
        there and normally skipped) -- on a one-GPU box this is how the RCCL code path gets executed at all
        (tests/test_hip_sharded.py: library load, ``device_id`` binding, ``all_gather_into_tensor``, all-reduce)."""
        self.rank, self.world = rank, world
        self.force = bool(force_device) and dist.is_available() and dist.is_initialized()
        self.staged = (world > 1 or self.force) and dist.get_backend() == "gloo"
        self._acct: Dict[str, Dict[str, object]] = {}

    def _out(self, t: Tensor) -> Tensor:
        return t.cpu() if (self.staged and t.is_cuda) else t

    # -- accounting ----------------------------------------------------------------------
    class _Span:
        """Brackets ONE SYNCHRONOUS collective.  RCCL runs it on PyTorch's internal communication stream; the event
        pair on the calling stream brackets it only because a synchronous op makes the calling stream wait for it.
        With ``async_op=True`` the pair would time nothing: every collective in this module is synchronous, and
        :meth:`Comm.exchange` waits for its requests inside the span."""

        def __init__(self, comm: "Comm", what: str, nbytes: int, device):
            self.c, self.what, self.nbytes, self.device = comm, what, int(nbytes), device

        def __enter__(self):
            self.ev = None
            if (not self.c.staged) and self.device is not None and self.device.type == "cuda":
                self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self.ev[0].record(torch.cuda.current_stream(self.device))
            self.t0 = time.perf_counter()
            return self

        def __exit__(self, *exc):
            a = self.c._acct.setdefault(self.what, {"bytes": 0, "calls": 0, "wall_s": 0.0, "events": []})
            a["bytes"] += self.nbytes
            a["calls"] += 1
            if self.ev is not None:
                self.ev[1].record(torch.cuda.current_stream(self.device))
                a["events"].append(self.ev)
            else:
                a["wall_s"] += time.perf_counter() - self.t0
            return False

    def span(self, what: str, nbytes: int, device=None) -> "_Span":
        return Comm._Span(self, what, nbytes, device)

    def stats(self) -> Dict[str, Dict[str, float]]:
        """{tag: {"bytes": sent by this rank, "ms": time, "calls": n}} accumulated since construction."""
        out = {}
        for k, a in self._acct.items():
            ms = a["wall_s"] * 1e3
            for e0, e1 in a["events"]:
                e1.synchronize()
                ms += e0.elapsed_time(e1)
            out[k] = {"bytes": int(a["bytes"]), "ms": round(ms, 3), "calls": int(a["calls"])}
        return out

    # -- collectives ---------------------------------------------------------------------
    def all_gather(self, t: Tensor, what: str = "all_gather") -> List[Tensor]:
        if self.world == 1 and not self.force:
            return [t]
        src = self._out(t.contiguous())
        with self.span(what, src.numel() * src.element_size() * (self.world - 1), t.device):
            if not self.staged:  # RCCL: one flat output buffer, no per-rank list copies
                flat = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
                dist.all_gather_into_tensor(flat, src)
                return list(flat.unbind(0))
            outs = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(outs, src)
            return [o.to(t.device) for o in outs]

    def all_reduce_min(self, t: Tensor, what: str = "all_reduce") -> Tensor:
        if self.world == 1 and not self.force:
            return t
        src = self._out(t)
        with self.span(what, 2 * src.numel() * src.element_size() * (self.world - 1) // self.world, t.device):
            dist.all_reduce(src, op=dist.ReduceOp.MIN)
        return src.to(t.device)

    def exchange(self, sends: List[Tuple[int, Tensor]], recv_like: List[Tuple[int, Tensor]],
                 what: str = "p2p") -> List[Tensor]:
        """Point-to-point batch: sends [(peer, tensor)], recv_like [(peer, empty tensor)]."""
        if self.world == 1:
            return []
        dev = sends[0][1].device if sends else (recv_like[0][1].device if recv_like else None)
        with self.span(what, sum(t.numel() * t.element_size() for _, t in sends), dev):
            return self._exchange(sends, recv_like)

    def _exchange(self, sends, recv_like) -> List[Tensor]:
        ops, staged_recv = [], []
        for peer, t in sends:
            ops.append(dist.P2POp(dist.isend, self._out(t.contiguous()), peer))
        for peer, t in recv_like:
            buf = self._out(t) if not self.staged else torch.empty(t.shape, dtype=t.dtype)
            staged_recv.append(buf)
            ops.append(dist.P2POp(dist.irecv, buf, peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return [b.to(t.device) for b, (_, t) in zip(staged_recv, recv_like)]


def exchange_halo(arr: Tensor, slabs, windows, rank: int, comm: Comm, needs=None, what: str = "halo") -> None:
    """Fill the halo planes of a window-shaped array (X, Y, Zl, ...) from their owners.  ``needs``:
    per-rank plane ranges (inside the windows) that are actually read afterwards; default: the windows."""
    sends, recvs = halo_plan(slabs, windows if needs is None else needs, rank)
    w0 = windows[rank][0]
    out = [(q, arr[:, :, lo - w0:hi - w0]) for q, lo, hi in sends]
    like = [(q, torch.empty((arr.shape[0], arr.shape[1], hi - lo) + tuple(arr.shape[3:]), dtype=arr.dtype,
                            device=arr.device)) for q, lo, hi in recvs]
    got = comm.exchange(out, like, what=what)
    for (q, lo, hi), t in zip(recvs, got):
        arr[:, :, lo - w0:hi - w0] = t


def assign_reach(shape: Sequence[int], crop: Sequence[int], overlap: Sequence[int], slabs, windows):
    """Per rank: the plane range stage 3 reads vectors from -- the union of the 500x500x50 crops
    (eval.py:248-284) that write the rank's planes last, clipped to its window.  Less than the
    full halo on most ranks (2048x2048x512 / 8: 24-42 planes a side instead of 48)."""
    eff = cropper.clamp_crop_(list(crop), shape)
    own_z = cropper.owner_table(shape[2], eff[2], overlap[2])
    out = []
    for (lo, hi), (wl, wh) in zip(slabs, windows):
        oz = own_z[lo:hi]
        oz = oz[oz >= 0]
        a, b = (int(oz.min()), int(oz.max()) + eff[2]) if oz.size else (lo, hi)
        out.append((max(wl, min(a, lo)), min(wh, max(b, hi))))
    return out


def exchange_blocks(arrays: Sequence[Tensor], blocks, windows, rank: int, comm: "Comm") -> None:
    """Deliver what this rank's tiles wrote into other ranks' slabs (:func:`block_plan`) and take in
    what theirs wrote into this one.  ``arrays``: window-shaped (X, Y, Zl, ...) tensors, updated in
    place; one packed message per (array, peer)."""
    w0 = windows[rank][0]

    def view(arr, box):
        x0, x1, y0, y1, z0, z1 = box
        return arr[x0:x1, y0:y1, z0 - w0:z1 - w0]

    out_by, in_by = {}, {}
    for src, dst, box in blocks:
        if src == rank:
            out_by.setdefault(dst, []).append(box)
        elif dst == rank:
            in_by.setdefault(src, []).append(box)
    sends, like, dst_views = [], [], []
    for arr in arrays:
        tail = int(np.prod(arr.shape[3:])) if arr.ndim > 3 else 1
        for peer in sorted(out_by):
            sends.append((peer, torch.cat([view(arr, b).reshape(-1) for b in out_by[peer]])))
        for peer in sorted(in_by):
            n = sum((b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]) for b in in_by[peer]) * tail
            like.append((peer, torch.empty(n, dtype=arr.dtype, device=arr.device)))
            dst_views.append([view(arr, b) for b in in_by[peer]])
    got = comm.exchange(sends, like, what="block_exchange")
    for views, flat in zip(dst_views, got):
        at = 0
        for v in views:
            v.copy_(flat[at:at + v.numel()].view(v.shape))
            at += v.numel()


def comm_budget(shape: Sequence[int], world: int, tile=(300, 300, 20), tile_overlap=(50, 50, 5), halo: int = HALO,
                pair_cap: Optional[int] = None, nnz_div: Optional[int] = None) -> List[Dict[str, object]]:
    """What every rank of a ``world``-rank run SENDS per step, per exchange tag, computed from the plans alone (no device, no
    process group) -- the numbers ``Comm.stats()`` / bench.py's ``multi_gpu.comm_rank0_per_step`` of a real run are read
    against.  Per rank: ``tiles``; ``block_exchange`` (bytes: what its tiles write into other ranks' slabs -- 8 B of
    interleaved vector + 1 B of skeleton per voxel; ``peers``); ``label_seam_planes`` (one int32 plane to the rank below);
    ``label_meta`` and ``label_gather`` (the two fixed-size all-gathers of :func:`lib.flood_fill.label_slab`: bytes x
    (world - 1) receivers); ``vector_halo`` (bytes, ``planes`` per neighbour: the planes its neighbours' stage-3 crops
    reach, :func:`assign_reach`).  ``renumber_allreduce`` depends on the label count and is not planned."""
    from .lib import flood_fill
    from .lib.eval import ASSIGN_CROP, ASSIGN_OVERLAP
    X, Y, Z = (int(v) for v in shape)
    plan, eff = tile_plan(shape, tile, tile_overlap, world, halo=halo)
    slabs = slab_bounds(Z, world)
    windows = [window_of(s_, Z, world, halo) for s_ in slabs]
    out = [{"tiles": len(p), "block_exchange": 0, "block_peers": set(), "label_seam_planes": 0, "label_meta": 0,
            "label_gather": 0, "vector_halo": 0, "vector_halo_planes": {}} for p in plan]
    if world == 1:
        for o in out:
            o["block_peers"] = []
        return out
    for src, dst, (x0, x1, y0, y1, z0, z1) in block_plan(shape, eff, tile_overlap, plan):
        out[src]["block_exchange"] += (x1 - x0) * (y1 - y0) * (z1 - z0) * (8 + 1)
        out[src]["block_peers"].add(dst)
    pair_cap = flood_fill.PAIR_CAP if pair_cap is None else pair_cap
    nnz_div = flood_fill.NNZ_DIV if nnz_div is None else nnz_div
    zmax = max(b - a for a, b in slabs)
    nnz_cap = max(1 << 16, (X * Y * zmax) // nnz_div)
    needs = assign_reach(shape, ASSIGN_CROP, ASSIGN_OVERLAP, slabs, windows)
    for r, o in enumerate(out):
        o["block_peers"] = sorted(o["block_peers"])
        o["label_seam_planes"] = X * Y * 4 if r > 0 else 0
        o["label_meta"] = (4 + 2 * pair_cap) * 4 * (world - 1)
        o["label_gather"] = nnz_cap * 8 * (world - 1)
        sends, _ = halo_plan(slabs, needs, r)
        for q, lo, hi in sends:
            o["vector_halo"] += X * Y * (hi - lo) * 8
            o["vector_halo_planes"][q] = hi - lo
    return out


MAX_TILE_BATCH = 64   # tiles per network launch the kernels' launch plans are built for


def tile_batch_bytes(eff: Sequence[int], split: bool = False) -> int:
    """Activation bytes ONE tile adds to a network context (skoots_amd.unet.HipUNet buffers): three 32-channel
    full-resolution tensors, the half- and quarter-resolution ones, the stem workspace and the 5-channel output."""
    v = int(eff[0]) * int(eff[1]) * int(eff[2])
    lanes = 2 if split else 1
    l0 = 3 * 32 * 2 * v * lanes                                  # L0a, L0b, skip0
    l1 = (3 * 64 + 32) * 2 * (v // 8) * lanes                    # L1a, L1b, skip1, L1r
    l2 = (2 * 128 + 64) * 2 * (v // 64) * lanes                  # L2a, L2b, L2r
    return l0 + l1 + l2 + 5 * 2 * v + 4 * v                      # + out5 + the stem's normalised workspace


def pick_tile_batch(n_tiles: int, eff: Sequence[int], device, contexts: int = 1, split: bool = False,
                    requested: Optional[int] = None) -> int:
    """Tiles per network launch: ``requested`` (default MAX_TILE_BATCH) clamped to the tile count and to what fits in
    60 % of the device memory that is free right now (``contexts`` activation sets: one per stream).  A tile's output
    bits do not depend on its batch (tests/test_hip_geometry.py), so the clamp changes speed only; a device too full
    for even one tile raises here, with the numbers, instead of failing inside a layer's allocation."""
    want = MAX_TILE_BATCH if requested is None else int(requested)
    if want < 1:
        raise ValueError("tile_batch must be >= 1")
    want = min(want, MAX_TILE_BATCH, max(1, int(n_tiles)))
    dev = torch.device(device)
    if dev.type != "cuda":
        return want
    free, _ = torch.cuda.mem_get_info(dev)
    free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)   # torch's cached blocks are reusable
    per = tile_batch_bytes(eff, split) * max(1, contexts)
    fit = int(0.6 * free // per)
    if fit < 1:
        raise RuntimeError(f"not enough free device memory for one tile of {tuple(eff)}: {free / 2**30:.1f} GiB free, "
                           f"{per / 2**30:.2f} GiB of activations per tile")
    return min(want, fit)


class ShardedVolume:
    """One rank's share of a volume: runs stages 1-3 + renumber, collectives included."""

    def __init__(self, shape: Sequence[int], rank: int, world: int, device, halo: int = HALO,
                 force_distributed: bool = False):
        """``force_distributed``: take the multi-rank code paths (slab labelling with its metadata / label gathers,
        distributed renumber) even with one rank, with the collectives really issued (``Comm(force_device=True)``):
        the single-rank result must equal the plain pipeline's -- the one-GPU box's rehearsal of the RCCL path."""
        self.force_distributed = bool(force_distributed)
        self.shape = tuple(int(v) for v in shape)
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        self.halo = halo
        self.sparse_labels = None  # label all-gather as sparse lists: None = automatic (tests force either)
        self.slabs = slab_bounds(self.shape[2], world)
        self.windows = [window_of(s, self.shape[2], world, halo) for s in self.slabs]
        self.slab, self.window = self.slabs[rank], self.windows[rank]
        self.timings: Dict[str, float] = {}
        self.comm = Comm(rank, world, force_device=self.force_distributed)   # persistent: its accounting accumulates over run() calls
        self.tiles_this_rank = 0

    def _side_stream(self, i: int):
        if not hasattr(self, "_streams"):
            self._streams = {}
        if i not in self._streams:
            self._streams[i] = torch.cuda.Stream(self.device)
        return self._streams[i]

    def _tick(self, name: str, t0: float) -> None:
        torch.cuda.synchronize(self.device)
        self.timings[name] = self.timings.get(name, 0.0) + time.perf_counter() - t0

    def run(self, image: Tensor, model, scale, mean: float, std: float, n: int = 10,
            decay: float = 1.0, tile=(300, 300, 20), tile_overlap=(50, 50, 5), tile_batch: Optional[int] = None,
            inject: Optional[Callable] = None, keep_planar_vectors: bool = False,
            conv_profile=None, streams: int = 1, stage_profile=None) -> Dict[str, object]:
        """``image``: this rank's window of the fp16 volume, shape (X, Y, window planes)."""
        from . import _ffi
        from .lib.eval import ASSIGN_CROP, ASSIGN_OVERLAP, VolumeState
        from .lib.flood_fill import label_skeleton, label_slab

        X, Y, Z = self.shape
        (zlo, zhi), (wlo, whi) = self.slab, self.window
        dev = self.device
        comm = self.comm
        assert tuple(image.shape) == (X, Y, whi - wlo), (tuple(image.shape), (X, Y, whi - wlo))
        state = VolumeState(self.shape, dev, window=self.window, keep_planar_vectors=keep_planar_vectors)
        state.profile = stage_profile

        # ---- stage 1 --------------------------------------------------------------------
        t0 = time.perf_counter()
        plan, eff = tile_plan(self.shape, tile, tile_overlap, self.world, halo=self.halo)
        origins = plan[self.rank]
        self.tiles_this_rank = len(origins)
        for (_, _, oz) in origins:
            assert wlo <= oz and oz + eff[2] <= whi, "tile outside the rank's window: increase the halo"
        owners = [cropper.owner_table(dm, c, o) for dm, c, o in zip(self.shape, eff, tile_overlap)]
        # the gate/dilate/scatter kernel reads a tile's output on its interior +- the dilation reach (3,3,1)
        reach = (3, 3, 1)
        out_box = ([max(0, o - r) for o, r in zip(tile_overlap, reach)],
                   [min(s_, s_ - o + r) for s_, o, r in zip(eff, tile_overlap, reach)])
        n_streams = max(1, int(streams)) if model is not None else 1
        if model is not None:   # None / too large a request: clamped to the tile count and the free device memory
            tile_batch = pick_tile_batch(len(origins), eff, dev, contexts=n_streams,
                                         split=getattr(model, "precision", "fp16") in ("split", "mix8"), requested=tile_batch)
        elif tile_batch is None:
            tile_batch = MAX_TILE_BATCH
        self.tile_batch_used = tile_batch
        if model is not None:
            # The extra contexts (activation buffers of the second, third ... stream) live ON THE MODEL across calls: they
            # hold its layer objects, so they die with it (a cache keyed by id(model) here could hand a later model with a
            # recycled id the old model's weights) and their buffers are freed when fewer streams are asked for.
            extra = model.__dict__.setdefault("_stream_ctxs", [])
            del extra[max(0, n_streams - 1):]
            while len(extra) < n_streams - 1:
                extra.append(model.clone_context())
            for c in extra:   # same weights object; follow the switches of the primary
                assert c.enc0 is model.enc0 and c.dec0 is model.dec0, "a stream context must share its model's layers"
                c.precision, c.fold_upsample, c.box_store, c.defer_activation = (model.precision, model.fold_upsample,
                                                                                model.box_store, model.defer_activation)
            ctxs = [model] + extra
        else:
            ctxs = [None]
        for c in ctxs:
            if c is not None:
                c.profile = conv_profile
        main = torch.cuda.current_stream(dev)
        lanes = [main] + [self._side_stream(i) for i in range(n_streams - 1)]
        for s_ in lanes[1:]:
            s_.wait_stream(main)
        for bi, i in enumerate(range(0, len(origins), tile_batch)):
            batch = origins[i:i + tile_batch]
            local = [(x, y, z - wlo) for (x, y, z) in batch]
            k = bi % n_streams
            with torch.cuda.stream(lanes[k]):
                out5 = (ctxs[k].forward_tiles(image, local, eff, mean, std, out_box=out_box)
                        if model is not None else None)
                outs = []
                for b, org in enumerate(batch):
                    o = out5[b] if out5 is not None else None
                    if inject is not None:
                        o = inject(o, (org[0], org[1], org[2] - wlo), eff)
                    outs.append(o)
                same = all(o.stride() == outs[0].stride() and o.dtype == outs[0].dtype and
                           o.untyped_storage().data_ptr() == outs[0].untyped_storage().data_ptr() for o in outs)
                if same and outs[0].stride(3) == 1:
                    state.scatter_tiles(outs, batch, tile_overlap, owners=owners)  # one launch per batch
                else:
                    for o, org in zip(outs, batch):
                        state.scatter_tile(o.contiguous(), org, tile_overlap, owners=owners)
        for s_ in lanes[1:]:
            main.wait_stream(s_)
        for c in ctxs:
            if c is not None:
                c.profile = None
        if self.world > 1:  # what this rank's tiles wrote into other ranks' slabs goes to them
            blocks = block_plan(self.shape, eff, tile_overlap, plan)
            exchange_blocks([state.vec4, state.skeleton], blocks, self.windows, self.rank, comm)
        self._tick("stage1", t0)

        # ---- stage 2 --------------------------------------------------------------------
        t0 = time.perf_counter()
        if self.world == 1 and not self.force_distributed:
            labels = label_skeleton(state.skeleton, reference_ids=False, profile=stage_profile)
            n_labels_hint = None
        else:
            labels, n_labels_hint, overflow = label_slab(state.skeleton, self.shape, self.slab, self.window,
                                                         self.slabs, self.rank, comm, sparse=self.sparse_labels,
                                                         profile=stage_profile)
        state.labels = labels
        self._tick("stage2", t0)
        if self.world > 1 or self.force_distributed:
            # label_slab ran without reading anything back; the stage's timing synchronisation above has happened, so
            # the overflow flag costs one tiny copy.  It is the same on every rank (all-gathered metadata).
            if bool(overflow.item()):
                from .lib.flood_fill import _label_slab_sync
                t0 = time.perf_counter()
                labels, n_labels_hint = _label_slab_sync(state.skeleton, self.shape, self.slab, self.window, self.slabs,
                                                         self.rank, comm, sparse=self.sparse_labels, profile=stage_profile)
                state.labels = labels
                self._tick("stage2", t0)
            n_labels_hint = int(n_labels_hint)

        # ---- stage 3 --------------------------------------------------------------------
        t0 = time.perf_counter()
        if self.world > 1:
            exchange_halo(state.vec4, self.slabs, self.windows, self.rank, comm,
                          needs=assign_reach(self.shape, ASSIGN_CROP, ASSIGN_OVERLAP, self.slabs, self.windows),
                          what="vector_halo")
        inst = state.assign(scale, n=n, decay=decay, crop=ASSIGN_CROP, overlap=ASSIGN_OVERLAP,
                            labels=labels, z_range=self.slab)
        self._tick("stage3", t0)

        # ---- renumber -------------------------------------------------------------------
        t0 = time.perf_counter()
        if self.world == 1 and not self.force_distributed:
            k = state.renumber()
        else:
            k = distributed_renumber(inst, self.shape, self.slab, int(n_labels_hint), comm)
        self._tick("renumber", t0)
        return {"instance_mask": inst, "labels": labels, "skeleton": state.skeleton, "vec4": state.vec4,
                "vectors": state.vec_planar, "n_instances": k, "state": state, "slab": self.slab,
                "window": self.window}


def distributed_renumber(inst: Tensor, shape, slab, max_label: int, comm: Comm) -> int:
    """fastremap.renumber semantics (eval.py:304-306) across ranks: ids 1..K by first
    appearance in the GLOBAL C order.  ``inst`` is this rank's (X, Y, slab planes) int32."""
    from . import _ffi
    X, Y, Z = shape
    dev = inst.device
    first = torch.full((max_label + 1,), -1, dtype=torch.int32, device=dev)  # 0xFFFFFFFF
    _ffi.check(_ffi.lib.sk_first_seen(_ffi.ptr(inst), X, Y, slab[1] - slab[0], slab[0], Z, max_label,
                                      _ffi.ptr(first), _ffi.stream_ptr(dev)))
    f64 = first.to(torch.int64) & 0xFFFFFFFF
    f64 = comm.all_reduce_min(f64, what="renumber_allreduce")
    seen = torch.nonzero(f64 != 0xFFFFFFFF).flatten()
    order = torch.argsort(f64[seen])
    lut = torch.zeros(max_label + 1, dtype=torch.int32, device=dev)
    lut[seen[order]] = torch.arange(1, seen.numel() + 1, dtype=torch.int32, device=dev)
    _ffi.check(_ffi.lib.sk_relabel_lut(_ffi.ptr(inst), inst.numel(), _ffi.ptr(lut), max_label + 1,
                                       _ffi.stream_ptr(dev)))
    torch.cuda.current_stream(dev).synchronize()
    return int(seen.numel())
