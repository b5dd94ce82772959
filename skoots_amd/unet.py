"""HIP runner of the build's U-Net (graph: oracle/unet_spec.py ``UNetSpec``).

Stands where the reference calls ``cfg_to_bism_model(cfg)`` + ``torch.compile`` and
runs the model on each tile under fp16 autocast (skoots/lib/utils.py:17-107,
skoots/lib/eval.py:117-124,142-143).  PyTorch only owns the device memory and the
stream; every layer is a kernel of libskoots_hip.so:

  stem (VALU, Cin=1)  ->  [conv3 MFMA -> GroupNorm finalize -> fused GN+SiLU] x N
  2x2x2 stride-2 / 1x1x1 convs through the gather GEMM, heads (tanh / sigmoid).

Tiles are read in place from the HBM-resident fp16 volume (no crop copies); a batch of
B tiles runs per launch so that every launch has >> 256 workgroups.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _ffi

GN_GROUPS = 8
GN_EPS = 1e-5
PRECISIONS = ("fp16", "split", "mix8", "fp32")


class _ConvLayer:
    """One Conv3d -> GroupNorm -> SiLU block with weights packed for the kernels."""

    def __init__(self, prefix: str, sd: Dict[str, Tensor], device, ksize: int):
        w = sd[prefix + ".conv.weight"].detach().float().cpu().contiguous()
        self.weight_f32 = w.to(device)  # torch layout, used by the fp32 precision mode
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        assert tuple(w.shape[2:]) == (ksize,) * 3, (prefix, tuple(w.shape))
        self.ksize = ksize
        self.name = prefix
        self.bias = sd[prefix + ".conv.bias"].detach().float().to(device).contiguous()
        self.gamma = sd[prefix + ".norm.weight"].detach().float().to(device).contiguous()
        self.beta = sd[prefix + ".norm.bias"].detach().float().to(device).contiguous()
        self._w_cpu, self._device, self._packed = w, device, {}
        if self.cin == 1:  # stem: (27, cout) fp32, tap-major (the kernel splits it into hi + lo itself)
            assert ksize == 3
            self._packed[False] = self._packed[True] = w.reshape(self.cout, 27).t().contiguous().to(device)
        self.flops_per_out_voxel = 2.0 * self.cin * self.cout * ksize ** 3

    def packed(self, split: bool = False) -> Tensor:
        """MFMA A-fragment image of the weight on the device: fp16 (``sk_conv3d``) or hi + lo split
        (``sk_conv3d_split``); packed on first use."""
        t = self._packed.get(split)
        if t is None:
            fn = _ffi.lib.sk_conv3d_pack_weight_split_host if split else _ffi.lib.sk_conv3d_pack_weight_host
            fpt = self._w_cpu.numpy().ctypes.data_as(C.POINTER(C.c_float))
            nbytes = fn(fpt, self.cout, self.cin, self.ksize, None)
            if nbytes < 0:
                _ffi.check(int(nbytes))
            buf = np.empty(nbytes, dtype=np.uint8)
            fn(fpt, self.cout, self.cin, self.ksize, buf.ctypes.data_as(C.c_void_p))
            t = self._packed[split] = torch.from_numpy(buf).to(self._device)
        return t

    @property
    def weight(self) -> Tensor:
        return self.packed(False)

    def packed_mix8(self) -> Tuple[Tensor, int]:
        """Weight image of ``sk_conv3d_mix8`` (fp16 fragments of w_hi + block-scaled fp8 fragments of w_lo and w) and its
        fp8 scale exponent; packed on first use."""
        t = self._packed.get("mix8")
        if t is None:
            t = self._packed["mix8"] = pack_conv_weight_mix8(self._w_cpu, self._device)
        return t

    def packed_upfold(self, c_skip: int, split: bool = False) -> Tensor:
        """Fragments of ``sk_conv3d_upfold`` / ``sk_conv3d_upfold_split`` (decoder conv, the nearest-upsample of the last
        ``cin - c_skip`` input channels folded into their weights); packed on first use."""
        key = ("upfold", c_skip, split)
        t = self._packed.get(key)
        if t is None:
            fn = _ffi.lib.sk_conv3d_pack_weight_upfold_split_host if split else _ffi.lib.sk_conv3d_pack_weight_upfold_host
            fpt = self._w_cpu.numpy().ctypes.data_as(C.POINTER(C.c_float))
            nbytes = fn(fpt, self.cout, c_skip, self.cin - c_skip, None)
            if nbytes < 0:
                _ffi.check(int(nbytes))
            buf = np.empty(nbytes, dtype=np.uint8)
            fn(fpt, self.cout, c_skip, self.cin - c_skip, buf.ctypes.data_as(C.c_void_p))
            t = self._packed[key] = torch.from_numpy(buf).to(self._device)
        return t


    def packed_upfold_mix8(self, c_skip: int) -> Tuple[Tensor, int]:
        """Weight image of ``sk_conv3d_upfold_mix8`` and its fp8 scale exponent; packed on first use."""
        key = ("upfold_mix8", c_skip)
        t = self._packed.get(key)
        if t is None:
            t = self._packed[key] = pack_conv_weight_upfold_mix8(self._w_cpu, c_skip, self._device)
        return t


class ConvProfile:
    """HIP-event timing of every 3x3x3 MFMA conv launch (the dominant kernel), recorded on
    the stream the kernels are launched on; bench.py turns it into the roofline figure."""

    def __init__(self):
        self.events = []  # (start, end, flops, layer name): flops = the ALGORITHMIC count 2*Cin*Cout*27 per output voxel
        self.executed_flops = 0.0  # what the launches put on the matrix pipe (the folded decoder convs: 8 of 27 taps
        #                            for the upsampled channels)

    def named(self):
        if self.events:
            self.events[-1][1].synchronize()
        return list(self.events)

    def totals(self):
        if not self.events:
            return 0.0, 0.0, 0
        self.events[-1][1].synchronize()
        ms = sum(e[0].elapsed_time(e[1]) for e in self.events)
        return ms, sum(e[2] for e in self.events), len(self.events)


class HipUNet:
    """``model.forward_tiles(image, origins, tile, mean, std) -> (B, 5, w, h, d)`` fp16."""

    def __init__(self, state_dict: Dict[str, Tensor], device="cuda:0",
                 dims: Sequence[int] = (32, 64, 128, 64, 32),
                 depths: Sequence[int] = (2, 2, 2, 2, 2), precision: str = "fp16"):
        """``precision``: "fp16" (fast path: fp16 MFMA operands, fp32 accumulation -- what the reference's fp16
        autocast does, eval.py:142; max-abs ~5e-3 against an fp32 forward), "split" (activations and weights as
        fp16 hi + lo pairs, three fp16 MFMAs per product: max-abs <= 1e-3 against fp32, BASELINE.json's tolerance,
        at ~1/3 of the fast path's speed), "mix8" ("split" whose 32 -> 32 3x3x3 convs -- enc0.1.., dec0.1.. -- take their two
        correction products w_lo x and w x_lo as one block-scaled fp8 matrix product, sk_conv3d_mix8: the corrections are
        2^-11 of the result, so e4m3's 2^-4 keeps them to ~2^-15; every other layer as "split") or "fp32" (every layer on
        the exact-fp32 matrix instruction; the parity reference of the others, ~1/11 of the fast path's speed)."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        self.precision = precision
        self.device = torch.device(device)
        self.dims, self.depths = tuple(dims), tuple(depths)
        d0, d1, d2, d3, d4 = self.dims
        if not (d0 == d4 == 32 and d1 == d3 and d1 in (32, 64, 128) and d2 in (32, 64, 128)):
            raise ValueError(f"unsupported dims {dims}: kernels are built for widths 32/64/128")
        sd = state_dict
        dev = self.device

        def stack(name, n):
            return [_ConvLayer(f"{name}.{i}", sd, dev, 3) for i in range(n)]

        self.enc0 = stack("enc0", depths[0])
        self.down0 = _ConvLayer("down0", sd, dev, 2)
        self.enc1 = stack("enc1", depths[1])
        self.down1 = _ConvLayer("down1", sd, dev, 2)
        self.mid = stack("mid", depths[2])
        self.red1 = _ConvLayer("red1", sd, dev, 1)
        self.dec1 = stack("dec1", depths[3])
        self.red0 = _ConvLayer("red0", sd, dev, 1)
        self.dec0 = stack("dec0", depths[4])
        if self.enc0[0].cin != 1:
            raise ValueError("the stem kernel is built for IN_CHANNELS == 1")
        self.head_w5 = sd["heads.weight"].detach().float().to(dev).contiguous()  # (5, C, 1, 1, 1)
        self.head_w = sd["heads.weight"].detach().float().reshape(5, d4).to(dev).contiguous()
        self.head_b = sd["heads.bias"].detach().float().to(dev).contiguous()
        self.zeros = torch.zeros(4096, dtype=torch.uint8, device=dev)
        self._bufs: Dict[Tuple, Tensor] = {}
        self.last_features: Dict[str, Tensor] = {}
        self.profile: Optional[ConvProfile] = None
        self.defer_activation = True  # single-consumer tensors stay RAW and are activated on load (tools/ A/B switch)
        self.fold_upsample = True     # decoder convs: nearest-upsample folded into the weights (sk_conv3d_upfold; tools/ A/B switch)
        self.box_store = True         # with an out_box the last conv stores only the box the heads read (sk_conv3d_box; tools/ A/B switch)
        self.stem_single_pass = False  # tools/ A/B switch: stem conv once (raw + statistics), enc0.1 activates it in LDS (fp16 mode)

    @property
    def split(self) -> bool:
        return self.precision in ("split", "mix8")   # tensors are [hi | lo] pairs

    @property
    def mix8(self) -> bool:
        return self.precision == "mix8"

    def clone_context(self) -> "HipUNet":
        """Same weights, separate activation buffers: lets two tile batches be in flight on two
        HIP streams (the HBM-bound GN/heads kernels of one overlap the MFMA-bound convs of the other)."""
        import copy
        other = copy.copy(self)
        other._bufs = {}
        other.last_features = {}
        other.profile = None
        other.__dict__.pop("_stream_ctxs", None)   # the primary's list of contexts (parallel.ShardedVolume.run) is not the clone's
        return other

    # -- reference-compatible construction ---------------------------------------------
    @classmethod
    def from_module(cls, module: torch.nn.Module, device="cuda:0", precision: str = "fp16") -> "HipUNet":
        return cls(module.state_dict(), device, getattr(module, "dims", (32, 64, 128, 64, 32)),
                   getattr(module, "depths", (2, 2, 2, 2, 2)), precision)

    # -- buffers -----------------------------------------------------------------------
    def _buf(self, tag: str, shape: Tuple[int, ...], dtype=torch.float16) -> Tensor:
        """Activation / scratch buffer ``tag``: ONE allocation per tag that only ever grows; a smaller request (the last
        tile batch of a volume -- 49 of 64 tiles at 1024x1024x256 --, another layer's partial sums) is a view of it.
        Rounds 1-3 re-allocated a tag whenever its shape changed: every step freed and re-requested tens of GB through
        torch's caching allocator, whose blocks had meanwhile been split for other tags -- the first steps after a
        precision switch then called hipMalloc inside the step (2 calls, ~480 ms on some boxes: what made the
        split-precision figure of the driver's line read 170 where the steady state is 215 Mvox/s)."""
        n = 1
        for v in shape:
            n *= int(v)
        key = (tag, dtype)
        t = self._bufs.get(key)
        if t is None or t.numel() < n:
            self._bufs.pop(key, None)
            t = None   # release the old block before asking for the larger one
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t[:n].view(shape)

    # -- layer launchers ---------------------------------------------------------------
    def _norm_act(self, layer: _ConvLayer, x: Tensor, partial: Tensor, nblk: int, apply: bool = True, mix_out: bool = False) -> Tensor:
        """GroupNorm statistics -> per-channel affine; ``apply`` runs the fused affine + SiLU pass in
        place.  With ``apply=False`` the tensor stays RAW and the (single) consumer applies the affine
        on load (gather GEMM / heads: every element is read exactly once there)."""
        B = x.shape[0]
        vox = x.shape[1] * x.shape[2] * x.shape[3]
        aff = self._buf("affine_" + layer.name, (B, 2, layer.cout), torch.float32)
        st = _ffi.stream_ptr(self.device)
        _ffi.check(_ffi.lib.sk_groupnorm_finalize(_ffi.ptr(partial), B, nblk, GN_GROUPS, layer.cout, vox,
                                                  _ffi.ptr(layer.gamma), _ffi.ptr(layer.beta), GN_EPS,
                                                  _ffi.ptr(aff), st))
        if apply:
            fn = _ffi.lib.sk_groupnorm_silu_split if self.split else _ffi.lib.sk_groupnorm_silu
            if mix_out:   # the consumer is sk_conv3d_mix8: [hi | x8 | lo8] lines
                fn = _ffi.lib.sk_groupnorm_silu_mix8
            _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(aff), B, vox, layer.cout, st))
        return aff

    def _conv(self, layer: _ConvLayer, srcs: List[Tuple], out_shape: Tuple[int, int, int],
              tag: str, activate: bool = True, store_box=None, mix_in: bool = False, mix_out: bool = False):
        """srcs: [(tensor, upsample flag[, affine])].  Returns the activated output, or
        (raw output, affine) when ``activate`` is False.  ``store_box`` = (lo, hi): the only reader of the output looks at
        this box of it (the heads, with an ``out_box``): the conv may leave the rest unwritten (sk_conv3d_box; its
        GroupNorm statistics cover the whole tile regardless)."""
        B = srcs[0][0].shape[0]
        ox, oy, oz = out_shape
        split = self.split
        lanes = 2 if split else 1   # split tensors hold [hi | lo] per voxel: twice the channels
        out = self._buf(tag, (B, ox, oy, oz, layer.cout * lanes))
        arr = (_ffi.ConvSrc * len(srcs))()
        cin = 0
        for i, src in enumerate(srcs):
            t, up = src[0], src[1]
            arr[i].data = t.data_ptr()
            arr[i].affine = src[2].data_ptr() if len(src) > 2 and src[2] is not None else None
            arr[i].c = t.shape[-1] // lanes
            arr[i].upsample = up
            cin += t.shape[-1] // lanes
        assert cin == layer.cin, (layer.name, cin, layer.cin)
        # decoder conv over cat([skip, upsample(x)]): the folded kernel where it covers the shape (fp16 mode, both
        # sources activated); it has its own workgroup count, hence its own number of partial-sum rows
        fold = (self.fold_upsample and layer.ksize == 3 and len(srcs) == 2 and bool(srcs[1][1])
                and not srcs[0][1] and arr[0].affine is None and arr[1].affine is None)
        nblk = _ffi.lib.sk_conv3d_upfold_num_blocks(ox, oy, oz, layer.cout) if fold else -1
        fold = nblk > 0
        if not fold:
            nblk = _ffi.lib.sk_conv3d_num_blocks(B, ox, oy, oz, layer.cout, layer.ksize)
        if nblk <= 0:
            raise ValueError(f"{layer.name}: unsupported output shape {out_shape}")
        partial = self._buf("partial", (B * nblk * (layer.cout // 4) * 2,), torch.float32)
        timed = self.profile is not None and layer.ksize == 3
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        if mix_in and fold:   # precision "mix8", decoder conv: both sources hold mix8 lines
            wimg, wexp = layer.packed_upfold_mix8(arr[0].c)
            _ffi.check(_ffi.lib.sk_conv3d_upfold_mix8(arr[0].data, arr[0].c, arr[1].data, arr[1].c, _ffi.ptr(wimg), wexp,
                                                      _ffi.ptr(layer.bias), _ffi.ptr(out), B, ox, oy, oz, layer.cout,
                                                      _ffi.ptr(partial), _ffi.stream_ptr(self.device)))
        elif mix_in:   # precision "mix8": the source holds [hi | x8 | lo8] lines
            assert len(srcs) == 1 and layer.ksize == 3 and layer.cin == layer.cout and arr[0].affine is None
            wimg, wexp = layer.packed_mix8()
            box = None
            if store_box is not None and not activate:
                box = (C.c_int32 * 6)(*[int(v) for v in store_box[0]], *[int(v) for v in store_box[1]])
            _ffi.check(_ffi.lib.sk_conv3d_mix8(arr, 1, _ffi.ptr(wimg), wexp, _ffi.ptr(layer.bias), _ffi.ptr(out), B, ox, oy, oz,
                                               layer.cout, _ffi.ptr(partial), _ffi.ptr(self.zeros), box,
                                               _ffi.stream_ptr(self.device)))
        elif fold:
            ufn = _ffi.lib.sk_conv3d_upfold_split if split else _ffi.lib.sk_conv3d_upfold
            _ffi.check(ufn(arr[0].data, arr[0].c, arr[1].data, arr[1].c,
                           _ffi.ptr(layer.packed_upfold(arr[0].c, split)), _ffi.ptr(layer.bias), _ffi.ptr(out),
                           B, ox, oy, oz, layer.cout, _ffi.ptr(partial), _ffi.stream_ptr(self.device)))
        elif store_box is not None and layer.ksize == 3 and not activate:
            box = (C.c_int32 * 6)(*[int(v) for v in store_box[0]], *[int(v) for v in store_box[1]])
            bfn = _ffi.lib.sk_conv3d_box_split if split else _ffi.lib.sk_conv3d_box
            _ffi.check(bfn(arr, len(srcs), _ffi.ptr(layer.packed(split)), _ffi.ptr(layer.bias),
                                              _ffi.ptr(out), B, ox, oy, oz, layer.cout, layer.ksize,
                                              _ffi.ptr(partial), _ffi.ptr(self.zeros), box, _ffi.stream_ptr(self.device)))
        else:
            fn = _ffi.lib.sk_conv3d_split if split else _ffi.lib.sk_conv3d
            _ffi.check(fn(arr, len(srcs), _ffi.ptr(layer.packed(split)), _ffi.ptr(layer.bias),
                          _ffi.ptr(out), B, ox, oy, oz, layer.cout, layer.ksize,
                          _ffi.ptr(partial), _ffi.ptr(self.zeros), _ffi.stream_ptr(self.device)))
        if timed:
            e1.record(torch.cuda.current_stream(self.device))
            self.profile.events.append((e0, e1, layer.flops_per_out_voxel * B * ox * oy * oz, layer.name))
            # split mode: three fp16 MFMA products (w_lo x_hi, w_hi x_hi, w_hi x_lo) per algorithmic product; mix8: the fp16 product
            # + one fp8 instruction stream that takes the time of 10/9 (K = 128: ten tap rows for nine) or 1 (K = 64, folded taps)
            # fp16 passes -- counted in fp16-pass equivalents against the fp16 peak
            passes = 3.0 if split else 1.0
            k128 = 1.0 + 10.0 / 9.0
            if mix_in and fold:
                per_voxel = 2.0 * layer.cout * (arr[0].c * 27 * k128 + arr[1].c * 8 * 2.0)
                passes = 1.0
            elif mix_in:
                per_voxel = layer.flops_per_out_voxel
                passes = k128 if layer.cout == 32 else 2.0
            else:
                per_voxel = 2.0 * layer.cout * (arr[0].c * 27 + arr[1].c * 8) if fold else layer.flops_per_out_voxel
            self.profile.executed_flops += per_voxel * B * ox * oy * oz * passes
        aff = self._norm_act(layer, out, partial, nblk, apply=activate, mix_out=mix_out and activate)
        return out if activate else (out, aff)

    def _down(self, layer: _ConvLayer, src, out_shape, tag: str, want_raw: bool, mix_out: bool = False, mix_writeback: bool = False):
        """Stride-2 down conv.  ``src`` = (tensor, affine | None).  With an affine the tensor is the RAW output of the
        previous block: the kernel activates it while staging it (GroupNorm affine + SiLU in LDS) and writes the
        activated values back -- the tensor is activated afterwards, as the decoder's skip conv needs it -- which saves
        the separate in-place GroupNorm pass over the skip tensor.  Returns (output, affine | None)."""
        t, aff = src
        if aff is None:
            out = self._conv(layer, [(t, 0, None)], out_shape, tag, activate=not want_raw, mix_out=mix_out)
            return out if want_raw else (out, None)
        B = t.shape[0]
        ox, oy, oz = out_shape
        split = self.split
        out = self._buf(tag, (B, ox, oy, oz, layer.cout * (2 if split else 1)))
        nblk = _ffi.lib.sk_conv3d_num_blocks(B, ox, oy, oz, layer.cout, 2)
        partial = self._buf("partial", (B * nblk * (layer.cout // 4) * 2,), torch.float32)
        fn = _ffi.lib.sk_conv3d_down_act_split if split else _ffi.lib.sk_conv3d_down_act
        if mix_writeback:   # the activated input goes back as mix8 lines (its other reader is sk_conv3d_upfold_mix8)
            fn = _ffi.lib.sk_conv3d_down_act_mix8
        _ffi.check(fn(_ffi.ptr(t), _ffi.ptr(aff), _ffi.ptr(layer.packed(split)), _ffi.ptr(layer.bias),
                      _ffi.ptr(out), B, ox, oy, oz, layer.cin, layer.cout, _ffi.ptr(partial),
                      _ffi.ptr(self.zeros), _ffi.stream_ptr(self.device)))
        aff_out = self._norm_act(layer, out, partial, nblk, apply=not want_raw, mix_out=mix_out and not want_raw)
        return (out, aff_out if want_raw else None)

    def _stem(self, layer: _ConvLayer, image: Tensor, origins, tile, mean: float, std: float,
              tag: str = "L0a", raw: bool = False, mix_out: bool = False):
        """First block (Cin = 1): normalise + conv statistics, GroupNorm finalize, then the conv again
        with the affine + SiLU fused into its epilogue: the raw tensor is never written."""
        B = len(origins)
        X, Y, Z = image.shape
        xt, yt, zt = tile
        out = self._buf(tag, (B, xt, yt, zt, layer.cout * (2 if self.split else 1)))
        nblk = _ffi.lib.sk_conv3d_stem_num_blocks(xt, yt, zt)
        partial = self._buf("partial", (B * nblk * (layer.cout // 4) * 2,), torch.float32)
        org = (C.c_int32 * (3 * B))(*[int(v) for o in origins for v in o])
        ws_bytes = _ffi.lib.sk_conv3d_stem_workspace_bytes(B, xt, yt, zt)
        ws = self._buf("stem_ws", (ws_bytes,), torch.uint8)
        st = _ffi.stream_ptr(self.device)
        if raw:   # one pass: the raw result + its statistics; the consumer activates
            _ffi.check(_ffi.lib.sk_conv3d_stem_raw(_ffi.ptr(image), X, Y, Z, org, B, xt, yt, zt, mean, std,
                                                   _ffi.ptr(layer.weight), _ffi.ptr(layer.bias), layer.cout, _ffi.ptr(out),
                                                   _ffi.ptr(partial), _ffi.ptr(ws), ws_bytes, st))
            return out, self._norm_act(layer, out, partial, nblk, apply=False)
        _ffi.check(_ffi.lib.sk_conv3d_stem(_ffi.ptr(image), X, Y, Z, org, B, xt, yt, zt, mean, std,
                                           _ffi.ptr(layer.weight), _ffi.ptr(layer.bias), layer.cout,
                                           _ffi.ptr(partial), _ffi.ptr(ws), ws_bytes, st))
        aff = self._norm_act(layer, out, partial, nblk, apply=False)
        fn = _ffi.lib.sk_conv3d_stem_apply_split if self.split else _ffi.lib.sk_conv3d_stem_apply
        if mix_out:
            fn = _ffi.lib.sk_conv3d_stem_apply_mix8
        _ffi.check(fn(B, xt, yt, zt, _ffi.ptr(layer.weight), _ffi.ptr(layer.bias),
                      _ffi.ptr(aff), _ffi.ptr(out), layer.cout, _ffi.ptr(ws), st))
        return out

    # -- forward -----------------------------------------------------------------------
    def forward_tiles(self, image: Tensor, origins: Sequence[Sequence[int]], tile: Sequence[int],
                      mean: float, std: float, keep_features: bool = False, out_box=None) -> Tensor:
        """image (X, Y, Z) fp16 on the GPU; B tile origins; tile extents (w, h, d).
        ``out_box`` = (lo, hi) tile-local: only that box of the 5-channel output is evaluated (every
        conv still covers the whole tile -- its GroupNorm statistics need it -- but the heads do not).
        The result is a view of this context's output buffer: valid until the next ``forward_tiles`` call on it."""
        _ffi.require_gpu(image, "image")
        if image.dtype != torch.float16 or image.ndim != 3:
            raise ValueError("image must be an (X, Y, Z) fp16 tensor")
        ext = tuple(int(v) for v in tile)
        # The network runs on extents padded up to a multiple of 4 (two stride-2 levels) with zeros in
        # normalised space -- exactly what oracle/unet_spec.py defines for such a crop -- and the output
        # is cropped back: a volume thinner than the 300x300x20 tile gives tiles of its own extent
        # (cropper.py:58-144), which need not be a multiple of 4.
        xt, yt, zt = ((v + 3) // 4 * 4 for v in ext)
        if (xt, yt, zt) != ext:
            if keep_features:
                raise ValueError("keep_features needs tile extents that are multiples of 4")
            full = self.forward_tiles(image, origins, (xt, yt, zt), mean, std, out_box=out_box)
            return full[:, :, :ext[0], :ext[1], :ext[2]]
        if self.precision == "fp32":
            return self._forward_fp32(image, origins, (xt, yt, zt), float(mean), float(std))
        B = len(origins)
        L0, L1, L2 = (xt, yt, zt), (xt // 2, yt // 2, zt // 2), (xt // 4, yt // 4, zt // 4)
        feats = self.last_features = {}
        # Fused GroupNorm + SiLU: a conv writes its RAW output and hands (tensor, affine) on; the consumer applies
        # silu(a*x + b) while it stages the tensor -- the stride-2 down convs and the single-chunk 3x3x3 convs in LDS, the
        # 1x1x1 convs and the heads on load -- so no separate normalisation pass touches HBM for that tensor.  The two
        # skip tensors are activated (and written back) by their stride-2 down conv, since the decoder reads them as well.
        # Where it was measured to cost more than the pass it saves, the pass stays (tools/kernel_ab.sh, 8 tiles of
        # 300x300x20: in-LDS activation inside the 64/128-channel convs +413 us for 325 us of passes; inside a conv that
        # reads an UPSAMPLED raw tensor +252 us for a 45 us pass over the low-resolution tensor; inside the single-chunk
        # 32->32 conv +188 us for a 345 us pass: kept).  ``keep_features`` and the split mode take the unfused path.
        fuse = self.defer_activation and not self.split and not keep_features
        fuse_down = self.defer_activation and not keep_features   # the stride-2 convs activate their raw input in both fast modes

        def lds_act(nxt):   # does the consuming 3x3x3 conv activate a raw input in LDS at a profit?
            return fuse and nxt.ksize == 3 and nxt.cin == 32

        def keep(name, t):
            if keep_features:
                feats[name] = (t[0] if isinstance(t, tuple) else t).clone()

        def block(layer, srcs, shape, tag, want_raw, store_box=None, mix_in=False, mix_out=False):
            """srcs: [((tensor, affine | None), upsample)]; returns (tensor, affine | None)."""
            flat = [(t, up, aff) for (t, aff), up in srcs]
            out = self._conv(layer, flat, shape, tag, activate=not want_raw, store_box=store_box, mix_in=mix_in, mix_out=mix_out)
            return out if want_raw else (out, None)

        # precision "mix8": an activated 32-channel L0 tensor whose only reader is a 32 -> 32 3x3x3 conv is stored as
        # [hi | x8 | lo8] lines and that conv runs sk_conv3d_mix8 (keep_features wants plain pairs: the split path then)
        mix8 = self.mix8 and not keep_features

        def mixes(nxt):
            return mix8 and nxt.ksize == 3 and nxt.cin == nxt.cout and nxt.cout in (32, 64, 128)

        def folds_mix(layer, shape, skip):
            """Does this decoder conv run sk_conv3d_upfold_mix8?  Then its skip tensor (activated and written back by the fused
            stride-2 conv: ``skip`` arrived raw) and its upsampled source are produced as mix8 lines."""
            return (mix8 and self.fold_upsample and layer.ksize == 3 and skip[1] is not None
                    and _ffi.lib.sk_conv3d_upfold_num_blocks(shape[0], shape[1], shape[2], layer.cout) > 0)

        stem_raw = self.stem_single_pass and len(self.enc0) > 1 and lds_act(self.enc0[1])
        a_mix = len(self.enc0) > 1 and mixes(self.enc0[1])
        a = self._stem(self.enc0[0], image, origins, L0, float(mean), float(std),
                       "skip0" if len(self.enc0) == 1 else "L0a", raw=stem_raw, mix_out=a_mix)
        a = a if stem_raw else (a, None)
        keep("enc0.0", a)
        tags = ["L0b", "L0a"]
        for i, layer in enumerate(self.enc0[1:]):
            last = i == len(self.enc0) - 2
            raw = (fuse_down and (self.down0.cin, self.down0.cout) == (32, 64)) if last else lds_act(self.enc0[i + 2])
            nxt_mix = not last and not raw and mixes(self.enc0[i + 2])
            a = block(layer, [(a, 0)], L0, "skip0" if last else tags[i % 2], raw, mix_in=a_mix, mix_out=nxt_mix)
            a_mix = nxt_mix
            keep(layer.name, a)
        s0 = a
        a_mix = len(self.enc1) > 0 and mixes(self.enc1[0])
        dec0_mix = folds_mix(self.dec0[0], L0, s0)
        a = self._down(self.down0, s0, L1, "L1a", False, mix_out=a_mix, mix_writeback=dec0_mix)   # activates s0 in place when it came in raw
        s0 = (s0[0], None)
        keep("down0", a)
        tags = ["L1b", "L1a"]
        for i, layer in enumerate(self.enc1):
            last = i == len(self.enc1) - 1
            raw = last and fuse_down and (self.down1.cin, self.down1.cout) == (64, 128)
            nxt_mix = not last and mixes(self.enc1[i + 1])
            a = block(layer, [(a, 0)], L1, "skip1" if last else tags[i % 2], raw, mix_in=a_mix, mix_out=nxt_mix)
            a_mix = nxt_mix
            keep(layer.name, a)
        s1 = a
        a_mix = len(self.mid) > 0 and mixes(self.mid[0])
        dec1_mix = folds_mix(self.dec1[0], L1, s1)
        a = self._down(self.down1, s1, L2, "L2a", False, mix_out=a_mix, mix_writeback=dec1_mix)
        s1 = (s1[0], None)
        keep("down1", a)
        tags = ["L2b", "L2a"]
        for i, layer in enumerate(self.mid):
            # the last one is read only by red1 (1x1x1, gather GEMM): activated on load there (both fast modes)
            last = i == len(self.mid) - 1
            raw = self.defer_activation and not keep_features and last
            nxt_mix = not last and mixes(self.mid[i + 1])
            a = block(layer, [(a, 0)], L2, tags[i % 2], raw, mix_in=a_mix, mix_out=nxt_mix)
            a_mix = nxt_mix
            keep(layer.name, a)
        r1 = block(self.red1, [(a, 0)], L2, "L2r", False, mix_out=dec1_mix)
        keep("red1", r1)
        tags = ["L1a", "L1b"]
        a_mix = False
        for i, layer in enumerate(self.dec1):
            last = i == len(self.dec1) - 1
            raw = self.defer_activation and not keep_features and last  # last: consumed by red0 (on load)
            src = [(s1, 0), (r1, 1)] if i == 0 else [(a, 0)]
            nxt_mix = not last and mixes(self.dec1[i + 1])
            a = block(layer, src, L1, tags[i % 2], raw, mix_in=dec1_mix if i == 0 else a_mix, mix_out=nxt_mix)
            a_mix = nxt_mix
            keep(layer.name, a)
        r0 = block(self.red0, [(a, 0)], L1, "L1r", False, mix_out=dec0_mix)
        keep("red0", r0)
        tags = ["L0a", "L0b"]
        a_mix = False
        for i, layer in enumerate(self.dec0):
            last = i == len(self.dec0) - 1
            raw = (self.defer_activation and last) or (not last and lds_act(self.dec0[i + 1]))  # last: consumed by the heads
            src = [(s0, 0), (r0, 1)] if i == 0 else [(a, 0)]
            # the last conv's raw output is read by the heads alone, and with an out_box only inside it
            sbox = out_box if (last and raw and out_box is not None and not keep_features and self.box_store) else None
            nxt_mix = not last and not raw and mixes(self.dec0[i + 1])
            a = block(layer, src, L0, tags[i % 2], raw, store_box=sbox, mix_in=dec0_mix if i == 0 else a_mix, mix_out=nxt_mix)
            a_mix = nxt_mix
            keep(layer.name, a)
        a, aff = a
        out5 = self._buf("out5", (B, 5, xt, yt, zt))
        i3 = C.c_int32 * 3
        blo = i3(*[int(v) for v in out_box[0]]) if out_box is not None else None
        bhi = i3(*[int(v) for v in out_box[1]]) if out_box is not None else None
        fn = _ffi.lib.sk_heads_split if self.split else _ffi.lib.sk_heads
        _ffi.check(fn(_ffi.ptr(a), _ffi.ptr(aff), _ffi.ptr(self.head_w), _ffi.ptr(self.head_b),
                      _ffi.ptr(out5), B, xt, yt, zt, self.dims[4], blo, bhi, _ffi.stream_ptr(self.device)))
        return out5

    # -- fp32 precision mode ------------------------------------------------------------
    def _conv_f32(self, layer, srcs, out_shape, weight=None, bias=None, cout=None, norm=True) -> Tensor:
        B = srcs[0][0].shape[0]
        ox, oy, oz = out_shape
        cout = layer.cout if cout is None else cout
        weight = layer.weight_f32 if weight is None else weight
        bias = layer.bias if bias is None else bias
        out = torch.empty((B, ox, oy, oz, cout), dtype=torch.float32, device=self.device)
        nblk = _ffi.lib.sk_conv3d_f32_num_blocks(ox, oy, oz)
        partial = torch.empty((B, nblk, cout // 4, 2), dtype=torch.float32, device=self.device) if norm else None
        arr = (_ffi.ConvSrc * len(srcs))()
        for i, (t, up) in enumerate(srcs):
            arr[i].data = t.data_ptr()
            arr[i].affine = None
            arr[i].c = t.shape[-1]
            arr[i].upsample = up
        st = _ffi.stream_ptr(self.device)
        _ffi.check(_ffi.lib.sk_conv3d_f32(arr, len(srcs), _ffi.ptr(weight), _ffi.ptr(bias), _ffi.ptr(out), B,
                                          ox, oy, oz, cout, layer.ksize if layer is not None else 1,
                                          _ffi.ptr(partial), st))
        if norm:
            aff = torch.empty((B, 2, cout), dtype=torch.float32, device=self.device)
            vox = ox * oy * oz
            _ffi.check(_ffi.lib.sk_groupnorm_finalize(_ffi.ptr(partial), B, nblk, GN_GROUPS, cout, vox,
                                                      _ffi.ptr(layer.gamma), _ffi.ptr(layer.beta), GN_EPS,
                                                      _ffi.ptr(aff), st))
            _ffi.check(_ffi.lib.sk_groupnorm_silu_f32(_ffi.ptr(out), _ffi.ptr(aff), B, vox, cout, st))
        return out

    def _forward_fp32(self, image: Tensor, origins, tile, mean: float, std: float) -> Tensor:
        xt, yt, zt = tile
        L0, L1, L2 = tile, (xt // 2, yt // 2, zt // 2), (xt // 4, yt // 4, zt // 4)
        def crop(x, y, z):  # normalise (eval.py:139, fp16 arithmetic), then zero-pad an overhanging tile
            c = image[x:x + xt, y:y + yt, z:z + zt].sub(mean).div(std).float()
            return torch.nn.functional.pad(c, (0, zt - c.shape[2], 0, yt - c.shape[1], 0, xt - c.shape[0]))

        a = torch.stack([crop(x, y, z) for (x, y, z) in origins]).unsqueeze(-1).contiguous()
        for layer in self.enc0:
            a = self._conv_f32(layer, [(a, 0)], L0)
        s0 = a
        a = self._conv_f32(self.down0, [(s0, 0)], L1)
        for layer in self.enc1:
            a = self._conv_f32(layer, [(a, 0)], L1)
        s1 = a
        a = self._conv_f32(self.down1, [(s1, 0)], L2)
        for layer in self.mid:
            a = self._conv_f32(layer, [(a, 0)], L2)
        r1 = self._conv_f32(self.red1, [(a, 0)], L2)
        for i, layer in enumerate(self.dec1):
            a = self._conv_f32(layer, [(s1, 0), (r1, 1)] if i == 0 else [(a, 0)], L1)
        r0 = self._conv_f32(self.red0, [(a, 0)], L1)
        for i, layer in enumerate(self.dec0):
            a = self._conv_f32(layer, [(s0, 0), (r0, 1)] if i == 0 else [(a, 0)], L0)
        y = self._conv_f32(None, [(a, 0)], L0, weight=self.head_w5, bias=self.head_b, cout=5, norm=False)
        y = y.permute(0, 4, 1, 2, 3)
        return torch.cat([torch.tanh(y[:, 0:3]), torch.sigmoid(y[:, 3:5])], dim=1).contiguous()

    def flops_per_tile_voxel(self) -> float:
        """Algorithmic conv FLOPs per full-resolution tile voxel (2*Cin*Cout*k^3 / downsampling)."""
        f = 0.0
        for layers, s in ((self.enc0, 1), ([self.down0], 8), (self.enc1, 8), ([self.down1], 64),
                          (self.mid, 64), ([self.red1], 64), (self.dec1, 8), ([self.red0], 8),
                          (self.dec0, 1)):
            f += sum(l.flops_per_out_voxel for l in layers) / s
        return f + 2.0 * self.dims[4] * 5


# what this build's network implements of the reference's MODEL config (skoots/config.py:20-34)
SUPPORTED_ARCHITECTURE = "skoots_amd_unet"   # oracle/unet_spec.py; NOT bism's "bism_unext" / "bism_unet" (absent package)


def cfg_to_model(cfg, device="cuda:0", state_dict: Optional[Dict[str, Tensor]] = None,
                 precision: str = "fp16") -> HipUNet:
    """Counterpart of ``cfg_to_bism_model`` (skoots/lib/utils.py:17-107): reads the same
    ``cfg.MODEL`` keys from an attribute/dict config.  The network body is the build's own U-Net
    (oracle/unet_spec.py: Conv3d k=3 -> GroupNorm(8) -> SiLU blocks): a config that asks for anything else --
    the reference's default ``bism_unext`` with LayerNorm / GELU / 7^3 depthwise kernels, whose code lives in
    the absent ``bism`` package -- raises instead of silently running a different network.  Checkpoints written
    by the reference's trainer (pickled yacs ``CfgNode`` + bism ``UNeXT_3D`` keys) are therefore NOT loadable;
    checkpoints written by ``skoots_amd.train`` are."""
    model = cfg["MODEL"] if isinstance(cfg, dict) else cfg.MODEL
    get = (lambda k, d: model.get(k, d)) if isinstance(model, dict) else (lambda k, d: getattr(model, k, d))
    dims, depths = get("DIMS", [32, 64, 128, 64, 32]), get("DEPTHS", [2, 2, 2, 2, 2])
    if get("IN_CHANNELS", 1) != 1:
        raise RuntimeError("IN_CHANNELS must be 1")
    for key, ok in (("ARCHITECTURE", (SUPPORTED_ARCHITECTURE,)), ("NORMALIZATION", ("groupnorm",)),
                    ("ACTIVATION", ("silu",)), ("KERNEL_SIZE", (3,))):
        v = get(key, ok[0])
        if v not in ok:
            raise RuntimeError(f"MODEL.{key}={v!r} is not implemented by skoots_amd (supported: {ok}); "
                               "bism architectures need the bism package, which this build does not reimplement")
    if state_dict is None:
        raise RuntimeError("a model_state_dict is required (random init lives in oracle/unet_spec.py)")
    return HipUNet(state_dict, device, dims, depths, precision)


def random_state_dict(dims=(32, 64, 128, 64, 32), depths=(2, 2, 2, 2, 2), seed: int = 101196) -> Dict[str, Tensor]:
    """Deterministic random-init parameters under the module's key names (seed: train/engine.py:53);
    what a from-scratch training run starts from and what smoke() / the benches run on."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def conv(name, cin, cout, k):
        fan = cin * k ** 3
        sd[name + ".conv.weight"] = (torch.rand((cout, cin, k, k, k), generator=g) * 2 - 1) / fan ** 0.5
        sd[name + ".conv.bias"] = (torch.rand(cout, generator=g) * 2 - 1) / fan ** 0.5
        sd[name + ".norm.weight"] = torch.rand(cout, generator=g) + 0.5
        sd[name + ".norm.bias"] = torch.rand(cout, generator=g) * 0.6 - 0.3

    d0, d1, d2, d3, d4 = dims
    for i in range(depths[0]):
        conv(f"enc0.{i}", 1 if i == 0 else d0, d0, 3)
    conv("down0", d0, d1, 2)
    for i in range(depths[1]):
        conv(f"enc1.{i}", d1, d1, 3)
    conv("down1", d1, d2, 2)
    for i in range(depths[2]):
        conv(f"mid.{i}", d2, d2, 3)
    conv("red1", d2, d3, 1)
    for i in range(depths[3]):
        conv(f"dec1.{i}", d1 + d3 if i == 0 else d3, d3, 3)
    conv("red0", d3, d4, 1)
    for i in range(depths[4]):
        conv(f"dec0.{i}", d0 + d4 if i == 0 else d4, d4, 3)
    sd["heads.weight"] = (torch.rand((5, d4, 1, 1, 1), generator=g) * 2 - 1) / d4 ** 0.5
    sd["heads.bias"] = (torch.rand(5, generator=g) * 2 - 1) / d4 ** 0.5
    return sd


def smoke_model(device="cuda:0") -> Optional[HipUNet]:
    """Deterministic random-init network for __graft_entry__.smoke() (built without the oracle)."""
    dims, depths = (32, 64, 128, 64, 32), (2, 2, 2, 2, 2)
    return HipUNet(random_state_dict(dims, depths), device, dims, depths)


# ----------------------------------------------------------------------------------------
# Operator-level entry points (used by the parity tests and by bench.py's conv-only leg)
# ----------------------------------------------------------------------------------------
def pack_conv_weight(weight: Tensor, device, split: bool = False) -> Tensor:
    """(cout, cin, k, k, k) fp32 -> MFMA A-fragment order (fp16 bytes) on the device; ``split``: the hi + lo
    fragment sets of ``sk_conv3d_split``."""
    w = weight.detach().float().cpu().contiguous().numpy()
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    fpt = w.ctypes.data_as(C.POINTER(C.c_float))
    fn = _ffi.lib.sk_conv3d_pack_weight_split_host if split else _ffi.lib.sk_conv3d_pack_weight_host
    nbytes = fn(fpt, cout, cin, k, None)
    if nbytes < 0:
        _ffi.check(int(nbytes))
    buf = np.empty(nbytes, dtype=np.uint8)
    fn(fpt, cout, cin, k, buf.ctypes.data_as(C.c_void_p))
    return torch.from_numpy(buf).to(device)


def pack_conv_weight_mix8(weight: Tensor, device) -> Tuple[Tensor, int]:
    """(C, C, 3, 3, 3) fp32, C = 32 | 64 | 128 -> (weight image of ``sk_conv3d_mix8`` on the device, its fp8 scale exponent)."""
    w = weight.detach().float().cpu().contiguous().numpy()
    fpt = w.ctypes.data_as(C.POINTER(C.c_float))
    nbytes = _ffi.lib.sk_conv3d_pack_weight_mix8_host(fpt, w.shape[0], w.shape[1], None, None)
    if nbytes < 0:
        _ffi.check(int(nbytes))
    buf = np.empty(nbytes, dtype=np.uint8)
    exp = C.c_int32(0)
    _ffi.lib.sk_conv3d_pack_weight_mix8_host(fpt, w.shape[0], w.shape[1], buf.ctypes.data_as(C.c_void_p), C.byref(exp))
    return torch.from_numpy(buf).to(device), int(exp.value)


def mix8_line(hi: Tensor, x8: Tensor, lo8: Tensor) -> Tensor:
    """The mix8 voxel line from its three parts: hi (..., C) fp16, x8 and lo8 (..., C) float8_e4m3fn (the CODES are
    stored: x8 stands for 16 x, lo8 for 2^15 (x - hi)) -> (..., 2C) fp16-typed tensor
    [hi (C) | per 32-channel chunk: x8 (32 bytes) | lo8 (32 bytes)]."""
    C = hi.shape[-1]
    lead = hi.shape[:-1]
    xb = x8.contiguous().view(torch.uint8).reshape(lead + (C // 32, 1, 32))
    lb = lo8.contiguous().view(torch.uint8).reshape(lead + (C // 32, 1, 32))
    tail = torch.cat([xb, lb], dim=-2).reshape(lead + (2 * C,))
    b = torch.cat([hi.contiguous().view(torch.uint8), tail], dim=-1)
    return b.contiguous().view(torch.float16)


def mix8_of(x: Tensor) -> Tensor:
    """fp32 (..., C) -> its mix8 line (host restatement of sk_groupnorm_silu_mix8's store)."""
    hi = x.half()
    x8 = (x * 16.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    lo8 = ((x - hi.float()) * 32768.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return mix8_line(hi, x8, lo8)


def conv3d_mix8(src: Tensor, packed_weight: Tensor, scale_exp: int, bias: Tensor, out_shape: Sequence[int], zeros: Tensor,
                store_box=None, out: Optional[Tensor] = None):
    """Raw 3x3x3 C -> C conv of precision "mix8" (C = 32 | 64 | 128): src (B, x, y, z, 2C) mix8 lines -> ((B, x, y, z, 2C) split
    pair, gn_partial)."""
    _ffi.require_gpu(src, "src")
    B, dev = src.shape[0], src.device
    ch = src.shape[-1] // 2
    ox, oy, oz = (int(v) for v in out_shape)
    if out is None:
        out = torch.empty((B, ox, oy, oz, 2 * ch), dtype=torch.float16, device=dev)
    nblk = _ffi.lib.sk_conv3d_num_blocks(B, ox, oy, oz, ch, 3)
    partial = torch.zeros((B, nblk, ch // 4, 2), dtype=torch.float32, device=dev)
    arr = (_ffi.ConvSrc * 1)()
    arr[0].data, arr[0].c, arr[0].upsample, arr[0].affine = src.data_ptr(), ch, 0, None
    box = (C.c_int32 * 6)(*[int(v) for v in store_box]) if store_box is not None else None
    _ffi.check(_ffi.lib.sk_conv3d_mix8(arr, 1, _ffi.ptr(packed_weight), int(scale_exp), _ffi.ptr(bias), _ffi.ptr(out), B, ox, oy, oz,
                                       ch, _ffi.ptr(partial), _ffi.ptr(zeros), box, _ffi.stream_ptr(dev)))
    return out, partial


def split_pair(x: Tensor) -> Tensor:
    """fp32 (..., C) -> the split layout (..., 2C) fp16 = [hi | lo], hi = fp16(x), lo = fp16(x - hi)."""
    hi = x.half()
    return torch.cat([hi, (x - hi.float()).half()], dim=-1).contiguous()


def join_pair(x: Tensor) -> Tensor:
    """Split layout (..., 2C) fp16 -> fp32 (..., C) = hi + lo."""
    c = x.shape[-1] // 2
    return x[..., :c].float() + x[..., c:].float()


def conv3d(srcs: List[Tuple[Tensor, int]], packed_weight: Tensor, bias: Tensor, cout: int, ksize: int,
           out_shape: Sequence[int], zeros: Tensor, want_stats: bool = True, split: bool = False):
    """Raw conv (no normalisation): srcs [(B,x,y,z,c) fp16 tensor, upsample flag] ->
    ((B, ox, oy, oz, cout) fp16, gn_partial (B, nblk, cout/4, 2) fp32 or None).  ``split``: sources and
    output are split pairs (..., 2c) (see :func:`split_pair`), weight packed with ``split=True``."""
    B = srcs[0][0].shape[0]
    dev = srcs[0][0].device
    ox, oy, oz = (int(v) for v in out_shape)
    lanes = 2 if split else 1
    out = torch.empty((B, ox, oy, oz, cout * lanes), dtype=torch.float16, device=dev)
    nblk = _ffi.lib.sk_conv3d_num_blocks(B, ox, oy, oz, cout, ksize)
    if nblk <= 0:
        raise ValueError(f"unsupported conv output shape {tuple(out_shape)}")
    partial = torch.zeros((B, nblk, cout // 4, 2), dtype=torch.float32, device=dev) if want_stats else None
    arr = (_ffi.ConvSrc * len(srcs))()
    for i, (t, up) in enumerate(srcs):
        _ffi.require_gpu(t, "src")
        arr[i].data = t.data_ptr()
        arr[i].affine = None
        arr[i].c = t.shape[-1] // lanes
        arr[i].upsample = up
    fn = _ffi.lib.sk_conv3d_split if split else _ffi.lib.sk_conv3d
    _ffi.check(fn(arr, len(srcs), _ffi.ptr(packed_weight), _ffi.ptr(bias), _ffi.ptr(out), B,
                  ox, oy, oz, cout, ksize, _ffi.ptr(partial), _ffi.ptr(zeros), _ffi.stream_ptr(dev)))
    return out, partial


def pack_conv_weight_upfold(weight: Tensor, c_skip: int, device, split: bool = False) -> Tensor:
    """Torch-layout (cout, c_skip + c_up, 3, 3, 3) fp32 weight -> the fragments of ``sk_conv3d_upfold`` (``split``:
    ``sk_conv3d_upfold_split``) on ``device``."""
    w = weight.detach().float().cpu().contiguous()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    fn = _ffi.lib.sk_conv3d_pack_weight_upfold_split_host if split else _ffi.lib.sk_conv3d_pack_weight_upfold_host
    fpt = w.numpy().ctypes.data_as(C.POINTER(C.c_float))
    nbytes = fn(fpt, cout, c_skip, cin - c_skip, None)
    if nbytes < 0:
        _ffi.check(int(nbytes))
    buf = np.empty(nbytes, dtype=np.uint8)
    fn(fpt, cout, c_skip, cin - c_skip, buf.ctypes.data_as(C.c_void_p))
    return torch.from_numpy(buf).to(device)


def pack_conv_weight_upfold_mix8(weight: Tensor, c_skip: int, device) -> Tuple[Tensor, int]:
    """Torch-layout (cout, c_skip + c_up, 3, 3, 3) fp32 weight -> (weight image of ``sk_conv3d_upfold_mix8``, fp8 scale exponent)."""
    w = weight.detach().float().cpu().contiguous()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    fn = _ffi.lib.sk_conv3d_pack_weight_upfold_mix8_host
    fpt = w.numpy().ctypes.data_as(C.POINTER(C.c_float))
    nbytes = fn(fpt, cout, c_skip, cin - c_skip, None, None)
    if nbytes < 0:
        _ffi.check(int(nbytes))
    buf = np.empty(nbytes, dtype=np.uint8)
    exp = C.c_int32(0)
    fn(fpt, cout, c_skip, cin - c_skip, buf.ctypes.data_as(C.c_void_p), C.byref(exp))
    return torch.from_numpy(buf).to(device), int(exp.value)


def conv3d_upfold_mix8(skip: Tensor, up: Tensor, packed_weight: Tensor, scale_exp: int, bias: Tensor, cout: int):
    """``conv3d_upfold`` for precision "mix8": skip / up hold mix8 lines (:func:`mix8_of`), the result is a split pair."""
    _ffi.require_gpu(skip, "skip")
    _ffi.require_gpu(up, "up")
    B, ox, oy, oz = (int(v) for v in skip.shape[:4])
    if tuple(up.shape[:4]) != (B, ox // 2, oy // 2, oz // 2):
        raise ValueError(f"up {tuple(up.shape)} is not half of skip {tuple(skip.shape)}")
    nblk = _ffi.lib.sk_conv3d_upfold_num_blocks(ox, oy, oz, cout)
    if nblk <= 0:
        raise ValueError(f"sk_conv3d_upfold does not cover the output shape {(ox, oy, oz)} / cout {cout}")
    out = torch.empty((B, ox, oy, oz, cout * 2), dtype=torch.float16, device=skip.device)
    partial = torch.zeros((B, nblk, cout // 4, 2), dtype=torch.float32, device=skip.device)
    _ffi.check(_ffi.lib.sk_conv3d_upfold_mix8(_ffi.ptr(skip), skip.shape[-1] // 2, _ffi.ptr(up), up.shape[-1] // 2, _ffi.ptr(packed_weight),
                                              int(scale_exp), _ffi.ptr(bias), _ffi.ptr(out), B, ox, oy, oz, cout, _ffi.ptr(partial),
                                              _ffi.stream_ptr(skip.device)))
    return out, partial


def conv3d_upfold(skip: Tensor, up: Tensor, packed_weight: Tensor, bias: Tensor, cout: int, want_stats: bool = True,
                  split: bool = False):
    """Raw decoder conv over cat([skip, nearest-upsample(up)]) with the upsample folded into the weights:
    skip (B, x, y, z, c) fp16, up (B, x/2, y/2, z/2, c') fp16 -> ((B, x, y, z, cout) fp16, gn_partial or None).
    ``split``: the tensors are split pairs (..., 2c) (:func:`split_pair`), the weight packed with ``split=True``."""
    _ffi.require_gpu(skip, "skip")
    _ffi.require_gpu(up, "up")
    B, ox, oy, oz = (int(v) for v in skip.shape[:4])
    if tuple(up.shape[:4]) != (B, ox // 2, oy // 2, oz // 2):
        raise ValueError(f"up {tuple(up.shape)} is not half of skip {tuple(skip.shape)}")
    nblk = _ffi.lib.sk_conv3d_upfold_num_blocks(ox, oy, oz, cout)
    if nblk <= 0:
        raise ValueError(f"sk_conv3d_upfold does not cover the output shape {(ox, oy, oz)} / cout {cout}")
    lanes = 2 if split else 1
    out = torch.empty((B, ox, oy, oz, cout * lanes), dtype=torch.float16, device=skip.device)
    partial = torch.zeros((B, nblk, cout // 4, 2), dtype=torch.float32, device=skip.device) if want_stats else None
    fn = _ffi.lib.sk_conv3d_upfold_split if split else _ffi.lib.sk_conv3d_upfold
    _ffi.check(fn(_ffi.ptr(skip), skip.shape[-1] // lanes, _ffi.ptr(up), up.shape[-1] // lanes, _ffi.ptr(packed_weight),
                  _ffi.ptr(bias), _ffi.ptr(out), B, ox, oy, oz, cout, _ffi.ptr(partial), _ffi.stream_ptr(skip.device)))
    return out, partial


def groupnorm_silu_(x: Tensor, partial: Tensor, gamma: Tensor, beta: Tensor, groups: int = GN_GROUPS,
                    eps: float = GN_EPS) -> Tensor:
    """In place GroupNorm (statistics from the conv partials) + SiLU on (B, x, y, z, C) fp16."""
    B, C_ = x.shape[0], x.shape[-1]
    vox = x[0].numel() // C_
    aff = torch.empty((B, 2, C_), dtype=torch.float32, device=x.device)
    st = _ffi.stream_ptr(x.device)
    _ffi.check(_ffi.lib.sk_groupnorm_finalize(_ffi.ptr(partial), B, partial.shape[1], groups, C_, vox,
                                              _ffi.ptr(gamma), _ffi.ptr(beta), eps, _ffi.ptr(aff), st))
    _ffi.check(_ffi.lib.sk_groupnorm_silu(_ffi.ptr(x), _ffi.ptr(aff), B, vox, C_, st))
    return x
