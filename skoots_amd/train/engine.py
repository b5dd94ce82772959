"""One training step of the U-Net on the HIP kernels (BASELINE.json configs[4]).

Reference: the inner step of skoots/train/engine.py:456-499 --

    out = model(images)                                   # (B, 5, X, Y, Z)
    embedding = vector_to_embedding(vector_scale, out[:, 0:3])
    emb_prob  = baked_embed_to_prob(embedding, baked, sigma(e))
    loss = w_e * tversky_e(emb_prob, masks > 0) + w_p * tversky_p(out[:, [-1]], masks > 0)
         + w_s * tversky_s(out[:, [-2]], skele_masks > 0)
    loss.backward(); optimizer.step()                     # AdamW, config.py:96-101

Here the same step runs as explicit kernels of libskoots_hip.so (no autograd): the forward
keeps each block's raw conv output, GroupNorm affine and statistics; the loss and its gradient
come from one fused kernel pair; the backward walks the recorded layer list in reverse (GN+SiLU
backward, weight gradient, data gradient); AdamW updates one flat parameter buffer.  All fp32
(the reference runs bf16 with channels_last_3d, engine.py:68,107-109; fp32 is the higher
precision).  The network graph is oracle/unet_spec.py's; data loading, augmentation, schedulers
and logging of the reference's loop are out of scope.
"""
from __future__ import annotations

import os

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .. import _ffi
from ..unet import GN_EPS, GN_GROUPS


class _Layer:
    """Views of one conv (+ GroupNorm) layer's parameters and gradients in the flat buffers."""

    def __init__(self, name: str, ksize: int, norm: bool):
        self.name, self.ksize, self.norm = name, ksize, norm
        self.weight = self.bias = self.gamma = self.beta = None
        self.g_weight = self.g_bias = self.g_gamma = self.g_beta = None
        self.cin = self.cout = 0


TRAIN_PRECISIONS = ("fp32", "mixed", "bf16")


class _Lib:
    """``libskoots_hip.so`` with the 16-bit entry points resolved to one build: "" = fp16, "_bf16" = the bf16 twins
    (include/skoots_hip_bf16.h).  Dtype-independent functions resolve to their plain names.

    ``profile`` (a :class:`skoots_amd.profile.KernelProfile`): HIP-event spans around every MFMA conv launch of the
    step -- forward and data-gradient convs (``sk_conv3d``) and weight gradients (``sk_train_conv_wgrad_f16``) -- with
    their algorithmic FLOPs, for bench.py's training roofline."""

    def __init__(self, suffix: str, device=None):
        self._suffix = suffix
        self._device = device
        self.profile = None

    def __getattr__(self, name: str):
        fn = getattr(_ffi.lib, name + self._suffix if (self._suffix and name in _ffi.BF16_TWINS) else name)
        if name in ("sk_conv3d", "sk_train_conv_wgrad_f16"):
            raw, kind = fn, ("conv_fwd_dgrad" if name == "sk_conv3d" else "conv_wgrad")

            def fn(srcs, n_src, *rest):
                if self.profile is None:
                    return raw(srcs, n_src, *rest)
                # sk_conv3d: (w, bias, out, B, ox, oy, oz, cout, k, ...); wgrad: (dy, scale, B, ox, oy, oz, cout, k, ...)
                B, ox, oy, oz, cout, k = rest[3:9] if name == "sk_conv3d" else rest[2:8]
                cin = sum(int(srcs[i].c) for i in range(n_src))
                with self.profile.span(kind, self._device, 2.0 * cin * cout * k ** 3 * B * ox * oy * oz):
                    return raw(srcs, n_src, *rest)
        setattr(self, name, fn)
        return fn


class TrainUNet:
    """The U-Net of oracle/unet_spec.py with fp32 master parameters on the GPU.

    ``forward(images)`` -> logits (B, X, Y, Z, 5) (pre-tanh / pre-sigmoid head outputs) and records
    what the backward needs; ``backward(dlogits)`` fills ``flat_grad``."""

    def __init__(self, state_dict: Dict[str, Tensor], device="cuda:0",
                 dims: Sequence[int] = (32, 64, 128, 64, 32), depths: Sequence[int] = (2, 2, 2, 2, 2),
                 precision: str = "fp32", f16_grad_handoff: bool = True):
        """``f16_grad_handoff=False``: single-reader data gradients travel as fp32 copies instead of scaled 16-bit tensors
        (the hand-off is lossy by one 16-bit rounding per tensor: parameter gradients move by ~1e-4 of the largest one).

        ``precision``: "fp32" (every kernel fp32; the parity mode), "bf16" (below, on bf16 tensors and bf16 matrix
        instructions: BASELINE configs[4]'s dtype) or "mixed" (fp32 master weights, GroupNorm,
        loss and optimizer; the convolutions of the forward pass, the data gradients and the weight gradients
        on the fp16 MFMA kernels with fp32 accumulation, output gradients scaled per tensor by a power of
        two -- the counterpart of the reference's bf16 step, engine.py:68,107-109)."""
        if precision not in TRAIN_PRECISIONS:
            raise ValueError(f"precision must be one of {TRAIN_PRECISIONS}")
        self.precision = precision
        # "bf16": the mixed step on the bf16 build of the same kernels (library entry points *_bf16): bf16 storage of
        # activations / output gradients and v_mfma_*_bf16 -- the dtype the reference trains in (engine.py:68,107-109)
        self.fast16 = precision in ("mixed", "bf16")
        self.t16 = torch.bfloat16 if precision == "bf16" else torch.float16
        self._L = _Lib("_bf16" if precision == "bf16" else "", torch.device(device))
        self.device = torch.device(device)
        self.dims, self.depths = tuple(dims), tuple(depths)
        # A/B switches of the mixed step (tools/bench_train.py sets them; defaults = the fast choices)
        self.fast_stem = True          # stem as an fp16-operand fast block
        self.fast_heads = True         # heads straight on the fp16 activation
        self.f16_grad_handoff = bool(f16_grad_handoff)   # single-reader fp16 data gradients handed on without an fp32 copy
        # tests only: a list here makes backward() record, per fast block, copies of exactly the 16-bit tensors its kernels
        # read and wrote (sources, raw output, incoming gradient, dy, data gradients, scales) next to the parameter
        # gradients they produced, so that every kernel of the step can be replayed in torch ON THE SAME OPERANDS
        # (tests/test_hip_train.py: test_bf16_step_kernels_replayed_in_situ)
        self.audit: Optional[list] = None

        def stack(name, n):
            return [_Layer(f"{name}.{i}", 3, True) for i in range(n)]

        self.enc0 = stack("enc0", depths[0])
        self.down0 = _Layer("down0", 2, True)
        self.enc1 = stack("enc1", depths[1])
        self.down1 = _Layer("down1", 2, True)
        self.mid = stack("mid", depths[2])
        self.red1 = _Layer("red1", 1, True)
        self.dec1 = stack("dec1", depths[3])
        self.red0 = _Layer("red0", 1, True)
        self.dec0 = stack("dec0", depths[4])
        self.heads = _Layer("heads", 1, False)
        self.layers: List[_Layer] = (self.enc0 + [self.down0] + self.enc1 + [self.down1] + self.mid + [self.red1] +
                                     self.dec1 + [self.red0] + self.dec0 + [self.heads])
        # flat parameter / gradient buffers, state_dict key order of the module
        self.param_names: List[str] = []
        shapes = []
        for l in self.layers:
            keys = ([f"{l.name}.conv.weight", f"{l.name}.conv.bias", f"{l.name}.norm.weight", f"{l.name}.norm.bias"]
                    if l.norm else [f"{l.name}.weight", f"{l.name}.bias"])
            for k in keys:
                if k not in state_dict:
                    raise KeyError(f"state_dict lacks {k}")
                self.param_names.append(k)
                shapes.append(tuple(state_dict[k].shape))
        total = sum(math.prod(s) for s in shapes)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._views: Dict[str, Tuple[Tensor, Tensor]] = {}
        off = 0
        for k, s in zip(self.param_names, shapes):
            n = math.prod(s)
            p = self.flat_param[off:off + n].view(s)
            p.copy_(state_dict[k].detach().float())
            self._views[k] = (p, self.flat_grad[off:off + n].view(s))
            off += n
        for l in self.layers:
            if l.norm:
                (l.weight, l.g_weight), (l.bias, l.g_bias) = self._views[f"{l.name}.conv.weight"], self._views[f"{l.name}.conv.bias"]
                (l.gamma, l.g_gamma), (l.beta, l.g_beta) = self._views[f"{l.name}.norm.weight"], self._views[f"{l.name}.norm.bias"]
            else:
                (l.weight, l.g_weight), (l.bias, l.g_bias) = self._views[f"{l.name}.weight"], self._views[f"{l.name}.bias"]
            l.cout, l.cin = int(l.weight.shape[0]), int(l.weight.shape[1])
            if tuple(l.weight.shape[2:]) != (l.ksize,) * 3:
                raise ValueError(f"{l.name}: weight shape {tuple(l.weight.shape)} does not fit ksize {l.ksize}")
        self._tape: List[tuple] = []
        self._ws: Optional[Tensor] = None
        self._half: Dict[int, Tensor] = {}   # data_ptr of an fp32 activation -> its fp16 twin (mixed mode)
        self._zero_page = torch.zeros(4096, dtype=torch.uint8, device=self.device)
        self._zero_bias = torch.zeros(128, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------------------
    def state_dict(self) -> Dict[str, Tensor]:
        """Parameters under the module's key names (what a checkpoint's ``model_state_dict`` holds)."""
        return {k: self._views[k][0].detach().clone() for k in self.param_names}

    def grads(self) -> Dict[str, Tensor]:
        return {k: self._views[k][1] for k in self.param_names}

    def _workspace(self, floats: int) -> Tensor:
        if self._ws is None or self._ws.numel() < floats:
            self._ws = torch.empty(int(floats), dtype=torch.float32, device=self.device)
        return self._ws

    @staticmethod
    def _srcs(srcs) -> "C.Array":
        arr = (_ffi.ConvSrc * len(srcs))()
        for i, (t, up) in enumerate(srcs):
            arr[i].data = t.data_ptr()
            arr[i].affine = None
            arr[i].c = t.shape[-1]
            arr[i].upsample = up
        return arr

    # ------------------------------------------------------------------------------
    # -- mixed precision helpers ------------------------------------------------------------
    def _fast(self, layer: _Layer, srcs) -> bool:
        """The fp16 MFMA kernels take this layer: GroupNorm block, widths 32/64/128, channel counts % 32."""
        return (self.fast16 and layer.norm and layer.cout in (32, 64, 128) and
                all(t.shape[-1] % 32 == 0 for t, _ in srcs))

    def _pack(self, layer: _Layer, transposed: int = 0, c_lo: int = 0, c_n: Optional[int] = None) -> Tensor:
        c_n = layer.cin if c_n is None else c_n
        cout_eff, cin_eff = (c_n, layer.cout) if transposed else (layer.cout, layer.cin)
        buf = torch.empty(layer.ksize ** 3 * (cin_eff // 16) * (cout_eff // 32) * 1024, dtype=torch.uint8, device=self.device)
        _ffi.check(self._L.sk_train_pack_weight(_ffi.ptr(layer.weight), layer.cout, layer.cin, layer.ksize, int(transposed),
                                                 c_lo, c_n, _ffi.ptr(buf), _ffi.stream_ptr(self.device)))
        return buf

    def _to_half(self, t: Tensor, scale: Optional[Tensor] = None) -> Tensor:
        h = torch.empty(t.shape, dtype=self.t16, device=self.device)
        _ffi.check(self._L.sk_train_cast_f32_f16(_ffi.ptr(t), _ffi.ptr(h), t.numel(), _ffi.ptr(scale),
                                                  _ffi.stream_ptr(self.device)))
        return h

    def _fast_conv(self, srcs16: List[Tuple[Tensor, int]], packed: Tensor, bias: Tensor, out_shape, cout: int, ksize: int,
                   partial: Optional[Tensor]) -> Tensor:
        B = srcs16[0][0].shape[0]
        ox, oy, oz = out_shape
        y16 = torch.empty((B, ox, oy, oz, cout), dtype=self.t16, device=self.device)
        _ffi.check(self._L.sk_conv3d(self._srcs(srcs16), len(srcs16), _ffi.ptr(packed), _ffi.ptr(bias), _ffi.ptr(y16), B,
                                      ox, oy, oz, cout, ksize, _ffi.ptr(partial), _ffi.ptr(self._zero_page),
                                      _ffi.stream_ptr(self.device)))
        return y16

    def _h(self, t: Tensor) -> Tensor:
        """fp16 form of an activation: itself, or the registered twin of an fp32 tensor."""
        return t if t.dtype == self.t16 else self._half[t.data_ptr()]

    def _block_mixed(self, layer: _Layer, srcs: List[Tuple[Tensor, int]], out_shape: Tuple[int, int, int],
                     want32: bool = False) -> Tensor:
        """Fast block: keeps the RAW fp16 conv output for the backward and hands the fp16 activation on (an fp32
        copy only when ``want32``: the consumer is an fp32 kernel, i.e. the heads)."""
        B = srcs[0][0].shape[0]
        ox, oy, oz = out_shape
        st = _ffi.stream_ptr(self.device)
        srcs16 = [(self._h(t), up) for t, up in srcs]
        nblk = self._L.sk_conv3d_num_blocks(B, ox, oy, oz, layer.cout, layer.ksize)
        partial = torch.empty((B, nblk, layer.cout // 4, 2), dtype=torch.float32, device=self.device)
        y16 = self._fast_conv(srcs16, self._pack(layer), layer.bias, out_shape, layer.cout, layer.ksize, partial)
        vox = ox * oy * oz
        affine = torch.empty((B, 2, layer.cout), dtype=torch.float32, device=self.device)
        stats = torch.empty((B, GN_GROUPS, 2), dtype=torch.float32, device=self.device)
        _ffi.check(self._L.sk_groupnorm_finalize_stats(_ffi.ptr(partial), B, nblk, GN_GROUPS, layer.cout, vox,
                                                        _ffi.ptr(layer.gamma), _ffi.ptr(layer.beta), GN_EPS,
                                                        _ffi.ptr(affine), _ffi.ptr(stats), st))
        z16 = torch.empty_like(y16)
        z32 = torch.empty(y16.shape, dtype=torch.float32, device=self.device) if want32 else None
        _ffi.check(self._L.sk_train_gn_silu_f16(_ffi.ptr(y16), _ffi.ptr(affine), _ffi.ptr(z16), _ffi.ptr(z32), B, vox,
                                                 layer.cout, st))
        out = z16
        if want32:
            self._half[z32.data_ptr()] = z16
            out = z32
        self._tape.append((layer, srcs, y16, affine, stats, out))
        return out

    def _stem_fast(self, layer: _Layer, srcs, out_shape) -> bool:
        """The stem (Cin = 1, 27 taps, Cout 32) as a fast block: fp16 image operand, exact weights (hi + lo split)."""
        B = srcs[0][0].shape[0]
        return (self.fast16 and layer.norm and layer.cin == 1 and layer.cout == 32 and layer.ksize == 3 and
                len(srcs) == 1 and B <= 32 and out_shape[2] % 2 == 0 and self.fast_stem)

    def _block_stem_mixed(self, layer: _Layer, srcs, out_shape) -> Tensor:
        image = srcs[0][0]                       # (B, X, Y, Z, 1) fp32
        B = image.shape[0]
        X, Y, Z = out_shape
        st = _ffi.stream_ptr(self.device)
        nblk = self._L.sk_conv3d_stem_num_blocks(X, Y, Z)
        partial = torch.empty((B, nblk, 8, 2), dtype=torch.float32, device=self.device)
        wsb = int(self._L.sk_conv3d_stem_workspace_bytes(B, X, Y, Z))
        ws = torch.empty(wsb, dtype=torch.uint8, device=self.device)
        w_t = layer.weight.reshape(32, 27).t().contiguous()   # (27, 32) tap-major, the stem kernel's layout
        y16 = torch.empty((B, X, Y, Z, 32), dtype=self.t16, device=self.device)
        _ffi.check(self._L.sk_train_stem_fwd_f16(_ffi.ptr(image), B, X, Y, Z, _ffi.ptr(w_t), _ffi.ptr(layer.bias),
                                                  _ffi.ptr(y16), _ffi.ptr(partial), _ffi.ptr(ws), wsb, st))
        vox = X * Y * Z
        affine = torch.empty((B, 2, 32), dtype=torch.float32, device=self.device)
        stats = torch.empty((B, GN_GROUPS, 2), dtype=torch.float32, device=self.device)
        _ffi.check(self._L.sk_groupnorm_finalize_stats(_ffi.ptr(partial), B, nblk, GN_GROUPS, 32, vox,
                                                        _ffi.ptr(layer.gamma), _ffi.ptr(layer.beta), GN_EPS,
                                                        _ffi.ptr(affine), _ffi.ptr(stats), st))
        z16 = torch.empty_like(y16)
        _ffi.check(self._L.sk_train_gn_silu_f16(_ffi.ptr(y16), _ffi.ptr(affine), _ffi.ptr(z16), None, B, vox, 32, st))
        self._keep = getattr(self, "_keep", []) + [ws, w_t]   # alive until the stream has consumed them
        self._tape.append((layer, srcs, y16, affine, stats, z16))
        return z16

    def _heads_fast(self, layer: _Layer) -> bool:
        return (self.fast16 and not layer.norm and layer.ksize == 1 and layer.cout == 5 and layer.cin == 32 and self.fast_heads)

    def _block_heads_mixed(self, layer: _Layer, srcs, out_shape) -> Tensor:
        """1x1x1 heads straight on the fp16 activation (an HBM stream: no fp32 copy of the last feature map)."""
        z16 = srcs[0][0]
        B = z16.shape[0]
        ox, oy, oz = out_shape
        logits = torch.empty((B, ox, oy, oz, 5), dtype=torch.float32, device=self.device)
        _ffi.check(self._L.sk_train_heads_fwd_f16(_ffi.ptr(z16), _ffi.ptr(layer.weight), _ffi.ptr(layer.bias),
                                                   _ffi.ptr(logits), B * ox * oy * oz, _ffi.stream_ptr(self.device)))
        self._tape.append((layer, srcs, logits, None, None, logits))
        return logits

    def _block(self, layer: _Layer, srcs: List[Tuple[Tensor, int]], out_shape: Tuple[int, int, int],
               want32: bool = False) -> Tensor:
        if self._stem_fast(layer, srcs, out_shape):
            return self._block_stem_mixed(layer, srcs, out_shape)
        if self._heads_fast(layer) and srcs[0][0].dtype == self.t16:
            return self._block_heads_mixed(layer, srcs, out_shape)
        if self._fast(layer, srcs):
            return self._block_mixed(layer, srcs, out_shape, want32)
        z = self._block_fp32(layer, srcs, out_shape)
        if self.fast16 and layer.norm:
            self._half[z.data_ptr()] = self._to_half(z)   # the stem's output feeds a fast layer
        return z

    def _block_fp32(self, layer: _Layer, srcs: List[Tuple[Tensor, int]], out_shape: Tuple[int, int, int]) -> Tensor:
        B = srcs[0][0].shape[0]
        ox, oy, oz = out_shape
        st = _ffi.stream_ptr(self.device)
        y = torch.empty((B, ox, oy, oz, layer.cout), dtype=torch.float32, device=self.device)
        if not layer.norm:
            _ffi.check(self._L.sk_conv3d_f32(self._srcs(srcs), len(srcs), _ffi.ptr(layer.weight), _ffi.ptr(layer.bias),
                                              _ffi.ptr(y), B, ox, oy, oz, layer.cout, layer.ksize, None, st))
            self._tape.append((layer, srcs, y, None, None, y))
            return y
        nblk = self._L.sk_conv3d_f32_num_blocks(ox, oy, oz)
        partial = torch.empty((B, nblk, layer.cout // 4, 2), dtype=torch.float32, device=self.device)
        _ffi.check(self._L.sk_conv3d_f32(self._srcs(srcs), len(srcs), _ffi.ptr(layer.weight), _ffi.ptr(layer.bias),
                                          _ffi.ptr(y), B, ox, oy, oz, layer.cout, layer.ksize, _ffi.ptr(partial), st))
        vox = ox * oy * oz
        affine = torch.empty((B, 2, layer.cout), dtype=torch.float32, device=self.device)
        stats = torch.empty((B, GN_GROUPS, 2), dtype=torch.float32, device=self.device)
        _ffi.check(self._L.sk_groupnorm_finalize_stats(_ffi.ptr(partial), B, nblk, GN_GROUPS, layer.cout, vox,
                                                        _ffi.ptr(layer.gamma), _ffi.ptr(layer.beta), GN_EPS,
                                                        _ffi.ptr(affine), _ffi.ptr(stats), st))
        z = torch.empty_like(y)
        _ffi.check(self._L.sk_train_gn_silu(_ffi.ptr(y), _ffi.ptr(affine), _ffi.ptr(z), B, vox, layer.cout, st))
        self._tape.append((layer, srcs, y, affine, stats, z))
        return z

    def forward(self, images: Tensor) -> Tensor:
        """images: (B, 1, X, Y, Z) or (B, X, Y, Z), already normalised (the reference's loader does
        it); extents multiples of 4.  Returns the head logits (B, X, Y, Z, 5)."""
        if images.ndim == 5:
            if images.shape[1] != 1:
                raise ValueError("images must have one channel")
            images = images[:, 0]
        x = images.to(self.device, torch.float32).contiguous().unsqueeze(-1)
        _ffi.require_gpu(x, "images")
        _, X, Y, Z, _ = x.shape
        if X % 4 or Y % 4 or Z % 4:
            raise ValueError("crop extents must be multiples of 4 (two stride-2 levels)")
        L0, L1, L2 = (X, Y, Z), (X // 2, Y // 2, Z // 2), (X // 4, Y // 4, Z // 4)
        self._tape = []
        self._half = {}
        self._image = x
        a = x
        for l in self.enc0:
            a = self._block(l, [(a, 0)], L0)
        s0 = a
        a = self._block(self.down0, [(s0, 0)], L1)
        for l in self.enc1:
            a = self._block(l, [(a, 0)], L1)
        s1 = a
        a = self._block(self.down1, [(s1, 0)], L2)
        for l in self.mid:
            a = self._block(l, [(a, 0)], L2)
        r1 = self._block(self.red1, [(a, 0)], L2)
        for i, l in enumerate(self.dec1):
            a = self._block(l, [(s1, 0), (r1, 1)] if i == 0 else [(a, 0)], L1)
        r0 = self._block(self.red0, [(a, 0)], L1)
        for i, l in enumerate(self.dec0):
            a = self._block(l, [(s0, 0), (r0, 1)] if i == 0 else [(a, 0)], L0,
                            want32=(i == len(self.dec0) - 1 and not self._heads_fast(self.heads)))
        return self._block(self.heads, [(a, 0)], L0)

    def backward(self, dlogits: Tensor) -> None:
        """Walk the recorded layers in reverse; gradients of every parameter land in ``flat_grad``."""
        if not self._tape:
            raise RuntimeError("backward() needs a preceding forward()")
        st = _ffi.stream_ptr(self.device)
        grads: Dict[int, object] = {self._tape[-1][5].data_ptr(): dlogits}
        # a tensor read by ONE fast conv and produced by a fast block receives its gradient as the scaled fp16 output of
        # that conv's data-gradient kernel and hands it to the GroupNorm backward as it is: (dx16, scale), no fp32 copy
        n_readers: Dict[int, int] = {}
        for _, srcs_, *_rest in self._tape:
            for t_, _up in srcs_:
                n_readers[t_.data_ptr()] = n_readers.get(t_.data_ptr(), 0) + 1
        fast_out = {e[5].data_ptr() for e in self._tape if e[2].dtype == self.t16}
        k2_read = {t_.data_ptr() for l_, srcs_, y_, *_r in self._tape if l_.ksize == 2 and y_.dtype == self.t16
                   for t_, _u in srcs_}
        for layer, srcs, y, affine, stats, out in reversed(self._tape):
            dz = grads.pop(out.data_ptr())
            B, ox, oy, oz, cout = y.shape
            vox = ox * oy * oz
            fast = y.dtype == self.t16   # recorded by _block_mixed
            ws = None
            dz_scale = None
            if isinstance(dz, tuple):
                final = len(dz) == 3   # (tensor, scale, True): complete, whatever the number of readers (see below)
                dz, dz_scale = dz[0], dz[1]
                assert fast and (final or n_readers.get(out.data_ptr(), 0) == 1), "a pending fp16 gradient was never summed"
            if fast:
                # GroupNorm + SiLU backward straight to the scaled fp16 output gradient (no fp32 dy, no max / cast passes)
                ws = self._workspace(max(self._L.sk_train_gn_bwd_f16_workspace_floats(B, vox, cout),
                                         self._L.sk_train_conv_wgrad_workspace_floats(B, ox, oy, oz, cout, layer.cin, layer.ksize)))
                scale = torch.empty(3, dtype=torch.float32, device=self.device)
                dy16 = torch.empty(y.shape, dtype=self.t16, device=self.device)
                if dz_scale is None:
                    _ffi.check(self._L.sk_train_gn_silu_bwd_f16(_ffi.ptr(dz), _ffi.ptr(y), _ffi.ptr(affine), _ffi.ptr(stats),
                                                                 _ffi.ptr(layer.gamma), B, vox, cout, GN_GROUPS,
                                                                 _ffi.ptr(dy16), _ffi.ptr(scale), _ffi.ptr(layer.g_gamma),
                                                                 _ffi.ptr(layer.g_beta), _ffi.ptr(ws), st))
                else:
                    _ffi.check(self._L.sk_train_gn_silu_bwd_f16h(_ffi.ptr(dz), _ffi.ptr(dz_scale), _ffi.ptr(y),
                                                                  _ffi.ptr(affine), _ffi.ptr(stats), _ffi.ptr(layer.gamma), B,
                                                                  vox, cout, GN_GROUPS, _ffi.ptr(dy16), _ffi.ptr(scale),
                                                                  _ffi.ptr(layer.g_gamma), _ffi.ptr(layer.g_beta),
                                                                  _ffi.ptr(ws), st))
                dy = None
                if layer.cin == 1:   # the stem: taps as the GEMM's N, fp32 image x scaled fp16 dy
                    _ffi.check(self._L.sk_train_stem_wgrad_f16(_ffi.ptr(srcs[0][0]), _ffi.ptr(dy16), _ffi.ptr(scale), B, ox,
                                                                oy, oz, _ffi.ptr(layer.g_weight), _ffi.ptr(layer.g_bias),
                                                                _ffi.ptr(ws), st))
                else:
                    srcs16 = [(self._h(t), up) for t, up in srcs]
                    _ffi.check(self._L.sk_train_conv_wgrad_f16(self._srcs(srcs16), len(srcs16), _ffi.ptr(dy16), _ffi.ptr(scale),
                                                                B, ox, oy, oz, cout, layer.ksize, _ffi.ptr(layer.g_weight),
                                                                _ffi.ptr(layer.g_bias), _ffi.ptr(ws), _ffi.ptr(self._zero_page),
                                                                st))
            else:
                if layer.norm:
                    ws = self._workspace(self._L.sk_train_gn_bwd_workspace_floats(B, vox, cout))
                    _ffi.check(self._L.sk_train_gn_silu_bwd(_ffi.ptr(dz), _ffi.ptr(y), _ffi.ptr(affine), _ffi.ptr(stats),
                                                             _ffi.ptr(layer.gamma), B, vox, cout, GN_GROUPS, _ffi.ptr(dz),
                                                             _ffi.ptr(layer.g_gamma), _ffi.ptr(layer.g_beta), _ffi.ptr(ws), st))
                dy = dz
                if srcs[0][0].dtype == self.t16:   # the heads on the fp16 activation (_block_heads_mixed)
                    nv = B * vox
                    ws = self._workspace(self._L.sk_train_heads_wgrad_workspace_floats(nv))
                    _ffi.check(self._L.sk_train_heads_wgrad_f16(_ffi.ptr(srcs[0][0]), _ffi.ptr(dy), _ffi.ptr(layer.g_weight),
                                                                 _ffi.ptr(layer.g_bias), nv, _ffi.ptr(ws), st))
                else:
                    ws = self._workspace(self._L.sk_train_conv_wgrad_workspace_floats(B, ox, oy, oz, cout, layer.cin,
                                                                                       layer.ksize))
                    _ffi.check(self._L.sk_train_conv_wgrad(self._srcs(srcs), len(srcs), _ffi.ptr(dy), B, ox, oy, oz, cout,
                                                            layer.ksize, _ffi.ptr(layer.g_weight), _ffi.ptr(layer.g_bias),
                                                            _ffi.ptr(ws), st))
            rec = None
            if self.audit is not None and fast:
                rec = {"name": layer.name, "ksize": layer.ksize,
                       "srcs": [((self._h(t) if layer.cin != 1 else t).clone(), up) for t, up in srcs],
                       "y16": y.clone(), "affine": affine.clone(), "stats": stats.clone(),
                       "dz": (dz.clone(), None) if dz_scale is None else (dz.clone(), dz_scale.clone()),
                       "dy16": dy16.clone(), "scale": scale.clone(), "g_weight": layer.g_weight.clone(),
                       "g_bias": layer.g_bias.clone(), "g_gamma": layer.g_gamma.clone(), "g_beta": layer.g_beta.clone(),
                       "weight": layer.weight.clone(), "bias": layer.bias.clone(), "gamma": layer.gamma.clone(),
                       "beta": layer.beta.clone(), "dx16": {}}
                self.audit.append(rec)
            lo = 0
            for t, up in srcs:
                c = t.shape[-1]
                if t.data_ptr() == self._image.data_ptr():
                    lo += c
                    continue  # no gradient w.r.t. the input image
                key = t.data_ptr()
                if fast and c not in (32, 64, 128):
                    raise RuntimeError(f"{layer.name}: mixed precision needs source widths of 32, 64 or 128 channels")
                if layer.ksize == 2 and fast:
                    # stride-2 data gradient: eight pointwise products W_p^T dY on the fast kernel, one per parity of the
                    # fine voxel, then interleaved into the fine grid
                    pend = grads.get(key)
                    pend = pend if isinstance(pend, tuple) else None   # the decoder's fp16 contribution, not yet summed
                    have = key in grads and pend is None
                    # this conv is the tensor's last reader to report (the decoder's contribution, if any, is `pend`): the
                    # sum leaves as a scaled 16-bit tensor + its scale for the producer's GroupNorm backward, no fp32 copy
                    h_out = (self.f16_grad_handoff and not have and key in fast_out and
                             n_readers.get(key, 0) == (2 if pend is not None else 1))
                    if not have and not h_out:
                        grads[key] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
                    packed = self._pack(layer, 2, 0, c)
                    per = packed.numel() // 8
                    t16 = torch.empty((8, B, ox, oy, oz, c), dtype=self.t16, device=self.device)
                    for par in range(8):
                        _ffi.check(self._L.sk_conv3d(self._srcs([(dy16, 0)]), 1, _ffi.ptr(packed[par * per:(par + 1) * per]),
                                                      _ffi.ptr(self._zero_bias), _ffi.ptr(t16[par]), B, ox, oy, oz, c, 1, None,
                                                      _ffi.ptr(self._zero_page), st))
                    if h_out:
                        dx16 = torch.empty(t.shape, dtype=self.t16, device=self.device)
                        oscale = torch.empty(3, dtype=torch.float32, device=self.device)
                        _ffi.check(self._L.sk_train_interleave2_h(_ffi.ptr(t16), _ffi.ptr(pend[0]) if pend is not None else None,
                                                                   _ffi.ptr(pend[1]) if pend is not None else None,
                                                                   _ffi.ptr(dx16), _ffi.ptr(oscale), B, ox, oy, oz, c,
                                                                   _ffi.ptr(scale), st))
                        grads[key] = (dx16, oscale, True)
                    elif pend is not None:
                        _ffi.check(self._L.sk_train_interleave2_add16(_ffi.ptr(t16), _ffi.ptr(pend[0]), _ffi.ptr(pend[1]),
                                                                       _ffi.ptr(grads[key]), B, ox, oy, oz, c, _ffi.ptr(scale),
                                                                       st))
                    else:
                        _ffi.check(self._L.sk_train_interleave2(_ffi.ptr(t16), _ffi.ptr(grads[key]), B, ox, oy, oz, c,
                                                                 _ffi.ptr(scale), int(have), st))
                elif layer.ksize == 2:
                    have = key in grads
                    if not have:
                        grads[key] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
                    _ffi.check(self._L.sk_train_conv_dgrad(_ffi.ptr(dy), _ffi.ptr(layer.weight), _ffi.ptr(grads[key]), B,
                                                            ox, oy, oz, cout, layer.cin, 0, layer.cin, 2, int(have), st))
                elif fast:
                    # data gradient on the fast conv kernel: the layer's weight packed transposed + tap-flipped
                    dx16 = self._fast_conv([(dy16, 0)], self._pack(layer, True, lo, c), self._zero_bias, (ox, oy, oz), c,
                                           layer.ksize, None)
                    if rec is not None:
                        rec["dx16"][lo] = dx16.clone()
                    if up:
                        if key in grads:
                            raise RuntimeError("an upsampled tensor has one consumer in this graph")
                        if self.f16_grad_handoff and key in fast_out and n_readers.get(key, 0) == 1:
                            c16 = torch.empty(t.shape, dtype=self.t16, device=self.device)
                            oscale = torch.empty(3, dtype=torch.float32, device=self.device)
                            _ffi.check(self._L.sk_train_sumpool2_hh(_ffi.ptr(dx16), _ffi.ptr(scale), _ffi.ptr(c16),
                                                                     _ffi.ptr(oscale), B, ox // 2, oy // 2, oz // 2, c, st))
                            grads[key] = (c16, oscale, True)
                        else:
                            grads[key] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
                            _ffi.check(self._L.sk_train_sumpool2_f16(_ffi.ptr(dx16), _ffi.ptr(scale), _ffi.ptr(grads[key]), B,
                                                                      ox // 2, oy // 2, oz // 2, c, st))
                    elif (key in fast_out and key not in grads and self.f16_grad_handoff and
                          (n_readers.get(key, 0) == 1 or (n_readers.get(key, 0) == 2 and key in k2_read))):
                        # single reader: handed to the GroupNorm backward as it is; a skip tensor whose other reader is a
                        # fast stride-2 conv (processed later): summed inside that conv's interleave pass
                        grads[key] = (dx16, scale)
                    else:
                        have = key in grads
                        if not have:
                            grads[key] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
                        _ffi.check(self._L.sk_train_cast_f16_f32(_ffi.ptr(dx16), _ffi.ptr(grads[key]), t.numel(),
                                                                  _ffi.ptr(scale), int(have), st))
                elif up:
                    fine = torch.empty((B, ox, oy, oz, c), dtype=torch.float32, device=self.device)
                    _ffi.check(self._L.sk_train_conv_dgrad(_ffi.ptr(dy), _ffi.ptr(layer.weight), _ffi.ptr(fine), B, ox, oy,
                                                            oz, cout, layer.cin, lo, c, layer.ksize, 0, st))
                    if key in grads:
                        raise RuntimeError("an upsampled tensor has one consumer in this graph")
                    grads[key] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
                    _ffi.check(self._L.sk_train_sumpool2(_ffi.ptr(fine), _ffi.ptr(grads[key]), B, ox // 2, oy // 2,
                                                          oz // 2, c, st))
                elif (self.f16_grad_handoff and t.dtype == self.t16 and layer.ksize == 1 and cout == 5 and len(srcs) == 1 and
                      key in fast_out and key not in grads and n_readers.get(key, 0) == 1 and c % 8 == 0 and 256 % (c // 8) == 0):
                    # the heads on the 16-bit activation: their data gradient leaves as a scaled 16-bit tensor as well
                    dl_scale = torch.zeros(3, dtype=torch.float32, device=self.device)
                    _ffi.check(self._L.sk_train_absmax_scale(_ffi.ptr(dy), dy.numel(), _ffi.ptr(dl_scale), st))
                    dx16 = torch.empty(t.shape, dtype=self.t16, device=self.device)
                    oscale = torch.empty(3, dtype=torch.float32, device=self.device)
                    _ffi.check(self._L.sk_train_heads_dgrad_f16(_ffi.ptr(dy), _ffi.ptr(dl_scale), _ffi.ptr(layer.weight),
                                                                 _ffi.ptr(dx16), _ffi.ptr(oscale), B * vox, c, st))
                    grads[key] = (dx16, oscale, True)
                else:
                    have = key in grads
                    if not have:
                        grads[key] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
                    _ffi.check(self._L.sk_train_conv_dgrad(_ffi.ptr(dy), _ffi.ptr(layer.weight), _ffi.ptr(grads[key]), B,
                                                            ox, oy, oz, cout, layer.cin, lo, c, layer.ksize, int(have), st))
                lo += c
        self._tape = []
        self._half = {}
        self._keep = []


def fused_loss(logits: Tensor, masks: Tensor, skele_masks: Tensor, baked: Tensor, sigma: Sequence[float],
               vector_scale: Sequence[float] = (60, 60, 12),
               loss_params=((0.25, 0.75, 1e-8), (0.5, 0.5, 1e-8), (0.5, 1.5, 1e-8)),
               weights: Sequence[float] = (1.0, 1.0, 1.0), need_grad: bool = True):
    """The three Tversky terms of the step (engine.py:465-493) and d(total)/d(logits) in two passes over
    the logits.  logits (B, X, Y, Z, 5) = head outputs before tanh / sigmoid; masks, skele_masks
    (B, 1, X, Y, Z) (> 0 = foreground); baked (B, 3, X, Y, Z); ``loss_params`` = (alpha, beta, eps) of the
    embedding, probability and skeleton term.  Returns (losses[4] = embed, prob, skeleton, total; dlogits)."""
    _ffi.require_gpu(logits, "logits")
    B, X, Y, Z, five = logits.shape
    if five != 5 or logits.dtype != torch.float32:
        raise ValueError("logits must be fp32 (B, X, Y, Z, 5)")
    n = X * Y * Z
    dev = logits.device
    m = masks.reshape(B, n).to(dev, torch.float32).contiguous()
    sk = skele_masks.reshape(B, n).to(dev, torch.float32).contiguous()
    bk = baked.reshape(B, 3, n).to(dev, torch.float32).contiguous()
    params: List[float] = []
    for (a, b, e), wt in zip(loss_params, weights):
        params += [float(a), float(b), float(e), float(wt)]
    losses = torch.empty(16, dtype=torch.float32, device=dev)
    dl = torch.empty_like(logits) if need_grad else None
    ws = torch.empty(int(_ffi.lib.sk_train_loss_workspace_floats(B, n)), dtype=torch.float32, device=dev)
    _ffi.check(_ffi.lib.sk_train_loss(_ffi.ptr(logits), _ffi.ptr(m), _ffi.ptr(sk), _ffi.ptr(bk), B, X, Y, Z,
                                      _ffi.float_array([float(v) for v in vector_scale]),
                                      _ffi.float_array([float(s) for s in sigma]), _ffi.float_array(params),
                                      _ffi.ptr(losses), _ffi.ptr(dl), _ffi.ptr(ws), _ffi.stream_ptr(dev)))
    return losses[:4], dl


class TrainStep:
    """Optimizer + loss configuration around a :class:`TrainUNet` (engine.py:272-341 of the reference).

    Defaults are the reference's (skoots/config.py:49-64,87,96-101,144): AdamW lr 5e-4, weight decay
    1e-6, betas (0.9, 0.999), eps 1e-8; Tversky (alpha, beta, eps) = embed (0.25, 0.75, 1e-8), probability
    (0.5, 0.5, 1e-8), skeleton (0.5, 1.5, 1e-8); relative weights 1; vector scaling (60, 60, 12)."""

    def __init__(self, model: TrainUNet, lr: float = 5e-4, weight_decay: float = 1e-6, betas=(0.9, 0.999),
                 eps: float = 1e-8, vector_scale=(60, 60, 12),
                 loss_embed=(0.25, 0.75, 1e-8), loss_prob=(0.5, 0.5, 1e-8), loss_skele=(0.5, 1.5, 1e-8),
                 weights=(1.0, 1.0, 1.0), process_group=False):
        """``process_group``: False = single process; None = the default torch.distributed group; or a group."""
        self.model = model
        self.process_group = process_group
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.vector_scale = [float(v) for v in vector_scale]
        self.loss_params = [list(map(float, loss_embed)), list(map(float, loss_prob)), list(map(float, loss_skele))]
        self.weights = [float(w) for w in weights]
        self.exp_avg = torch.zeros_like(model.flat_param)
        self.exp_avg_sq = torch.zeros_like(model.flat_param)
        self.step_count = 0

    def fused_loss(self, logits: Tensor, masks: Tensor, skele_masks: Tensor, baked: Tensor, sigma: Sequence[float],
                   weights: Optional[Sequence[float]] = None, need_grad: bool = True):
        return fused_loss(logits, masks, skele_masks, baked, sigma, self.vector_scale, self.loss_params,
                          self.weights if weights is None else weights, need_grad)

    def sync_gradients(self) -> None:
        """Data-parallel training (engine.py:113-115 wraps the model in DistributedDataParallel): average the
        flat gradient buffer over the ranks -- one all-reduce of ~6.6 MB per step (RCCL on the GPU)."""
        sync_gradients(self.model.flat_grad, self.process_group)

    def optimizer_step(self) -> None:
        self.step_count += 1
        p = self.model.flat_param
        _ffi.check(_ffi.lib.sk_train_adamw(_ffi.ptr(p), _ffi.ptr(self.model.flat_grad), _ffi.ptr(self.exp_avg),
                                           _ffi.ptr(self.exp_avg_sq), p.numel(), self.lr, self.betas[0], self.betas[1],
                                           self.eps, self.weight_decay, self.step_count, _ffi.stream_ptr(p.device)))

    def __call__(self, images: Tensor, masks: Tensor, skele_masks: Tensor, baked: Tensor,
                 sigma: Sequence[float] = (20.0, 20.0, 20.0), weights: Optional[Sequence[float]] = None) -> Tensor:
        """One step (engine.py:456-499).  Returns a device tensor (embed, prob, skeleton, total)."""
        logits = self.model.forward(images)
        losses, dl = self.fused_loss(logits, masks, skele_masks, baked, sigma, weights)
        self.model.backward(dl)
        if self.process_group is not False:
            self.sync_gradients()
        self.optimizer_step()
        return losses

    # -- checkpoint (the payload the reference documents: cfg, model_state_dict, optimizer_state_dict) ------
    def checkpoint(self, cfg: Optional[dict] = None) -> dict:
        return {"cfg": cfg if cfg is not None else {"MODEL": {"DIMS": list(self.model.dims), "DEPTHS": list(self.model.depths),
                                                              "IN_CHANNELS": 1},
                                                    "SKOOTS": {"VECTOR_SCALING": [int(v) for v in self.vector_scale]}},
                "model_state_dict": {k: v.cpu() for k, v in self.model.state_dict().items()},
                "optimizer_state_dict": {"step": self.step_count, "exp_avg": self.exp_avg.cpu(),
                                         "exp_avg_sq": self.exp_avg_sq.cpu(), "lr": self.lr,
                                         "weight_decay": self.weight_decay, "betas": list(self.betas), "eps": self.eps,
                                         "param_names": list(self.model.param_names)}}

    def save(self, path: str, cfg: Optional[dict] = None) -> None:
        """Plain-tensor checkpoint that ``skoots_amd.lib.eval.eval`` loads with ``weights_only=True``."""
        torch.save(self.checkpoint(cfg), path)

    def load_optimizer_state(self, state: dict) -> None:
        if list(state["param_names"]) != list(self.model.param_names):
            raise ValueError("optimizer state belongs to a different parameter layout")
        self.step_count = int(state["step"])
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])


def sync_gradients(flat_grad: Tensor, group=None) -> None:
    """Average a flat gradient buffer over the ranks of ``group`` (sum all-reduce, then 1/world)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("sync_gradients needs an initialised torch.distributed process group")
    world = dist.get_world_size(group)
    if world == 1:
        return
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    flat_grad.mul_(1.0 / world)


def train_step(step: TrainStep, images: Tensor, masks: Tensor, skele_masks: Tensor, baked: Tensor,
               sigma: Sequence[float] = (20.0, 20.0, 20.0)) -> Tensor:
    """Functional spelling of ``TrainStep.__call__``."""
    return step(images, masks, skele_masks, baked, sigma)
