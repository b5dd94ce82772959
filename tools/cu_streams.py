"""CU-masked HIP streams for tools/cu_*_probe.py: an EXPERIMENT of round 4, measured and not adopted.

The idea: the eval step's MFMA-bound convs and its HBM-bound passes on disjoint compute units.

A conv workgroup takes half a CU's registers and LDS, two of them a whole CU: nothing else can co-reside, which is why two
plain streams gained 1.3 %.  But the two kinds of kernel want different things: the passes
(GroupNorm + SiLU, stride-2 convs, stem, heads, scatter) stream HBM and saturate it from a fraction of the chip -- 1.5 TB/s
from 32 CUs, 2.4 from 64 (tools/cu_mask_probe.py) --, the convs want matrix pipes and lose only 4.5 % / 10.8 % of their
rate on 224 / 192 of the 256 CUs (the part is power-limited: fewer active CUs hold a higher clock).  With
``hipExtStreamCreateWithCUMask`` (through ``sk_stream_create_cu_mask``: the stream must belong to the runtime instance the
kernels are launched from) a tile batch's kernels went to a "conv" stream that owns ``32 - h`` CUs of every XCD and an "hbm"
stream that owns the other ``h``, ordered by events, two batches in flight.  Result (32-tile batches, ms per batch): one
stream 22.8; partitioned with h = 4 / 6 / 8: one context 50.5 / 52.0 / 36.3, two contexts 42.9 / 43.0 / 31.8 -- the stem,
the stride-2 convs, the heads and the 1x1x1 convs are not pure HBM streams: on an eighth of the chip they take 5-6x, not
the 3.3x the copy rate suggests, and then THEY are the longer queue.

Mask layout on MI355X (measured, tools/cu_mask_probe.py): bit ``8 c + x`` = compute unit ``c`` (0..31) of XCD ``x`` (0..7);
an XCD whose bits are all clear gets ALL its CUs, so a partition gives every XCD the same split."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch

from skoots_amd import _ffi

XCDS, CUS_PER_XCD = 8, 32


def mask_words(cu_lo: int, cu_hi: int) -> "C.Array":
    """Mask selecting compute units [cu_lo, cu_hi) of every XCD."""
    if not (0 <= cu_lo < cu_hi <= CUS_PER_XCD):
        raise ValueError(f"bad CU range [{cu_lo}, {cu_hi}) per XCD")
    words = [0] * (XCDS * CUS_PER_XCD // 32)
    for c in range(cu_lo, cu_hi):
        for x in range(XCDS):
            b = 8 * c + x
            words[b // 32] |= 1 << (b % 32)
    return (C.c_uint32 * len(words))(*words)


class RoleStreams:
    """One (conv, hbm) stream pair: ``hbm_cus`` compute units of every XCD for the HBM-bound kernels, the rest for the convs."""

    def __init__(self, device, hbm_cus: int):
        self.device = torch.device(device)
        self.hbm_cus = int(hbm_cus)
        self._handles: List[C.c_void_p] = []
        self.streams: Dict[str, torch.cuda.Stream] = {}
        for role, (lo, hi) in (("hbm", (0, self.hbm_cus)), ("conv", (self.hbm_cus, CUS_PER_XCD))):
            h = C.c_void_p()
            w = mask_words(lo, hi)
            with torch.cuda.device(self.device):
                _ffi.check(_ffi.lib.sk_stream_create_cu_mask(w, len(w), C.byref(h)))
            self._handles.append(h)
            self.streams[role] = torch.cuda.ExternalStream(h.value, device=self.device)

    def __getitem__(self, role: str) -> torch.cuda.Stream:
        return self.streams[role]

    def close(self) -> None:
        for s in self.streams.values():
            s.synchronize()
        for h in self._handles:
            _ffi.lib.sk_stream_destroy(h)
        self._handles, self.streams = [], {}
