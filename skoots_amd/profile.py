"""HIP-event timing of named kernel launches, recorded on the stream the kernels are launched on
(``torch.cuda.Event`` sees only torch's current stream, which is the stream every ``sk_*`` call receives).

bench.py turns the spans into roofline entries: ``work`` is the launch's ALGORITHMIC bytes (or FLOPs), from
SURVEY.md section 8(d): gate/dilate/scatter 17 B per interior voxel, labelling 9 B per voxel, follow + assign
64 B per voxel."""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, List, Tuple

import torch

GATE_BYTES_PER_VOXEL = 17.0     # read 5 x 2 B, write 3 x 2 B + 1 B            (eval.py:145-176)
CCL_BYTES_PER_VOXEL = 9.0       # 1 B mask read + 4 B label write + 4 B relabel (flood_fill.py:13-122)
ASSIGN_BYTES_PER_VOXEL = 64.0   # own vector + 9 dependent hops + label gather + write (vector_to_embedding.py:79-132)


class KernelProfile:
    def __init__(self):
        self.spans: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event, float]]] = {}

    @contextmanager
    def span(self, name: str, device, work: float = 0.0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(device))
        try:
            yield
        finally:
            e1.record(torch.cuda.current_stream(device))
            self.spans.setdefault(name, []).append((e0, e1, float(work)))

    def totals(self) -> Dict[str, Tuple[float, float, int]]:
        """{name: (milliseconds, work, launches)}; synchronises on the last event of every name."""
        out = {}
        for name, lst in self.spans.items():
            lst[-1][1].synchronize()
            out[name] = (sum(a.elapsed_time(b) for a, b, _ in lst), sum(w for _, _, w in lst), len(lst))
        return out


@contextmanager
def maybe_span(profile, name: str, device, work: float = 0.0):
    if profile is None:
        yield
    else:
        with profile.span(name, device, work):
            yield
