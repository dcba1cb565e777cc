"""hipGraph capture of forwards that run the drop-in modules - how a serving loop drives decode steps.

At decode sizes a Linear forward is one ~8 us kernel (csrc/decode1.hip) behind ~8 us of Python (torch.empty, ctypes).  The
C ABI is stream-ordered and free of host synchronisation (include/lqer_hip.h), and since round 3 the one-launch decode route
is capturable (its granule tag carries the launch's dispatch id), so any callable built from these modules - one Linear, a
decoder layer, a whole token step of a model - can be captured once and replayed:

    step = GraphedCallable(lambda: model(static_ids), warmup=2)     # or GraphedCallable(fn, x_static) with inputs
    step.copy_inputs(new_x); out = step()                            # out is a static tensor: clone it to keep it

The reference has no counterpart (its emulation is eager torch); this is host-side plumbing around torch.cuda.CUDAGraph,
inference only (no autograd), one capture per input shape.
"""
from __future__ import annotations

import torch


class GraphedCallable:
    def __init__(self, fn, *static_inputs: torch.Tensor, warmup: int = 2):
        self.fn = fn
        self.static_inputs = static_inputs
        dev = static_inputs[0].device if static_inputs else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev), torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up off the capture stream: packs operands, sizes the workspaces
                for _ in range(max(warmup, 1)):
                    fn(*static_inputs)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_output = fn(*static_inputs)

    def copy_inputs(self, *new_inputs: torch.Tensor) -> None:
        if len(new_inputs) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} inputs, got {len(new_inputs)}")
        for dst, src in zip(self.static_inputs, new_inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            dst.copy_(src)

    def __call__(self, *new_inputs: torch.Tensor):
        if new_inputs:
            self.copy_inputs(*new_inputs)
        self.graph.replay()
        return self.static_output
