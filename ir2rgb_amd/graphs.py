"""HIP-graph replay of a shape-static, autograd-free forward (inference / the north-star generator forward).

The generator forward is ~250 kernel launches of 5-130 us each; issued from Python it is host-bound on
one MI355X (host issue ~8 ms per 512x1024 forward against ~7.4 ms of GPU time).  ``GraphedForward``
captures the whole call once into a HIP graph and replays it with one launch: inputs are copied into the
graph's static input buffers, outputs are the graph's own tensors (clone them if they must survive the
next replay).  The reference has no counterpart (models/networks.py forwards are plain eager calls);
numerics are those of the eager path, bit for bit (tests/test_networks_gpu.py).

BatchNorm layers in training mode keep updating their running statistics at every replay (the update is
part of the captured kernels), exactly as repeated eager forwards would.

``two_streams`` (default): the captured forward runs the generators' two independent branches (the two encoders, then
the two decoders) on two HIP streams, which fills the gaps BatchNorm's dependent small kernels leave between the
convolutions (7.07 -> 6.61 ms on the 512x1024 north-star forward); bit-identical to the one-stream order.
"""
import torch

from . import layers


class GraphedForward:
    def __init__(self, fn, *example_inputs, warmup=3, two_streams=True):
        """fn(*tensors) -> tensor | tuple/list of tensors (None entries allowed); every argument a CUDA tensor or None."""
        if not any(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise ValueError("GraphedForward needs CUDA tensors (ir2rgb_amd has no CPU path)")
        self.fn = fn
        self.static_in = [t.clone() if isinstance(t, torch.Tensor) else t for t in example_inputs]
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up off the default stream: packed weights, caches, allocator
                for _ in range(warmup):
                    fn(*self.static_in)
                layers.flush_bn_counters()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # the capture forks the generators' independent branches onto a second stream (ir2rgb_amd.networks.branch_streams)
            from .networks import branch_streams
            with torch.cuda.graph(self.graph), branch_streams(two_streams):
                self.static_out = fn(*self.static_in)
                layers.flush_bn_counters()

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_in):
            raise ValueError("GraphedForward: argument count differs from the captured call")
        for dst, src in zip(self.static_in, inputs):
            if isinstance(dst, torch.Tensor):
                if not isinstance(src, torch.Tensor) or src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("GraphedForward: input shape / dtype differs from the captured call")
                dst.copy_(src)
            elif src is not None:
                raise ValueError("GraphedForward: an argument captured as None must stay None")
        self.graph.replay()
        return self.static_out
