"""The vid2vid inner loop for IR->RGB on MI355X: generator recurrence, reference flow, discriminator
losses, three optimizer steps -- one process per GPU, gradients all-reduced with RCCL.

This is the build's own harness for SURVEY section 8 rows a8, a12, a13, a14 and section 8(e); it
reproduces the semantics of the reference's wrappers without their single-process multi-GPU
placement logic:

    generator recurrence / pyramid      models/generator.py:99-182, :217-235; base_model.py:64-82
    reference flow + confidence         models/flownet.py:20-57
    image / temporal discriminator loss models/discriminator.py:90-200, :236-248, :257-283; models/loss.py:8-41,:105-113
    loop body and optimizer order       train_vid2vid.py:54-111, :166-169

Differences that do not change results: D parameters are frozen during the generator-loss pass
(the reference computes those gradients and discards them with optimizer_d.zero_grad()); the G
gradient all-reduce is overlapped with the D backward; sequences are frame-parallel across ranks.
"""
import contextlib
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import autograd, layers, networks
from . import streamcheck as SC
from .losses import fused_losses
from .optim import FusedAdam
from .ext import warp_diff_norm
from .frames import FrameHistory

DEFAULTS = dict(  # options/base_options.py, options/train_options.py
    input_nc=3, output_nc=3, n_input_gen_frames=3, first_layer_gen_filters=128, gen_network="composite", gen_ds_layers=3,
    gen_blocks=9, n_blocks_local=3, norm="batch", n_scales_spatial=1, fg=False, no_flow=False, n_local_enhancers=1,
    feat_num=3, first_layer_dis_filters=64, num_D=2, n_layers_D=3, no_ganFeat=False, n_frames_D=3, n_scales_temporal=2,
    lr=2e-4, beta1=0.5, lambda_feat=10.0, lambda_T=10.0, lambda_F=10.0, no_first_img=False, max_frames_per_gpu=1,
    n_frames_bp=1, compute_dtype=torch.bfloat16, flownet_dtype=torch.bfloat16,
    no_vgg=True,         # VGG19 perceptual loss off: the pretrained weights cannot be downloaded here (False: ir2rgb_amd.vgg,
                         # randomly initialised unless a torchvision state_dict is loaded into trainer.vgg_loss.vgg)
    shared_fake_forward=True,   # one netD forward on generated frames serves the D and the G loss (autograd.backward_flags)
    resident_inputs=False,      # the window tensors are not written on the main stream (see reference_flows)
    reuse_skipped_flows=True,   # reference flows of temporally skipped frame pairs seen in an earlier window are kept, not recomputed
    allreduce_chunk_elems=32 * 1024 * 1024,   # fp32 elements per gradient all-reduce (128 MB)
    batched_D=True,      # (with shared_fake_forward) real | generated | raw frames go through a discriminator as ONE batch of sample groups
    fused_adam=True,     # one-launch HIP Adam (ir2rgb_amd.optim); False = torch.optim.Adam(foreach=True)
    fused_losses=True,   # grouped HIP loss kernels (ir2rgb_amd.losses); False = the same terms through torch ops
    batched_repack=True,  # all packed weight copies refreshed by one launch after the optimizer steps (layers.WeightRepacker)
    build_flow_net=True,  # False: trainer.flow_net is left None for the caller to set (tests plug a stand-in for FlowNet2)
    branch_streams_fine_scales=True,   # the finer spatial scales' generators run their two branches on two HIP streams, forward
    discriminator_streams=True,        # the image discriminator's scales and the temporal discriminators each on their own HIP stream (forward and, through autograd, backward): independent networks of small layers that fill a fraction of the chip one at a time
    adam_stream=False,                 # the generators' Adam step on its own HIP stream beside the discriminators' backward passes: measured again in round 3 with the discriminators on their own streams, 26.9 ms per window against 26.3 without (round 1: 39.05 vs 38.8) -- off
                                       # and backward (the coarsest scale's kernels are what bench.py brackets: one stream)
)


def avg_pool_pyramid(t, n_scales):
    """[B,T,C,H,W] -> list of n_scales tensors, each AvgPool2d(3,2,1,count_include_pad=False) of the previous
    (base_model.py:64-82)."""
    out = [t]
    for _ in range(1, n_scales):
        b, tt, c, h, w = out[-1].shape
        d = autograd.avg_pool3s2(out[-1].reshape(-1, h, w).unsqueeze(1))
        out.append(d.view(b, tt, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1))
    return out


_GRID_CACHE = {}


def _grid(b, h, w, device, dtype):
    """get_grid once per shape: the lattice is built on the host, so re-creating it costs a synchronous
    host-to-device copy per call (the reference caches it on the module too, base_model.py:131-132)."""
    key = (b, h, w, str(device), dtype)
    if key not in _GRID_CACHE:
        _GRID_CACHE[key] = networks.get_grid(b, h, w, device=device, dtype=dtype)
    return _GRID_CACHE[key]


def resample(image, flow):
    """Model.resample (base_model.py:129-136): same align_corners mismatch as the generator's warp."""
    b, c, h, w = image.shape
    grid = _grid(b, h, w, flow.device, flow.dtype)
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    return F.grid_sample(image, (grid + fl).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=False)


def masked_l1(a, b, mask):
    m = mask.expand(-1, a.size(1), -1, -1)
    return F.l1_loss(a * m, b * m)


def gan_loss(pred, real):
    """GANLoss with gan_mode 'ls' -> MSE against a constant target, summed over the D scales (loss.py:8-41)."""
    total = 0
    for scale in pred:
        p = scale[-1].float()
        total = total + F.mse_loss(p, torch.full_like(p, 1.0 if real else 0.0))
    return total


class _SplitGroupsFn(torch.autograd.Function):
    """t [G * n, ...] -> G views [n, ...]; backward: the pieces' gradients in ONE buffer of t's layout (autograd's own
    slicing would allocate and fill a full-size zero tensor per piece and add them up; materialised zero gradients would
    also arrive NCHW-contiguous and drag the whole buffer out of NHWC).  The buffer exists from the forward on and its
    slices are registered as the gradient destinations of the pieces (losses.GRAD_DST): a fused loss kernel that
    differentiates a piece writes straight into it, anything else is copied in; a piece without gradient is zero-filled
    unless its sample group is inactive in this pass (``owner._ir2rgb_active``, autograd.backward_flags: nobody reads it)."""

    @staticmethod
    def forward(ctx, t, G, owner):
        from . import losses
        n = t.shape[0] // G
        ctx.set_materialize_grads(False)
        ctx.G, ctx.n, ctx.owner = G, n, owner
        fmt = torch.channels_last if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) else torch.contiguous_format
        ctx.buf = torch.empty(tuple(t.shape), dtype=t.dtype, device=t.device, memory_format=fmt)
        pieces = tuple(t[g * n:(g + 1) * n] for g in range(G))
        for g, piece in enumerate(pieces):
            losses.GRAD_DST[losses._dst_key(piece)] = ctx.buf[g * n:(g + 1) * n]
        return pieces

    @staticmethod
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return None, None, None
        out, n = ctx.buf, ctx.n
        active = getattr(ctx.owner, "_ir2rgb_active", None) if ctx.owner is not None else None
        for i, g in enumerate(grads):
            dst = out[i * n:(i + 1) * n]
            if g is None:
                if active is None or i < active:
                    dst.zero_()
            elif g.data_ptr() != dst.data_ptr():
                dst.copy_(g)
        return out, None, None


def split_groups(t, G, owner=None):
    """``owner``: a convolution module of the network that produced ``t`` (carries the pass's active-group flag)."""
    return _SplitGroupsFn.apply(t, G, owner)


@contextlib.contextmanager
def frozen(module):
    ps = [p for p in module.parameters() if p.requires_grad]
    for p in ps:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in ps:
            p.requires_grad_(True)


class FlowNet(torch.nn.Module):
    """Frozen FlowNet2 + confidence mask (models/flownet.py)."""

    def __init__(self, conv_dtype=torch.bfloat16, seed=1, use_graph=None):
        super().__init__()
        from .flownet2_pytorch.models import FlowNet2
        import os
        self.use_graph = (os.environ.get("IR2RGB_FLOWNET_GRAPH", "1") != "0") if use_graph is None else bool(use_graph)
        self._graphs = {}   # shape key -> call count | (graph, in1, in2, (flow, conf)) | False (capture failed)
        rng = torch.random.get_rng_state()
        torch.manual_seed(seed)  # no checkpoint offline: the reference's own init (models.py:68-77)
        self.flowNet = FlowNet2(conv_dtype=conv_dtype)
        torch.random.set_rng_state(rng)
        self.flowNet.eval()
        for p in self.flowNet.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def forward(self, input_A, input_B, side=None):
        """``side``: a HIP stream a graph REPLAY of this call may run on (see Vid2VidTrainer.train_window); calls that
        still have lazy work to do (eager warm-up, capture) stay on the current stream.  ``self.ran_on`` tells which."""
        if input_A.dim() == 5:
            b, n, c, h, w = input_A.shape
            flow, conf = self.compute_flow_and_conf(input_A.reshape(-1, c, h, w), input_B.reshape(-1, c, h, w), side)
            return flow.view(b, n, 2, h, w), conf.view(b, n, 1, h, w)
        return self.compute_flow_and_conf(input_A, input_B, side)

    def will_replay(self, n, im):
        """True when a call on ``n`` frame pairs shaped like ``im`` [., 3, H, W] would be a graph replay (no lazy work)."""
        return isinstance(self._graphs.get(((n,) + tuple(im.shape[1:]), im.dtype, str(im.device))), tuple)

    def compute_flow_and_conf(self, im1, im2, side=None):
        """FlowNet2 is frozen, runs without autograd and with fixed shapes: ~330 small launches per call.
        After two eager calls at a shape the whole call (convolutions, operators, interpolations, the
        confidence mask) is captured into a HIP graph and replayed -- one launch, no host work between
        the kernels.  Any failure to capture falls back to the eager path for that shape (logged once)."""
        self.ran_on = None
        key = (tuple(im1.shape), im1.dtype, str(im1.device))
        ent = self._graphs.get(key)
        if not self.use_graph or not im1.is_cuda or ent is False:
            return self._flow_and_conf_eager(im1, im2)
        if ent is None or isinstance(ent, int):
            n = (ent or 0) + 1
            self._graphs[key] = n
            if n <= 2:
                return self._flow_and_conf_eager(im1, im2)
            try:
                a, b = im1.clone(), im2.clone()
                torch.cuda.synchronize(im1.device)
                g = torch.cuda.CUDAGraph()
                # (with a process group alive its watchdog thread polls events meanwhile: only this thread's calls are
                # subject to the capture rules then)
                mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
                with torch.cuda.graph(g, capture_error_mode=mode):
                    out = self._flow_and_conf_eager(a, b)
                ent = self._graphs[key] = (g, a, b, out)
            except Exception as e:  # noqa: BLE001  capture is an optimisation, never a requirement
                self._graphs[key] = False
                print(f"[ir2rgb_amd] FlowNet2 graph capture failed at {key[0]} ({type(e).__name__}: {e}); staying eager",
                      flush=True)
                return self._flow_and_conf_eager(im1, im2)
        g, a, b, (flow, conf) = ent
        if side is not None and torch.cuda.current_stream(im1.device) == side:
            # the caller already works on the second stream (Vid2VidTrainer.reference_flows, resident inputs): no wait
            a.copy_(im1)
            b.copy_(im2)
            g.replay()
            self.ran_on = side
            out = flow.clone(), conf.clone()
            if SC.ENABLED:
                SC.produced(out[0], "reference flow (FlowNet2 replay)"), SC.produced(out[1], "flow confidence (FlowNet2 replay)")
            return out
        if side is not None:
            main = torch.cuda.current_stream(im1.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                a.copy_(im1)
                b.copy_(im2)
                g.replay()
                out = flow.clone(), conf.clone()
                if SC.ENABLED:
                    SC.produced(out[0], "reference flow (FlowNet2 replay)"), SC.produced(out[1], "flow confidence (FlowNet2 replay)")
            for t in (im1, im2):
                t.record_stream(side)
            self.ran_on = side
            return out
        a.copy_(im1)
        b.copy_(im2)
        g.replay()
        return flow.clone(), conf.clone()   # the graph owns its outputs: the next replay overwrites them

    def _flow_and_conf_eager(self, im1, im2):
        assert im1.size(1) == 3 and im1.shape == im2.shape
        old_h, old_w = im1.shape[2:]
        new_h, new_w = old_h // 64 * 64, old_w // 64 * 64
        resize = old_h != new_h      # flownet.py:42 tests the height only ...
        if not resize and old_w != new_w:
            # ... and with a width that is not a multiple of 64 the reference dies in FlowNet2's torch.cat
            raise ValueError(f"FlowNet: width {old_w} is not a multiple of 64 while height {old_h} is "
                             "(the reference resizes only when the height is off, flownet.py:42)")
        if resize:
            im1 = F.interpolate(im1, size=(new_h, new_w), mode="bilinear")
            im2 = F.interpolate(im2, size=(new_h, new_w), mode="bilinear")
        flow = self.flowNet(torch.stack([im1, im2], dim=2)).float().contiguous()
        _, _, norm = warp_diff_norm(im1.float().contiguous(), im2.float().contiguous(), flow, want_warped=False,
                                    want_diff=False)
        conf = (norm * norm < 0.02).float()  # flownet.py:50,56-57: sum of squares < 0.02
        if resize:
            flow = F.interpolate(flow, size=(old_h, old_w), mode="bilinear") * old_h / new_h
            conf = F.interpolate(conf, size=(old_h, old_w), mode="bilinear")
        return flow, conf


class FlatGrads:
    """One flat fp32 gradient buffer per optimizer.  ``zero`` drops the .grad references, so the first
    contribution of a backward pass is adopted by autograd without an add kernel per parameter;
    ``all_reduce_async`` makes every .grad a view of the flat buffer and, for world > 1, issues a chunked
    RCCL all-reduce (~128 MB per collective) that AVERAGES (ReduceOp.AVG: no scaling pass over the buffer).
    A parameter that received no gradient gets a zero one, as the reference's zero_grad() + Adam step would
    see (train_vid2vid.py:93-105).

    ``direct=True`` (world > 1, every parameter used once per backward pass -- the generators): the convolutions
    write their weight gradients straight into their slices (ir2rgb_amd.autograd.GRAD_SINKS), so 99.9 % of the
    buffer is in place when the pass ends; the rest (biases, BatchNorm parameters, first / thin / padded layers) is
    gathered by one multi-tensor copy, as everything is when ``direct`` is off (the discriminators: several
    contributions per parameter and pass, summed by the autograd engine before they are adopted).  The in-place
    weights sit at the front of the buffer in parameter order, cut into chunks; a chunk goes onto the wire from the
    autograd hook of the parameter that completes it, i.e. WHILE the backward pass is still running (the generators'
    1.4 GB of residual-block gradients are produced over the last ~5 ms of their pass: the all-reduce then ends about
    when the pass does instead of starting there); ``all_reduce_async`` sends what is left."""

    def __init__(self, params, chunk_elems=32 * 1024 * 1024, world=1, direct=False):
        self.params = [p for p in params if p.requires_grad]
        pad4 = lambda k: (k + 3) & ~3  # noqa: E731  every view starts on a 16-byte boundary (vector path of adam_kernel)
        n = sum(pad4(p.numel()) for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.direct = bool(direct and world > 1)
        # layout: with ``direct`` the convolution weights (written in place by their weight-gradient kernels) come first,
        # in parameter order, so that whole chunks of the buffer are complete -- and can be all-reduced -- while the
        # backward pass is still running; biases / BatchNorm parameters (gathered by one copy when the pass ends) follow
        is_sink = [self.direct and p.dim() == 4 for p in self.params]
        order = [i for i, s in enumerate(is_sink) if s] + [i for i, s in enumerate(is_sink) if not s]
        self.views, off = [None] * len(self.params), 0
        starts = {}
        for i in order:
            p = self.params[i]
            starts[i] = off
            self.views[i] = self.flat[off:off + p.numel()].view_as(p)
            off += pad4(p.numel())
        self.chunk = chunk_elems
        self.handles = []
        self.scale_after = None
        # early chunks: [lo, hi) ranges of the sink region, each a list of parameter indices; a chunk is all-reduced from
        # the autograd hook of the parameter whose gradient completes it (all_reduce_async picks up what is left)
        self.chunks, self._fired, self._pending, self._issued, self._world = [], set(), [], [], world
        self._direct_capable, self._sink_ids = self.direct, []
        if self.direct:
            cur, lo = [], 0
            sink_ids = self._sink_ids = [i for i in order if is_sink[i]]
            for k, i in enumerate(sink_ids):
                cur.append(i)
                hi = starts[i] + pad4(self.params[i].numel())
                if hi - lo >= chunk_elems or k == len(sink_ids) - 1:
                    self.chunks.append((lo, hi, tuple(cur)))
                    cur, lo = [], hi
            self.sink_end = self.chunks[-1][1] if self.chunks else 0
            chunk_of = {i: c for c, (_, _, ids) in enumerate(self.chunks) for i in ids}
            for i in sink_ids:
                p = self.params[i]
                autograd.GRAD_SINKS[p] = self.views[i]
                p.register_post_accumulate_grad_hook(self._make_hook(i, chunk_of[i]))
        else:
            self.sink_end = 0

    def set_direct(self, on):
        """Arm / disarm the in-place sinks for the backward passes to come.  They are only sound while every parameter
        receives ONE contribution per pass: a window that generates several frames applies each generator several times,
        and a second contribution would overwrite the first in the same slice (autograd then sums two aliases of it) --
        the trainer switches to the gathered form for such windows (Vid2VidTrainer.generate)."""
        on = bool(on) and self._direct_capable
        if on == self.direct:
            return
        self.direct = on
        for i in self._sink_ids:
            if on:
                autograd.GRAD_SINKS[self.params[i]] = self.views[i]
            else:
                autograd.GRAD_SINKS.pop(self.params[i], None)
        self._pending, self._issued = [], []

    def _make_hook(self, i, c):
        def hook(p):
            if not self._pending or i in self._fired:
                return
            self._fired.add(i)
            if p.grad is None or p.grad.data_ptr() != self.views[i].data_ptr():
                self._pending[c] = -1                      # this gradient is not in place: the chunk waits for the gather
                return
            if self._pending[c] > 0:
                self._pending[c] -= 1
                if self._pending[c] == 0:
                    lo, hi, _ = self.chunks[c]
                    self._reduce(lo, hi)
                    self._issued[c] = True
        return hook

    def _reduce(self, lo, hi):
        avg = dist.get_backend() == "nccl"     # RCCL averages in the collective; gloo (CPU tests) has no AVG
        self.scale_after = None if avg else 1.0 / self._world
        for i in range(lo, hi, self.chunk):
            self.handles.append(dist.all_reduce(self.flat[i:min(i + self.chunk, hi)], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM,
                                                async_op=True))

    def zero(self):
        for p in self.params:
            p.grad = None
        if self.direct:      # arm the early chunks for the backward pass that follows
            self._fired = set()
            self._pending = [len(ids) for _, _, ids in self.chunks]
            self._issued = [False] * len(self.chunks)

    def all_reduce_async(self, world):
        self._world = world
        src, dst, missing = [], [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                missing.append(v)
                p.grad = v
            elif world > 1 and p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
                p.grad = v
            elif world > 1:
                p.grad = v              # written in place by its convolution (GRAD_SINKS)
        issued, self._pending = self._issued, []           # (disarm the hooks)
        if missing:
            if any(issued):
                lo_hi = [(lo, hi) for (lo, hi, _), done in zip(self.chunks, issued) if done]
                base = self.flat.data_ptr()
                for v in missing:       # a parameter without gradient inside a chunk that is already on the wire cannot happen:
                    o = (v.data_ptr() - base) // 4          # its chunk never completes
                    assert not any(lo <= o < hi for lo, hi in lo_hi), "FlatGrads: early chunk reduced before it was complete"
            torch._foreach_zero_(missing)      # one multi-tensor launch instead of one fill per parameter
        if world <= 1:
            return
        if src:
            torch._foreach_copy_(dst, src)
        # what the hooks have not sent: unfinished chunks of the sink region (merged into runs), then the gathered tail
        run = None
        for (lo, hi, _), done in zip(self.chunks, issued or [False] * len(self.chunks)):
            if done:
                if run is not None:
                    self._reduce(*run)
                    run = None
            else:
                run = (lo, hi) if run is None else (run[0], hi)
        tail_lo = self.sink_end
        if run is not None:
            tail_lo = run[0]
        self._reduce(tail_lo, self.flat.numel())
        self._issued = []

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.scale_after is not None:
            self.flat.mul_(self.scale_after)
            self.scale_after = None


# IR2RGB_D_T_SLOTS=own (default): one stream per temporal discriminator; shared: both on one (measured slower: 27.4 vs
# 26.2-27.0 ms per window on the same box)
_T_SLOTS = os.environ.get("IR2RGB_D_T_SLOTS", "own")


class _LossDict(dict):
    """A dict of named loss terms that may also carry them as the vectors the fused loss kernels wrote
    (``vecs`` = (image-term vector | None, discriminator-term vector, generator-term vector), see get_losses)."""
    vecs = None


class Vid2VidTrainer:
    """One rank's models, optimizers and per-sequence state; ``train_window`` is the loop body."""

    def __init__(self, device, world_size=1, seed=0, **overrides):
        o = dict(DEFAULTS)
        o.update(overrides)
        self.opt, self.device, self.world = o, device, world_size
        tG = o["n_input_gen_frames"]
        self.n_scales, self.t_scales, self.tD = o["n_scales_spatial"], o["n_scales_temporal"], o["n_frames_D"]
        g_in, g_prev = o["input_nc"] * tG, (tG - 1) * o["output_nc"]
        kw = {k: o[k] for k in ("gen_blocks", "n_local_enhancers", "feat_num", "n_blocks_local", "fg", "no_flow")}
        torch.manual_seed(seed)  # identical initial weights on every rank
        self.netG = [networks.build_generator_module(g_in, o["output_nc"], g_prev, o["first_layer_gen_filters"],
                                                      o["gen_network"], o["gen_ds_layers"], o["norm"], 0, **kw)]
        for s in range(1, self.n_scales):
            self.netG.append(networks.build_generator_module(g_in, o["output_nc"], g_prev, o["first_layer_gen_filters"] // 2 ** s,
                                                             o["gen_network"] + "-local", o["gen_ds_layers"], o["norm"], s, **kw))
        self.netD = networks.build_discriminator_module(o["input_nc"] + o["output_nc"], o["first_layer_dis_filters"],
                                                        o["n_layers_D"], o["norm"], o["num_D"], not o["no_ganFeat"])
        dt_in = o["output_nc"] * self.tD + 2 * (self.tD - 1)
        self.netD_T = [networks.build_discriminator_module(dt_in, o["first_layer_dis_filters"], o["n_layers_D"], o["norm"],
                                                           o["num_D"], not o["no_ganFeat"]) for _ in range(self.t_scales)]
        for m in self.netG + [self.netD] + self.netD_T:
            m.to(device).train()
            m.compute_dtype = o["compute_dtype"]
        import os
        # gloo (the CPU-side rehearsal backend) moves CUDA tensors through the host and synchronises the device: next to it a
        # second stream under autograd is pathological (the two-rank rehearsal of bench.py: 7.9 s per window with the
        # finer-scale generators' branches on two streams, 0.39 s without; the discriminators' streams cost 30 ms there).
        # RCCL ranks keep all streams (tests/test_rccl_gpu.py).
        gloo = world_size > 1 and dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl"
        fine = os.environ.get("IR2RGB_BRANCH_FINE", "0" if gloo else "1") != "0"
        for g in self.netG[1:]:
            g.branch_streams_training = bool(o["branch_streams_fine_scales"]) and fine
        self.d_streams = bool(o["discriminator_streams"]) and os.environ.get("IR2RGB_D_STREAMS", "1") != "0" and device.type == "cuda"
        self.adam_stream_on = os.environ.get("IR2RGB_ADAM_STREAM", "1" if o["adam_stream"] else "0") != "0" and device.type == "cuda"
        self._adam_stream, self._adam_pending = None, False
        self.netD.scale_streams = self.d_streams
        # FlowNet2 is replayed from a HIP graph in every configuration (a capture next to a process group runs in
        # thread-local mode, FlowNet.compute_flow_and_conf), on its own stream -- also for data-parallel ranks on RCCL
        # (tests/test_rccl_gpu.py: bit-identical to the single-process trainer next to RCCL's collectives).  Only the gloo
        # rehearsal of bench.py keeps it on the main stream: gloo moves CUDA tensors through the host and synchronises
        # (7.6 s per window measured with the replay on a second stream, 0.43 s on the main one).  IR2RGB_FLOW_STREAM_DP=0/1
        # overrides.
        self.flow_net = None
        if o["build_flow_net"]:
            self.flow_net = FlowNet(o["flownet_dtype"], use_graph=None).to(device)
        self._side_wgrad = None          # set per window in generate(): safe only when n_load == 1
        self.vgg_loss = None
        if not o["no_vgg"]:
            from .vgg import VGGLoss
            self.vgg_loss = VGGLoss().to(device)
            self.vgg_loss.vgg.compute_dtype = o["compute_dtype"]

        g_params = [p for g in self.netG for p in g.parameters()]  # niter_fix_global = 0: all scales train
        # NOT fused=True: torch's fused Adam updates the parameters without bumping their version counters,
        # and the packed MFMA weights (ir2rgb_amd.layers.packed_weight) are refreshed on a version change
        adam = dict(lr=o["lr"], betas=(o["beta1"], 0.999), foreach=True)
        # (generators: one use per parameter and pass when one frame is generated per window, see generate())
        self.grads_G = FlatGrads(g_params, chunk_elems=o["allreduce_chunk_elems"], world=world_size, direct=o["max_frames_per_gpu"] == 1)
        self.grads_D = FlatGrads(self.netD.parameters())
        self.grads_DT = [FlatGrads(d.parameters()) for d in self.netD_T]
        if o["fused_adam"]:
            make = lambda ps: FusedAdam(ps, lr=o["lr"], betas=(o["beta1"], 0.999))  # noqa: E731
        else:
            make = lambda ps: torch.optim.Adam(ps, **adam)  # noqa: E731
        self.optimizer_G = make(self.grads_G.params)
        self.optimizer_D = make(self.grads_D.params)
        self.optimizer_D_T = [make(g.params) for g in self.grads_DT]
        self.repacker = layers.WeightRepacker(list(self.netG) + [self.netD] + list(self.netD_T)) if o["batched_repack"] else None
        self.reset_sequence()

    def _flow_stream(self, t):
        """The side stream FlowNet2 runs on (None: same stream as everything else; IR2RGB_FLOW_STREAM=0 or CPU tensors)."""
        import os
        if not t.is_cuda or os.environ.get("IR2RGB_FLOW_STREAM", "1") == "0":
            return None
        if self.world > 1:
            dp = os.environ.get("IR2RGB_FLOW_STREAM_DP")
            if dp is None:
                dp = "1" if (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl") else "0"
            if dp != "1":
                return None
        if not isinstance(self.flow_net, FlowNet):
            return None
        if getattr(self, "_flow_side", None) is None:
            self._flow_side = torch.cuda.Stream(t.device)
        return self._flow_side

    # ------------------------------------------------------------------ per-sequence state
    def reset_sequence(self):
        self.fake_B_prev = None          # pyramid of the last tG-1 generated frames
        self._pair_flows = {}            # temporal scale -> [(push count, flow, conf)] of its newest pairs (reference_flows)
        self._early_on = False           # FlowNet2 ahead of the main stream (reference_flows)
        self._backward_done = None       # event: the previous window's backward passes are through
        # histories of the four streams the temporal discriminators sub-sample (train_vid2vid.py:45-52: real_B_all,
        # fake_B_all, flow_ref_all, conf_ref_all), each in one preallocated device buffer (ir2rgb_amd.frames)
        ts, tD = self.t_scales, self.tD
        self.hist = {"real": FrameHistory(ts, tD), "fake": FrameHistory(ts, tD), "flow": FrameHistory(1, tD),
                     "conf": FrameHistory(1, tD)}

    # ------------------------------------------------------------------ generator (a8)
    def generate(self, real_A_all, real_B_all):
        """One window: [B, n_frames_load + tG - 1, C, H, W] inputs -> the reference's 7-tuple
        (generator.py:99-123)."""
        tG, ns = self.opt["n_input_gen_frames"], self.n_scales
        n_load = real_A_all.size(1) - tG + 1
        if self._side_wgrad != (n_load == 1):
            # every generator is applied once per backward pass when one frame is generated per window:
            # only then may its weight gradients run on the side stream (ir2rgb_amd.autograd) ...
            self._side_wgrad = n_load == 1
            for g in self.netG:
                autograd.enable_side_wgrad(g, self._side_wgrad)
        # ... and only then may they be written straight into the all-reduce buffer (FlatGrads.set_direct)
        self.grads_G.set_direct(n_load == 1)
        first = self.fake_B_prev is None
        if not first:
            fake_pyr = list(self.fake_B_prev)
        elif self.opt["no_first_img"]:       # the model also generates the first frame (generator.py:219-220)
            fake_pyr = avg_pool_pyramid(torch.zeros_like(real_B_all[:, :tG - 1]), ns)
        else:                                # training: the first frames are given (generator.py:221-222)
            fake_pyr = avg_pool_pyramid(real_B_all[:, :tG - 1], ns)
        A_pyr = avg_pool_pyramid(real_A_all, ns)
        fake_raw, flows, weights = [], [], []
        for t in range(n_load):
            feat = flow_feat = None
            for s in range(ns):                      # coarse to fine
                si = ns - 1 - s
                As = A_pyr[si]
                b, _, _, h, w = As.shape
                A_in = As[:, t:t + tG].reshape(b, -1, h, w)
                prev = fake_pyr[si][:, t:t + tG - 1]
                if t % self.opt["n_frames_bp"] == 0:
                    prev = prev.detach()
                out = self.netG[s](A_in, prev.reshape(b, -1, h, w), None, feat, flow_feat, None,
                                   self.opt["no_first_img"] and first)
                fake_B, flow, weight, raw, feat, flow_feat, _ = out
                fake_pyr[si] = torch.cat([fake_pyr[si], fake_B.unsqueeze(1)], 1)
                if s == ns - 1:
                    fake_raw.append(raw.unsqueeze(1))
                    flows.append(flow.unsqueeze(1))
                    weights.append(weight.unsqueeze(1))
        self.fake_B_prev = [B[:, -tG + 1:].detach() for B in fake_pyr]
        fake_B = fake_pyr[0][:, tG - 1:]
        one = lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, 1)  # noqa: E731  (one frame per window: no copy)
        return (fake_B, one(fake_raw), one(flows), one(weights), real_A_all[:, tG - 1:], real_B_all[:, tG - 2:])

    # ------------------------------------------------------------------ discriminator losses (a13)
    def _gan_and_fm(self, pred_real, pred_fake):
        o = self.opt
        fw, dw = 4.0 / (o["n_layers_D"] + 1), 1.0 / o["num_D"]
        if o["fused_losses"]:
            return self._fused_D_terms(pred_real, [pred_fake])[1].unbind(0)
        loss_gan = gan_loss(pred_fake, True)
        loss_fm = torch.zeros_like(loss_gan)
        if not o["no_ganFeat"]:
            for i in range(min(len(pred_fake), o["num_D"])):
                for j in range(len(pred_fake[i]) - 1):
                    loss_fm = loss_fm + dw * fw * (pred_fake[i][j] - pred_real[i][j].detach()).abs().mean(dtype=torch.float32) * o["lambda_feat"]
        return loss_gan, loss_fm

    def _fused_D_terms(self, pred_real, pred_fakes):
        """compute_loss_D (discriminator.py:154-166, :186-200) for every generated input in ``pred_fakes`` against the same
        real forward, as TWO loss launches: the discriminator-side vector [D_real, D_fake] and the generator-side vector
        [G_GAN, G_GAN_Feat], each already summed over the inputs.  (Two groups, not one: a backward pass then differentiates
        only the group its total depends on, and every prediction tensor receives one gradient per pass.)  The real
        logits appear in one term of weight len(pred_fakes): the reference evaluates that term once per compute_loss_D call
        on identical activations -- two terms on one tensor would write one gradient destination twice (losses.GRAD_DST)."""
        o = self.opt
        fw, dw = 4.0 / (o["n_layers_D"] + 1), 1.0 / o["num_D"]
        k = float(len(pred_fakes))
        d_terms = [("mse", scale[-1], 1.0, k, 0) for scale in pred_real]
        g_terms = []
        for pf in pred_fakes:
            d_terms += [("mse", scale[-1], 0.0, 1.0, 1) for scale in pf]
            g_terms += [("mse", scale[-1], 1.0, 1.0, 0) for scale in pf]
            if not o["no_ganFeat"]:
                for i in range(min(len(pf), o["num_D"])):
                    for j in range(len(pf[i]) - 1):
                        g_terms.append(("l1", pf[i][j], pred_real[i][j], dw * fw * o["lambda_feat"], 1))
        return fused_losses(d_terms, 2, o["compute_dtype"]), fused_losses(g_terms, 2, o["compute_dtype"])

    def _loss_D(self, netD, real_in, fake_in, pred_real=None, pred_fake=None):
        """Three forwards exactly as compute_loss_D (discriminator.py:154-166).  ``pred_real`` / ``pred_fake``: the results
        of ``netD(real_in)`` / ``netD(fake_in)`` when the caller already holds them (see image_losses, _batched_D)."""
        if pred_real is None:
            pred_real = netD(real_in)
        shared = self.opt["shared_fake_forward"]
        if pred_fake is not None:
            pred_fake_d = pred_fake
        elif shared:
            with layers.repeated_forward(2):
                pred_fake_d = pred_fake = netD(fake_in)
        else:
            pred_fake_d = netD(fake_in.detach())
        if self.opt["fused_losses"] and pred_fake is not None:
            dvec, gvec = self._fused_D_terms(pred_real, [pred_fake])
            return tuple(dvec.unbind(0)) + tuple(gvec.unbind(0))
        if self.opt["fused_losses"]:
            out = fused_losses([("mse", scale[-1], 1.0, 1.0, 0) for scale in pred_real] +
                               [("mse", scale[-1], 0.0, 1.0, 1) for scale in pred_fake_d], 2, self.opt["compute_dtype"])
            loss_D_real, loss_D_fake = out[0], out[1]
        else:
            loss_D_real, loss_D_fake = gan_loss(pred_real, True), gan_loss(pred_fake_d, False)
        if pred_fake is None:
            with frozen(netD):
                pred_fake = netD(fake_in)
        loss_G_GAN, loss_G_FM = self._gan_and_fm(pred_real, pred_fake)
        return loss_D_real, loss_D_fake, loss_G_GAN, loss_G_FM

    def _batched_D(self, netD, inputs, repeats, order=None):
        """``netD`` on several inputs as ONE batch of sample groups (MultiScaleDiscriminator.forward): every convolution
        and every weight gradient runs once per layer instead of once per input, BatchNorm sees each input as the
        separate forward it is in the reference (``repeats[g]``: how many reference forwards group g stands for, i.e.
        how often its batch statistics enter the running statistics; ``order``: the order the groups' statistics enter
        them in).  Callers put the generated frames FIRST: the generator's backward pass then works on the leading
        groups only (autograd.backward_flags(active_groups=...)).  -> one prediction pyramid per input."""
        from . import losses as _losses
        G = len(inputs)
        # gradient destinations registered by this network's previous forward are dropped here: an entry lives from one
        # forward of a network to its next, whoever the caller is (losses.GRAD_DST is keyed by address)
        keys = self.__dict__.setdefault("_dst_keys", {})
        for k in keys.pop(id(netD), ()):
            _losses.GRAD_DST.pop(k, None)
        before = set(_losses.GRAD_DST)
        with layers.repeated_forward(tuple(repeats)):
            out = netD(torch.cat(inputs, 0), sample_groups=G, group_order=order)
        owner = next(m for m in netD.modules() if isinstance(m, torch.nn.Conv2d))
        preds = [[[] for _ in out] for _ in range(G)]
        for i, scale in enumerate(out):
            for t in scale:
                for g, piece in enumerate(split_groups(t, G, owner)):
                    preds[g][i].append(piece)
        keys[id(netD)] = [k for k in _losses.GRAD_DST if k not in before]
        return preds

    def image_losses(self, real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref):
        o = self.opt
        L = _LossDict()
        wF, wT = o["lambda_F"] / (2 ** (self.n_scales - 1)), o["lambda_T"]
        imgvec = None
        if o["fused_losses"]:
            terms = [("ml1", flow, flow_ref, conf_ref, wF, 0),
                     ("ml1", resample(real_B_prev, flow), real_B, conf_ref, wT, 1),
                     ("ml1", fake_B, resample(fake_B_prev, flow_ref), conf_ref, wT, 3)]
            if o["no_first_img"]:
                terms.append(("ml1", weight, None, conf_ref, 1.0, 2))
            imgvec = fused_losses(terms, 4, o["compute_dtype"])
            L["F_Flow"], L["F_Warp"], L["W"], L["G_Warp"] = imgvec.unbind(0)
        else:
            L["F_Flow"] = masked_l1(flow, flow_ref, conf_ref) * wF
            L["F_Warp"] = masked_l1(resample(real_B_prev, flow), real_B, conf_ref) * wT
            L["W"] = masked_l1(weight, torch.zeros_like(weight), conf_ref) if o["no_first_img"] else torch.zeros((), device=flow.device)
            L["G_Warp"] = masked_l1(fake_B, resample(fake_B_prev, flow_ref).detach(), conf_ref) * wT
        if o["no_vgg"]:
            L["G_VGG"] = torch.zeros((), device=flow.device)  # VGG19 weights are not available offline (no_vgg)
        else:   # discriminator.py:132-133, :141-142
            L["G_VGG"] = (self.vgg_loss(fake_B, real_B) + self.vgg_loss(fake_B_raw, real_B)) * o["lambda_feat"]
        # The reference calls compute_loss_D twice (final and raw image, discriminator.py:125-131) and each call
        # evaluates netD on the same real pair with the same weights: identical activations, so it is
        # evaluated once here and counted twice (BatchNorm running statistics advance twice as well;
        # reference call sites discriminator.py:134 and :143).
        real_in = torch.cat((real_A, real_B), 1)
        fake_in, raw_in = torch.cat((real_A, fake_B), 1), torch.cat((real_A, fake_B_raw), 1)
        if o["batched_D"] and o["shared_fake_forward"]:
            pred_fake, pred_raw, pred_real = self._batched_D(self.netD, [fake_in, raw_in, real_in], (2, 2, 2), (2, 0, 1))
        else:
            with layers.repeated_forward(2):
                pred_real = self.netD(real_in)
            pred_fake = pred_raw = None
        if o["fused_losses"] and pred_fake is not None:
            # both compute_loss_D calls in two launches; the sums D_real + D_real2 ... are formed inside the kernels
            dvec, gvec = self._fused_D_terms(pred_real, [pred_fake, pred_raw])
            L["D_real"], L["D_fake"] = dvec.unbind(0)
            L["G_GAN"], L["G_GAN_Feat"] = gvec.unbind(0)
            L.vecs = (imgvec, dvec, gvec)
            return L
        d_real, d_fake, g_gan, g_fm = self._loss_D(self.netD, real_in, fake_in, pred_real, pred_fake)
        d_real2, d_fake2, g_gan2, g_fm2 = self._loss_D(self.netD, real_in, raw_in, pred_real, pred_raw)
        L["D_real"], L["D_fake"] = d_real + d_real2, d_fake + d_fake2
        L["G_GAN"], L["G_GAN_Feat"] = g_gan + g_gan2, g_fm + g_fm2
        return L

    def temporal_losses(self, s, real_B, fake_B, flow_ref, conf_ref):
        """compute_loss_D_T (discriminator.py:168-184) for temporal scale s."""
        b = real_B.size(0)
        h, w = real_B.shape[-2:]
        fl = (flow_ref / 20).reshape(b, -1, h, w)
        real_in = torch.cat([real_B.reshape(b, -1, h, w), fl], 1)
        fake_in = torch.cat([fake_B.reshape(b, -1, h, w), fl], 1)
        if self.opt["batched_D"] and self.opt["shared_fake_forward"]:
            pred_fake, pred_real = self._batched_D(self.netD_T[s], [fake_in, real_in], (2, 1), (1, 0))
            if self.opt["fused_losses"]:
                dvec, gvec = self._fused_D_terms(pred_real, [pred_fake])
                LT = _LossDict(zip(("D_T_real", "D_T_fake", "G_T_GAN", "G_T_GAN_Feat"), tuple(dvec.unbind(0)) + tuple(gvec.unbind(0))))
                LT.vecs = (None, dvec, gvec)
                return LT
            d_real, d_fake, g_gan, g_fm = self._loss_D(self.netD_T[s], real_in, fake_in, pred_real, pred_fake)
        else:
            d_real, d_fake, g_gan, g_fm = self._loss_D(self.netD_T[s], real_in, fake_in)
        return {"D_T_real": d_real, "D_T_fake": d_fake, "G_T_GAN": g_gan, "G_T_GAN_Feat": g_fm}

    # ------------------------------------------------------------------ temporal frame bookkeeping
    # get_skipped_frames (discriminator.py:257-271) lives in ir2rgb_amd.frames.FrameHistory.push: one preallocated
    # device buffer per stream instead of a torch.cat of the whole history per window.
    def reference_flows(self, real_B, real_B_prev, side=None):
        """All FlowNet2 evaluations of one window in ONE batched call: the reference flow of the current
        frame (train_vid2vid.py:65) and the flows of the temporally skipped real triplets
        (discriminator.py:281-283).  Both depend on real frames only, so batching them changes nothing
        numerically (FlowNet2 is per-sample, frozen, eval mode)."""
        ts = self.t_scales
        # ``resident_inputs`` (the window tensors are not produced on the main stream: bench.py's synthetic sequence, a
        # loader that prefetches on its own stream and has synchronised): the real-frame bookkeeping and FlowNet2's replay
        # then run on the second stream WITHOUT waiting for the main stream, i.e. beside the previous window's optimizer
        # step (2.7 ms of pure HBM streaming) as well as beside this window's generator forward.  Only for replays.
        early = False
        if side is not None and self.opt["resident_inputs"] and isinstance(self.flow_net, FlowNet) and real_B.shape[1] == 1:
            early = self.flow_net.will_replay(self._count_pairs(real_B), real_B.reshape((-1,) + tuple(real_B.shape[2:])))
        main = torch.cuda.current_stream(real_B.device) if real_B.is_cuda else None
        if early:
            if not self._early_on:                          # first time: the histories were last written on the main stream
                side.wait_stream(main)
                self._early_on = True
            import os
            if self._backward_done is not None and os.environ.get("IR2RGB_FLOW_BOUND", "1") == "1":
                # not before the previous window's generator backward pass is through: the host runs windows ahead of the
                # GPU, and a FlowNet2 that started whenever it was issued would also share the chip with that pass's
                # compute-bound convolutions (no gain, and it spoils bench.py's per-kernel brackets); beside the
                # discriminators' latency-bound passes and the HBM-bound optimizer step it is what fills the chip
                side.wait_event(self._backward_done)
            with torch.cuda.stream(side):
                return self._reference_flows(real_B, real_B_prev, side)
        if self._early_on:                                  # back to the main stream (a new shape: lazy work ahead)
            main.wait_stream(side)
            self._early_on = False
        return self._reference_flows(real_B, real_B_prev, side)

    def _count_pairs(self, real_B):
        """Frame pairs FlowNet2 will see for this one-frame push (host arithmetic only: what _reference_flows is about to
        assemble -- the frame itself, plus per temporal scale whose tuple exists the newest pair, or all tD-1 of them)."""
        h, b = self.hist["real"], real_B.shape[0]
        total, pushed = h.len + 1, h.pushed + 1
        reuse = self.opt["reuse_skipped_flows"] and self.tD == 3
        n = b
        for s in range(1, self.t_scales):
            step = self.tD ** s
            if total - step * (self.tD - 1) >= 1:
                hit = reuse and any(e[0] == pushed - step for e in self._pair_flows.get(s, []))
                n += b * (1 if hit else self.tD - 1)
        return n

    def _reference_flows(self, real_B, real_B_prev, side):
        ts = self.t_scales
        rb_s = self.hist["real"].push(real_B)
        pushed = self.hist["real"].pushed
        firsts, seconds, owners = [real_B.reshape((-1,) + tuple(real_B.shape[2:]))], [real_B_prev.reshape((-1,) + tuple(real_B.shape[2:]))], []
        # Of the tD-1 frame pairs of a temporally skipped tuple (frames tD**s apart) only the newest is new: with one frame
        # per window, pair k of this window is pair k+1 of the window tD**s pushes ago (same two real frames; FlowNet2 is
        # frozen and per-sample), so its flow and confidence are kept instead of being recomputed (the reference recomputes
        # them, discriminator.py:281-283; a third of its FlowNet2 work per window at t_scales 2 / tD 3).
        reuse = self.opt["reuse_skipped_flows"] and real_B.shape[1] == 1 and self.tD == 3     # (tD 3: one older pair per tuple)
        cached = {}
        for s in range(1, ts):
            if rb_s[s] is not None and rb_s[s].size(1) == self.tD:
                a, b = rb_s[s][:, 1:], rb_s[s][:, :-1]
                old = self._pair_flows.get(s, [])
                hit = [e for e in old if e[0] == pushed - self.tD ** s] if reuse else []
                if hit:
                    cached[s] = hit[0]
                    a, b = a[:, -1:], b[:, -1:]                          # only the newest pair goes through FlowNet2
                firsts.append(a.reshape((-1,) + tuple(a.shape[2:])))
                seconds.append(b.reshape((-1,) + tuple(b.shape[2:])))
                owners.append((s, a.shape[0], a.shape[1]))
        if side is not None and isinstance(self.flow_net, FlowNet):
            flow, conf = self.flow_net(torch.cat(firsts), torch.cat(seconds), side=side)
        else:
            flow, conf = self.flow_net(torch.cat(firsts), torch.cat(seconds))
        n0 = firsts[0].shape[0]
        b, t = real_B.shape[:2]
        h, w = real_B.shape[-2:]
        flow_ref, conf_ref = flow[:n0].view(b, t, 2, h, w), conf[:n0].view(b, t, 1, h, w)
        extra, off = {}, n0
        # FlowNet2 may have replayed on the second stream, which the main stream joins only before the losses
        # (train_window): the concatenation below reads its result, so it has to be queued on that stream too
        ran = getattr(self.flow_net, "ran_on", None) if isinstance(self.flow_net, FlowNet) else None
        on_side = ran is not None and torch.cuda.current_stream(flow.device) != ran
        with (torch.cuda.stream(ran) if on_side else contextlib.nullcontext()):
            extra = self._assemble_pair_flows(flow, conf, owners, cached, n0, pushed, reuse, h, w)
        return flow_ref, conf_ref, rb_s, extra

    def _assemble_pair_flows(self, flow, conf, owners, cached, off, pushed, reuse, h, w):
        extra = {}
        for s, bb, tt in owners:
            fl, cf = flow[off:off + bb * tt].view(bb, tt, 2, h, w), conf[off:off + bb * tt].view(bb, tt, 1, h, w)
            off += bb * tt
            if reuse:       # the newest pair's result serves the windows to come (tD - 2 more uses, tD**s pushes apart)
                keep = [e for e in self._pair_flows.get(s, []) if e[0] > pushed - self.tD ** s * (self.tD - 2)]
                self._pair_flows[s] = keep + [(pushed, fl[:, -1:], cf[:, -1:])]
                if SC.ENABLED:
                    SC.consumed(flow, "reference flow (FlowNet2 replay)")
                    SC.produced(fl[:, -1:], "kept pair flow"), SC.produced(cf[:, -1:], "kept pair confidence")
            if s in cached:                                              # (tD == 3: exactly one older pair)
                if fl.is_cuda:
                    for t in cached[s][1:]:
                        t.record_stream(torch.cuda.current_stream(fl.device))   # (it may have been made on the other stream)
                if SC.ENABLED:
                    SC.consumed(flow, "reference flow (FlowNet2 replay)")
                    SC.consumed(cached[s][1], "kept pair flow"), SC.consumed(cached[s][2], "kept pair confidence")
                fl, cf = torch.cat([cached[s][1], fl], 1), torch.cat([cached[s][2], cf], 1)
            extra[s] = (fl, cf)
        return extra

    def skipped_frames(self, rb_s, extra_flows, fake_B, flow_ref, conf_ref):
        """get_all_skipped_frames, dense variant (discriminator.py:219-234, :273-283); the real-frame
        bookkeeping and the FlowNet2 calls were done by reference_flows."""
        ts = self.t_scales
        fb_s = self.hist["fake"].push(fake_B)
        fl0, cf0 = self.hist["flow"].push(flow_ref), self.hist["conf"].push(conf_ref)
        fl_s, cf_s = [None] * ts, [None] * ts
        if fl0[0] is not None:
            fl_s[0], cf_s[0] = fl0[0][:, 1:], cf0[0][:, 1:]
        for s in range(1, ts):
            if s in extra_flows:
                fl_s[s], cf_s[s] = extra_flows[s]
        return rb_s, fb_s, fl_s, cf_s

    # ------------------------------------------------------------------ the loop body (a14)
    def get_losses(self, L, LT):
        """Vid2VidModelD.get_losses (discriminator.py:236-248): ``L`` the image-loss dict, ``LT`` the list of temporal
        dicts of the active temporal scales.  Returns (loss_G, loss_D, [loss_D_T per active scale]).
        Dicts that came out of the fused loss kernels carry their terms as vectors (``_LossDict.vecs``):
        the totals are then one concatenation + one sum for loss_G and one sum per discriminator -- not ~45 scalar adds
        forward and as many select / add nodes backward."""
        mine = [getattr(d, "vecs", None) for d in [L] + list(LT)]
        if all(v is not None for v in mine) and mine[0][0] is not None:
            imgvec, dvec, gvec = mine[0]
            g_parts = [imgvec, gvec] + [m[2] for m in mine[1:]]
            loss_G = torch.cat(g_parts).sum()
            if not self.opt["no_vgg"]:
                loss_G = loss_G + L["G_VGG"]
            loss_D = dvec.sum() * 0.5
            return loss_G, loss_D, [m[1].sum() * 0.5 for m in mine[1:]]
        loss_D = (L["D_fake"] + L["D_real"]) * 0.5
        loss_G = L["G_GAN"] + L["G_GAN_Feat"] + L["G_VGG"] + L["G_Warp"] + L["F_Flow"] + L["F_Warp"] + L["W"]
        loss_D_T = []
        for lt in LT:                                            # G_T_Warp is identically zero (discriminator.py:106)
            loss_G = loss_G + lt["G_T_GAN"] + lt["G_T_GAN_Feat"]
            loss_D_T.append((lt["D_T_fake"] + lt["D_T_real"]) * 0.5)
        return loss_G, loss_D, loss_D_T

    def backward_passes(self, loss_G, loss_D, loss_D_T, g_inputs=None, early_adam=False):
        """The three ``loss.backward()`` of train_vid2vid.py:104-111 (zero_grad included); each optimizer's gradient
        all-reduce is issued as soon as its pass ends, so it overlaps the next pass.  ``g_inputs``: the tensors
        the generator's pass differentiates with respect to (default: the generator parameters).  ``early_adam`` (train_window):
        the generators' optimizer step is issued right after their pass, on its own stream; optimizer_steps() then skips it."""
        self.grads_G.zero()
        self.grads_D.zero()
        for gdt in self.grads_DT:
            gdt.zero()
        shared = self.opt["shared_fake_forward"]
        d_nets = [self.netD] + self.netD_T
        if g_inputs is None and shared:
            g_inputs = self.grads_G.params
        # shared discriminator forwards are walked twice: by the generator's pass (frames only) and by the
        # discriminators' passes (parameters only); without sharing the flags are no-ops
        # (inputs=...: the engine then runs only the nodes that lead to those tensors, so the
        # discriminators' passes never enter the generator graph)
        batched = shared and self.opt["batched_D"]
        with autograd.backward_flags([self.netD] if shared else [], autograd.SKIP_PARAM_GRADS, 2 if batched else None), \
                autograd.backward_flags(self.netD_T if shared else [], autograd.SKIP_PARAM_GRADS, 1 if batched else None):
            loss_G.backward(retain_graph=shared, inputs=g_inputs)
        self.grads_G.all_reduce_async(self.world)
        if self._early_on:           # (the next window's FlowNet2 may start from here on, see reference_flows)
            if self._backward_done is None:
                self._backward_done = torch.cuda.Event()
            self._backward_done.record()
        if self.adam_stream_on and early_adam:
            # The generators' optimizer step needs nothing but their finished pass: it starts here, on its own stream, beside
            # the discriminators' backward passes (which never touch generator weights or gradients) -- 2 ms of pure HBM
            # streaming next to small latency-bound launches.  optimizer_steps() joins it before the repack.
            main = torch.cuda.current_stream(self.device)
            if self._adam_stream is None:
                self._adam_stream = torch.cuda.Stream(self.device)
            self._adam_stream.wait_stream(main)
            with torch.cuda.stream(self._adam_stream):
                self.grads_G.wait()
                self.optimizer_G.step()
            self._adam_pending = True
        with autograd.backward_flags(d_nets if shared else [], autograd.SKIP_INPUT_GRAD):
            if self.d_streams and shared and loss_D_T:
                # one pass over the three disjoint graphs: every node runs on the stream of its forward, so the
                # discriminators' passes overlap as their forwards did (separate backward() calls would each end with
                # the calling stream waiting for the pass, i.e. one discriminator after the other)
                params = list(self.grads_D.params)
                for s in range(len(loss_D_T)):
                    params += list(self.grads_DT[s].params)
                torch.autograd.backward([loss_D] + list(loss_D_T), inputs=params)
                self.grads_D.all_reduce_async(self.world)
                for s in range(len(loss_D_T)):
                    self.grads_DT[s].all_reduce_async(self.world)
            else:
                loss_D.backward(inputs=self.grads_D.params if shared else None)
                self.grads_D.all_reduce_async(self.world)
                for s, ld in enumerate(loss_D_T):
                    ld.backward(inputs=self.grads_DT[s].params if shared else None)
                    self.grads_DT[s].all_reduce_async(self.world)

    def optimizer_steps(self, n_temporal):
        """The three ``optimizer.step()`` of train_vid2vid.py:104-111, then one launch refreshing every packed weight."""
        if self._adam_pending:          # (issued by backward_passes on the Adam stream)
            self._adam_pending = False
        else:
            self.grads_G.wait()
            self.optimizer_G.step()
        self.grads_D.wait()
        self.optimizer_D.step()
        for s in range(n_temporal):
            self.grads_DT[s].wait()
            self.optimizer_D_T[s].step()
        if self._adam_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._adam_stream)
        if self.repacker is not None:
            self.repacker.run()          # every packed forward / data-gradient weight copy, one launch

    def train_window(self, input_A, input_B):
        """train_vid2vid.py:54-111 for one window.  Returns a dict of detached scalar losses: the totals ``G``, ``D``,
        ``D_T{s}`` and every term under the reference's names (``G_GAN`` ... ``W``; temporal ones with the scale
        appended, ``G_T_GAN0`` ...).  ``self.last_outputs`` keeps (fake_B, fake_B_raw, flow, weight), detached."""
        fake_prev_last = self.fake_B_prev
        from . import losses as _losses
        _losses.GRAD_DST.clear()            # (gradient destinations of the previous window's discriminator outputs)
        # The reference flows depend on real frames only (train_vid2vid.py:62-65 computes them after the generator, from
        # real_Bp = input_B[:, tG-2:]): FlowNet2 -- frozen, no autograd, replayed from a HIP graph from its third call at a
        # shape on -- runs on a second HIP stream BESIDE the generator forward and is joined before the losses that read
        # its result (-1.0 ms per window).  Calls that are not yet replays (eager warm-up, capture) run on the main
        # stream, before the generator: a capture has to own its stream.
        tG = self.opt["n_input_gen_frames"]
        real_Bp_in = input_B[:, tG - 2:]
        side = self._flow_stream(input_B)
        flow_ref, conf_ref, rb_s, extra_flows = self.reference_flows(real_Bp_in[:, 1:], real_Bp_in[:, :-1], side)
        ran_on = getattr(self.flow_net, "ran_on", None)
        from . import conv as _conv
        _conv.SIDE_BUSY = ran_on is not None        # (per-launch timing brackets skip kernels that share the chip)
        fake_B, fake_B_raw, flow, weight, real_A, real_Bp = self.generate(input_A, input_B)
        _conv.SIDE_BUSY = False
        real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
        if ran_on is not None:                      # FlowNet2 replayed on the side stream: join it here
            torch.cuda.current_stream(input_B.device).wait_stream(ran_on)
            for t in [flow_ref, conf_ref] + [x for pair in extra_flows.values() for x in pair] + [x for x in rb_s if x is not None]:
                t.record_stream(torch.cuda.current_stream(input_B.device))
        # compute_fake_B_prev (generator.py:283-287) AS THE REFERENCE'S LOOP EVALUATES IT.  train_vid2vid.py:60,:67-68
        # hands the previous window's pyramid LIST to model_g and afterwards to compute_fake_B_prev; in between,
        # generate_frame_train appends the new frames to the elements of that very list (generator.py:113, :175:
        # ``fake_B_pyr[si] = concat([fake_B_pyr[si], fake_B])`` on the caller's object).  So from the second window on
        # ``fake_B_last[0][:, -1:]`` is the frame generated in THIS window, and G_Warp compares fake_B with its own
        # flow_ref-warped copy.  Reproduced on purpose (pinned by tests/golden/window_*.npz, whose generating script
        # executes the reference's statements and therefore has the same aliasing); only the first window of a
        # sequence uses the given previous frame.
        fbp = real_B_prev[:, 0:1] if fake_prev_last is None else fake_B[:, -1:].detach()
        if fake_B.size(1) > 1:
            fbp = torch.cat([fbp, fake_B[:, :-1].detach()], 1)
        flat = lambda t: t.reshape((-1,) + tuple(t.shape[2:]))  # noqa: E731
        if SC.ENABLED:      # the losses read FlowNet2's results on this stream
            SC.consumed(flow_ref, "reference flow (FlowNet2 replay)"), SC.consumed(conf_ref, "flow confidence (FlowNet2 replay)")
        if self.d_streams:
            # The temporal discriminators first, each on its own stream (they wait for what the main stream has queued so
            # far: the generator and the frame bookkeeping), then the image discriminator on the main stream: four
            # independent networks of small layers side by side instead of one after the other.  The streams are joined
            # before the totals; the backward passes follow the forward streams (autograd), see backward_passes.
            from .networks import _Branch
            rb_s, fb_s, fl_s, cf_s = self.skipped_frames(rb_s, extra_flows, fake_B, flow_ref, conf_ref)
            active = [s for s in range(self.t_scales) if rb_s[s] is not None]
            LT, joins = [], []
            for s in active:
                br = _Branch(rb_s[s], fb_s[s], fl_s[s], cf_s[s], slot=16 + (s if _T_SLOTS == "own" else 0), force=True)
                with br:
                    lt = self.temporal_losses(s, rb_s[s], fb_s[s], fl_s[s], cf_s[s])
                LT.append(lt)
                joins.append((br, lt))
            L = self.image_losses(flat(real_B), flat(fake_B), flat(fake_B_raw), flat(real_A), flat(real_B_prev), flat(fbp),
                                  flat(flow), flat(weight), flat(flow_ref), flat(conf_ref))
            for br, lt in joins:
                vec = getattr(lt, "vecs", None)
                br.join(*(list(lt.values()) + ([v for v in vec if v is not None] if vec else [])))
        else:
            L = self.image_losses(flat(real_B), flat(fake_B), flat(fake_B_raw), flat(real_A), flat(real_B_prev), flat(fbp),
                                  flat(flow), flat(weight), flat(flow_ref), flat(conf_ref))
            rb_s, fb_s, fl_s, cf_s = self.skipped_frames(rb_s, extra_flows, fake_B, flow_ref, conf_ref)
            active = [s for s in range(self.t_scales) if rb_s[s] is not None]
            LT = [self.temporal_losses(s, rb_s[s], fb_s[s], fl_s[s], cf_s[s]) for s in active]
        loss_G, loss_D, loss_D_T = self.get_losses(L, LT)
        self.backward_passes(loss_G, loss_D, loss_D_T, early_adam=True)
        self.optimizer_steps(len(loss_D_T))
        self.last_outputs = tuple(t.detach() for t in (fake_B, fake_B_raw, flow, weight))
        out = {"G": loss_G.detach(), "D": loss_D.detach()}
        out.update({f"D_T{s}": l.detach() for s, l in enumerate(loss_D_T)})
        out.update({k: v.detach() for k, v in L.items()})
        for s, lt in zip(active, LT):
            out.update({f"{k}{s}": v.detach() for k, v in lt.items()})
        return out


def synthetic_sequence(n_frames, h, w, seed, device, channels=3):
    """smooth random field (Gaussian noise, 15x15 box blur, tanh) translated by a
    per-sequence constant (dx,dy) in [-8,8]^2 px per frame.  Returns (A, B) [1, n_frames, C, H, W] in [-1,1]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    dx, dy = [int(v) for v in torch.randint(-8, 9, (2,), generator=g)]
    pad = 8 * n_frames + 8
    out = []
    for _ in range(2):
        base = torch.randn(1, channels, h + 2 * pad, w + 2 * pad, generator=g).to(device)
        base = torch.tanh(F.avg_pool2d(F.pad(base, (7, 7, 7, 7), mode="reflect"), 15, stride=1) * 6)
        frames = [base[:, :, pad + t * dy:pad + t * dy + h, pad + t * dx:pad + t * dx + w] for t in range(n_frames)]
        out.append(torch.stack(frames, 1).contiguous())
    return out[0], out[1]
