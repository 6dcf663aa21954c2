"""Frame bookkeeping of the vid2vid loop as allocation-free device code.

Two things in the reference's loop re-allocate and re-copy whole frame stacks every window:

* ``get_skipped_frames`` (models/discriminator.py:257-271): ``B_all = torch.cat([B_all.detach(), B], dim=1)`` -- the
  history of up to tD**(t_scales-1) * (tD-1) frames is copied into a fresh allocation for every new frame, four times
  per window (real, generated, reference flow, confidence), before the temporally skipped tD-tuples are sliced out;
* ``VideoSeq.__getitem__`` (data/dataset/vid2vid.py:44-49) hands out windows of a video as reshaped slices.

``FrameHistory`` keeps one stream's history in ONE preallocated device buffer: a new frame costs one frame copy, the
tuples are strided views of the buffer (gathered once, into the contiguous stack the temporal discriminator is fed
anyway), and when the write position reaches the end the live tail moves to the front -- once per ``keep`` windows, so
on average one more frame copy per window instead of ``keep``.  ``WindowSlicer`` holds a sequence on the device and
serves its windows as zero-copy views.
"""
import torch

from . import streamcheck as SC


class FrameHistory:
    """History of one frame stream ([B, T, C, H, W] pushes) with the reference's temporal sub-sampling.

    ``push(B)`` is ``get_skipped_frames(B_all, B, t_scales, tD)`` (discriminator.py:257-271) with ``B_all`` held here:
    it returns the list ``[skipped_0, ..., skipped_{t_scales-1}]`` where ``skipped_s`` stacks, along the batch axis, the
    tD-tuples of frames tD**s apart that end at the newest frames, or None while the history is too short.  The stored
    history is detached (the reference detaches it when it is re-used, discriminator.py:258); the tuples that contain
    frames of the CURRENT push keep their autograd link to ``B``."""

    def __init__(self, t_scales, tD):
        self.t_scales, self.tD = int(t_scales), int(tD)
        self.keep = self.tD ** (self.t_scales - 1) * (self.tD - 1)     # frames the next push can still reach back to
        self.buf = None
        self.start = self.len = 0                                       # live frames: buf[:, start:start+len]
        self.pushed = 0                                                 # frames pushed since the last reset

    def reset(self):
        self.start = self.len = self.pushed = 0

    def _reserve(self, B):
        b, n = B.shape[:2]
        need = self.keep + n
        if self.buf is None or self.buf.shape[0] != b or self.buf.shape[2:] != B.shape[2:] or self.buf.dtype != B.dtype \
                or self.buf.device != B.device or self.buf.shape[1] < 2 * need:
            live = self.frames() if (self.buf is not None and self.len and self.buf.shape[0] == b
                                     and self.buf.shape[2:] == B.shape[2:]) else None
            self.buf = torch.empty((b, 2 * need) + tuple(B.shape[2:]), dtype=B.dtype, device=B.device)
            self.start = self.len = 0
            if live is not None:
                self.buf[:, :live.shape[1]].copy_(live)
                self.len = live.shape[1]
        if self.start + self.len + n > self.buf.shape[1]:              # wrap: the live tail moves to the front
            tail = min(self.len, self.keep)
            a = self.start + self.len - tail                            # >= keep + n >= tail: the ranges cannot overlap
            self.buf[:, :tail].copy_(self.buf[:, a:a + tail])
            self.start, self.len = 0, tail

    def frames(self):
        """The stored (detached) history, oldest first: a view."""
        return self.buf[:, self.start:self.start + self.len]

    def load(self, frames):
        """Replace the history by ``frames`` [B, T, C, H, W] (its last ``keep`` frames are what counts)."""
        frames = frames.detach()[:, -self.keep:] if self.keep else frames.detach()[:, :0]
        self.start = self.len = 0
        self._reserve(frames[:, :1] if frames.shape[1] else frames)
        self.buf[:, :frames.shape[1]].copy_(frames)
        self.len = frames.shape[1]

    def push(self, B):
        n = B.shape[1]
        self.pushed += n
        self._reserve(B)
        if SC.ENABLED:
            SC.consumed(self.buf, "frame history")
        with torch.no_grad():
            self.buf[:, self.start + self.len:self.start + self.len + n].copy_(B)
        if SC.ENABLED:
            SC.produced(self.buf, "frame history")
        total = self.len + n                                            # frames of history + this push
        hist = self.buf[:, self.start:self.start + total]              # detached storage of all of them
        skipped = [None] * self.t_scales
        for s in range(self.t_scales):
            step = self.tD ** s
            span = step * (self.tD - 1)
            n_groups = min(total - span, n)
            groups = []
            for t in range(0, max(n_groups, 0), self.tD):
                idx = [total - 1 - t - span + k * step for k in range(self.tD)]       # B_all[:, -span-t-1 : -t : step]
                if B.requires_grad and idx[-1] >= self.len:
                    # frames of this push carry gradient: take them from B itself, the older ones from the buffer
                    parts = [B[:, i - self.len] if i >= self.len else hist[:, i] for i in idx]
                    groups.append(torch.stack(parts, 1))
                else:
                    groups.append(hist[:, idx[0]:idx[-1] + 1:step].clone(memory_format=torch.contiguous_format))   # the one gather
            if groups:
                skipped[s] = groups[0] if len(groups) == 1 else torch.cat(groups)
        # what the next push may still need: the last ``keep`` frames
        drop = max(total - self.keep, 0)
        self.start, self.len = self.start + drop, total - drop
        return skipped


class WindowSlicer:
    """A video held on the device, cut into the overlapping windows the training loop consumes: window i = frames
    [i * n_frames_load, i * n_frames_load + n_frames_load + tG - 1) of both streams, as zero-copy views [B, t, C, H, W].

    This is what ``VideoSeq`` (data/dataset/vid2vid.py:24-52) serves the loop of train_vid2vid.py:54, with the
    window length the generator's forward actually consumes (n_frames_load + tG - 1, generator.py:99-123, :217-235).
    The reference's ``t_len`` adds ``n_frames_load`` a second time (vid2vid.py:41) and steps its windows by ONE frame
    whatever ``n_frames_load`` is (vid2vid.py:46-47); with its default n_frames_load = 1 that hands the generator one
    frame more than it reads and leaves real_A / real_B one frame longer than fake_B, which compute_loss_D then cannot
    concatenate -- not reproduced.  Input layout as the reference's loader delivers it: frames
    stacked along the channel axis, [B, n_frames * C, H, W]."""

    def __init__(self, ir_frames, rgb_frames, n_input_gen_frames=3, n_frames_load=1, input_nc=3, output_nc=3):
        if ir_frames.dim() != 4 or rgb_frames.dim() != 4:
            raise ValueError("WindowSlicer: [B, n_frames * C, H, W] tensors expected")
        b, ca, h, w = ir_frames.shape
        if ca % input_nc or rgb_frames.shape[1] % output_nc:
            raise ValueError("WindowSlicer: channel count is not a multiple of the per-frame channels")
        self.n_frames = ca // input_nc
        if rgb_frames.shape[1] // output_nc != self.n_frames:
            raise ValueError("WindowSlicer: the two streams hold different numbers of frames")
        self.A = ir_frames.view(b, self.n_frames, input_nc, h, w)
        self.B = rgb_frames.view(b, self.n_frames, output_nc, h, w)
        self.tG, self.n_load = int(n_input_gen_frames), int(n_frames_load)
        self.t_len = self.n_load + self.tG - 1

    def __len__(self):
        return max((self.n_frames - (self.tG - 1)) // self.n_load, 0)

    def __getitem__(self, i):
        if not 0 <= i < len(self):
            raise IndexError(i)
        s = i * self.n_load
        return self.A[:, s:s + self.t_len], self.B[:, s:s + self.t_len]

    def __iter__(self):
        return (self[i] for i in range(len(self)))
