"""Window arithmetic of the reference's video loader.

``VideoSeq`` is the object ``train_vid2vid.py`` iterates per video (data/dataset/vid2vid.py:24-52): the
frames of one sequence stacked along the channel axis, cut into overlapping windows of ``t_len`` frames.
Restated here (the reference module imports torchvision / PIL at the top and cannot be imported in this
environment; there are no reference tests for it: parity unpinned, the arithmetic below follows the
source line by line, including the doubled ``n_frames_load`` term of ``t_len``)."""


class VideoSeq:
    def __init__(self, ir_frames, rgb_frames, annotations=None, **kwargs):
        self.ir_frames, self.rgb_frames, self.annotations = ir_frames, rgb_frames, annotations
        t_g = kwargs["n_input_gen_frames"]
        self.n_gpus = kwargs["gen_gpus"]
        self.input_nc, self.output_nc = kwargs["input_nc"], kwargs["output_nc"]
        # rgb_frames: [B, n_frames * output_nc, H, W]                                   (vid2vid.py:36-37)
        _, n_ch, self.height, self.width = self.rgb_frames.size()
        n_frames_total = n_ch // self.output_nc
        n_frames_load = kwargs["max_frames_per_gpu"] * kwargs["gen_gpus"]             # vid2vid.py:39
        self.n_frames_load = min(n_frames_load, n_frames_total - t_g + 1)             # vid2vid.py:40
        self.t_len = self.n_frames_load + n_frames_load + t_g - 1                     # vid2vid.py:41 (sic)
        self.n_frames_total = n_frames_total - self.t_len + 1                         # vid2vid.py:42

    def __getitem__(self, i):
        t, h, w = self.t_len, self.height, self.width
        ir = self.ir_frames[:, i * self.input_nc:(i + t) * self.input_nc].view(-1, t, self.input_nc, h, w)
        rgb = self.rgb_frames[:, i * self.output_nc:(i + t) * self.output_nc].view(-1, t, self.output_nc, h, w)
        return ir, rgb

    def __len__(self):
        return self.n_frames_total // self.t_len                                      # vid2vid.py:51-52
