"""bench.py -- generator+discriminator training frames/sec at 512x1024 (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one window of the vid2vid inner loop on one GPU: two-scale generator
forward (G0 composite ngf 128 @256x512 + G1 composite-local ngf 64 @512x1024), FlowNet2 reference
flow, image discriminator (num_D 2) and two temporal discriminators, three backward passes, three
Adam steps, RCCL all-reduce of all gradients when N > 1 (frame-parallel: every rank runs its own
synthetic sequence; weak scaling).  Synthetic inputs and seeded random-init weights (no datasets or
checkpoints offline), resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline     -- the dominant kernel (MFMA implicit-GEMM 3x3 convolution), HIP-event timed per launch
                  inside the timed region on the launch stream: algorithmic flops / mean duration
                  against the 2.5 PFLOP/s dense bf16 peak (MI355X_MICROARCH.md);
  cpu_baseline -- the plain-torch fp32 restatement of the single-scale generator (oracle/networks_oracle.py), forward
                  and forward+backward at 512x1024 on the host cores (median of 3, CPU model and core count stated),
                  rank 0 at N=1 only;
  extra        -- generator-forward timings incl. the north-star 512x1024 single-scale forward.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = 2500.0  # dense bf16/f16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
H, W = 512, 1024
PREROLL = 12          # windows until both temporal discriminator scales are active AND every lazy step is behind us:
                      # the skipped pairs' flows are reused from window 9 on, which changes FlowNet2's batch, and its
                      # HIP graph at the new shape is captured on the third call (window 11)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota when one is set
    (os.cpu_count() reports the whole host; 256 threads on a 16-CPU share ran the port 15x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(threads):
    """the plain-torch fp32 restatement of the single-scale CompositeGenerator(ngf 128) -- the
    reference's graph (oracle/networks_oracle.py, pinned by the reference goldens) -- at 512x1024 on the host cores:
    forward, and forward + backward, median of 3 after one warm-up forward.  A bounded sample (one frame per run) of the
    generator part of the workload; the GPU metric additionally runs FlowNet2, the discriminators and Adam."""
    from ir2rgb_amd import networks as N
    from oracle import networks_oracle as NO
    torch.set_num_threads(threads)
    opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
    torch.manual_seed(0)
    g = N.build_generator_module(9, 3, 6, 128, "composite", 3, "batch", 0, **opt).train()
    gen = torch.Generator().manual_seed(1)
    A = torch.tanh(torch.randn(1, 9, H, W, generator=gen))
    P = torch.tanh(torch.randn(1, 6, H, W, generator=gen))

    def fwd():
        with torch.no_grad():
            t0 = time.time()
            NO.generator_forward(g, A, P)
            return time.time() - t0

    def fwd_bwd():
        t0 = time.time()
        out = NO.generator_forward(g, A, P)
        (out[0].mean() + out[1].abs().mean() * 0.01 + out[2].mean()).backward()
        dt = time.time() - t0
        g.zero_grad(set_to_none=True)
        return dt

    fwd()                                                  # warm-up (thread pool, allocator, oneDNN primitives)
    t_f = sorted(fwd() for _ in range(3))[1]
    t_fb = sorted(fwd_bwd() for _ in range(3))[1]
    return {"value": round(1.0 / t_fb, 4), "unit": "frames/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "forward_s": round(t_f, 2), "forward_backward_s": round(t_fb, 2), "forward_TFLOPs": round(6.632 / t_f, 3),
            "sample": "1 frame per run, median of 3: fp32 plain-torch restatement of CompositeGenerator(ngf 128, 3 down, 9 blocks) "
                      f"at 512x1024 (6.63 TFLOP forward): forward {t_f:.1f} s, forward+backward {t_fb:.1f} s (value = 1 / the "
                      "latter); the GPU metric additionally runs the second spatial scale, FlowNet2, the discriminators, "
                      "their backward passes and Adam"}


def _timed(fn, n, warm=3):
    """Mean milliseconds of fn() over n back-to-back calls (HIP events on the current stream)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def config5_forward(dev):
    """BASELINE config 5: the two-scale generator forward at 1024x2048 in f16 (G0 composite ngf 128 @512x1024 feeding G1
    composite-local ngf 64 @1024x2048; reference models/networks.py:103-317): 6.632 + 2.44 = 9.07 TFLOP of convolutions."""
    from ir2rgb_amd import networks as N
    from ir2rgb_amd.graphs import GraphedForward
    opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
    torch.manual_seed(0)
    g0 = N.build_generator_module(9, 3, 6, 128, "composite", 3, "batch", 0, **opt).to(dev).train()
    g1 = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
    g0.compute_dtype = g1.compute_dtype = torch.float16
    h, w = 2 * H, 2 * W
    x, p = torch.tanh(torch.randn(1, 9, h, w, device=dev)), torch.tanh(torch.randn(1, 6, h, w, device=dev))
    x0, p0 = torch.nn.functional.avg_pool2d(x, 2), torch.nn.functional.avg_pool2d(p, 2)

    def fwd(x0, p0, x, p):
        o0 = g0(x0, p0, None, None, None, None, False)
        return g1(x, p, None, o0[4], o0[5], None, False)[:4]

    tflop = 6.632 + 2.44
    out = {"algorithmic_TFLOP": tflop, "dtype": "f16"}
    with torch.no_grad():
        ms = _timed(lambda: fwd(x0, p0, x, p), 5)
        out["eager"] = {"ms": round(ms, 3), "TFLOPs": round(tflop / ms * 1e3, 1), "frac_of_peak": round(tflop / ms * 1e3 / PEAK_TFLOPS, 4)}
    gf = GraphedForward(fwd, x0, p0, x, p)
    ms = _timed(lambda: gf(x0, p0, x, p), 10, warm=2)
    out["hip_graph"] = {"ms": round(ms, 3), "TFLOPs": round(tflop / ms * 1e3, 1), "frac_of_peak": round(tflop / ms * 1e3 / PEAK_TFLOPS, 4)}
    return out


def config3_operators(dev):
    """BASELINE config 3: the three FlowNet2 operators at the sizes a 512x1024 frame pair gives them (algorithmic
    bytes 31.2 MB / 16.8 MB / 8.4 MB), back-to-back launches through the reference's operator API, HBM peak 8 TB/s."""
    import ctypes
    from ir2rgb_amd import _lib, conv as CV
    from ir2rgb_amd.flownet2_pytorch.networks.channelnorm_package.channelnorm import ChannelNorm
    from ir2rgb_amd.flownet2_pytorch.networks.correlation_package.correlation import Correlation
    from ir2rgb_amd.flownet2_pytorch.networks.resample2d_package.resample2d import Resample2d
    f1, f2 = torch.randn(1, 256, 64, 128, device=dev), torch.randn(1, 256, 64, 128, device=dev)
    img, flow = torch.randn(1, 3, H, W, device=dev), torch.randn(1, 2, H, W, device=dev) * 4
    corr, res, cn = Correlation(20, 1, 20, 1, 2, 1), Resample2d(), ChannelNorm()
    out = {}

    def graphed_us(fn, reps=20):
        """Device time per launch: ``reps`` calls captured in one HIP graph (no Python / launch overhead between them)."""
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                keep = [fn() for _ in range(reps)]
        torch.cuda.current_stream(dev).wait_stream(side)
        ms = _timed(graph.replay, 10, warm=2)
        del keep
        return ms * 1e3 / reps

    with torch.no_grad():
        for name, fn, mb, gflop in (("correlation_fwd_fp32_nchw", lambda: corr(f1, f2), 31.2, 1.85),
                                    ("resample2d_fwd", lambda: res(img, flow), 16.8, None),
                                    ("channelnorm_fwd", lambda: cn(img), 8.4, None)):
            us = _timed(fn, 50, warm=5) * 1e3
            out[name] = {"us": round(us, 1), "GB_per_s": round(mb * 1e6 / (us * 1e-6) / 1e9, 0), "frac_of_hbm_peak": round(mb * 1e6 / (us * 1e-6) / 8e12, 3)}
            if gflop:
                out[name]["TFLOPs"] = round(gflop * 1e3 / us, 1)
            try:        # the same launches replayed from a HIP graph: the kernels' own time
                gus = graphed_us(fn)
                out[name].update({"hip_graph_us": round(gus, 1), "hip_graph_GB_per_s": round(mb * 1e6 / (gus * 1e-6) / 1e9, 0),
                                  "hip_graph_frac_of_hbm_peak": round(mb * 1e6 / (gus * 1e-6) / 8e12, 3)})
            except RuntimeError as e:   # (a capture that fails leaves the API-level number standing)
                out[name]["hip_graph_error"] = str(e)[:120]
        # the form FlowNet2 runs inside this package: half NHWC feature maps -> banded MFMA products, fp32 planes out
        a = f1.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = f2.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        o = torch.empty(1, 441, 64, 128, device=dev)
        lib = _lib.lib()

        def mf():
            rc = lib.ir2rgb_correlation_nhwc_half(CV._p(a), 256, 0, CV._p(b), 256, 0, CV._p(o), 0, 0, 0, 1.0, 1, 256, 64, 128, CV.BF16,
                                                  _lib.current_stream(o))
            assert rc == 0
        us = _timed(mf, 50, warm=5) * 1e3
        mb = (2 * 4.19 + 14.45)
        out["correlation_fwd_nhwc_half_mfma"] = {"us": round(us, 1), "GB_per_s": round(mb * 1e6 / (us * 1e-6) / 1e9, 0),
                                                 "TFLOPs_useful": round(1.85e3 / us, 1)}
    out["note"] = ("us: API-level launches incl. the Python operator wrappers (allocation of the output, autograd Function); "
                   "hip_graph_us: the same launches replayed from a HIP graph, 20 per replay (device time per launch)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # Rehearsal switches (never set by the driver): IR2RGB_BENCH_BACKEND=gloo + IR2RGB_BENCH_SHARE_GPU=1 run the
    # N > 1 path with all ranks on one GPU, to exercise everything but RCCL itself on a one-GPU box.
    backend = os.environ.get("IR2RGB_BENCH_BACKEND", "nccl")
    if os.environ.get("IR2RGB_BENCH_SHARE_GPU", "0") == "1":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ir2rgb_amd import conv as C
    from ir2rgb_amd import vid2vid as V
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    # resident_inputs: the synthetic sequence sits in HBM before the timed region and nothing writes it afterwards
    trainer = V.Vid2VidTrainer(dev, world_size=world, seed=0, n_scales_spatial=2, compute_dtype=dtype, resident_inputs=True)
    n_windows = PREROLL + args.warmup + args.steps
    A, B = V.synthetic_sequence(n_windows + 2, H, W, 1234 + 1000 * rank, dev)
    torch.cuda.synchronize()

    def window(i):
        return trainer.train_window(A[:, i:i + 3], B[:, i:i + 3])

    def shape_totals(prof):
        out = {}
        for key, rec in prof.get("shapes", {}).items():
            out[key] = (sum(a.elapsed_time(b) for a, b in rec["events"]) * 1e-3, len(rec["events"]), rec["flops"], rec["kernel"])
        return out

    i = 0
    for k in range(PREROLL + args.warmup):   # untimed: steady-state sequence state + W warm-up steps
        if k == PREROLL + args.warmup - 1:
            torch.cuda.synchronize()
            C.PROFILE = {}                   # last untimed window: every convolution shape bracketed
            t_w = time.perf_counter()
        window(i)
        i += 1
    torch.cuda.synchronize()
    survey_s = time.perf_counter() - t_w
    survey, C.PROFILE = shape_totals(C.PROFILE), None
    # dominant kernel = the convolution shape with the largest total time in that window
    dominant = max(survey, key=lambda kk: survey[kk][0])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    C.PROFILE = {"only": dominant}           # timed region: events around the dominant shape only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        window(i)
        i += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof, C.PROFILE = shape_totals(C.PROFILE), None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    def traffic_of(key):
        """(HBM-side bytes per launch, where they were measured) from the committed PMC passes
        (profiles/r03_dominant_conv_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/conv_hot.py,
        the file names the commit they were taken at); (None, None) for a shape they do not cover.  PMC counters cannot be
        collected inside this process."""
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_dominant_conv_traffic.json")) as f:
                doc = json.load(f)
            v = doc["shapes"].get(",".join(v if isinstance(v, str) else str(int(v)) for v in key))
            return v, (None if v is None else "%s @ commit %s (PMC passes outside this run)" % (doc["source"], doc["commit"]))
        except OSError:
            return None, None

    # ---- roofline of the dominant kernel: HIP events around each of its launches inside the timed region
    tot, n_launch, flops, kernel_name = prof[dominant]
    avg = tot / n_launch
    achieved = flops / avg / 1e12
    traffic, traffic_source = traffic_of(dominant)
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kernel_name + " (MFMA implicit GEMM)",
                "shape": dict(zip(("Cin", "Hin", "Win", "Cout", "kh", "kw", "stride", "reflect", "transposed"), dominant)),
                "launches": n_launch, "avg_us": round(avg * 1e6, 1), "flops_per_launch": flops,
                "note": "launches issued while FlowNet2 replays on the side stream (the generator forward) share the chip and are not bracketed (see DESIGN.md)"}
    peak_mem = torch.cuda.max_memory_allocated(dev)
    all_conv_s = sum(v[0] for v in survey.values())
    all_conv_flops = sum(v[2] * v[1] for v in survey.values())

    # ---- generator-forward timings (north-star roofline config: single-scale composite ngf 128 @512x1024)
    extra = {"bracketed_conv_time_share_of_survey_window": round(all_conv_s / survey_s, 3),
             "bracketed_conv_aggregate_TFLOPs": round(all_conv_flops / all_conv_s / 1e12, 1),
             "max_memory_allocated_GB": round(peak_mem / 1e9, 2)}
    if rank == 0:
        from ir2rgb_amd import networks as N
        opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
        del trainer
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        g = N.build_generator_module(9, 3, 6, 128, "composite", 3, "batch", 0, **opt).to(dev).train()
        g.compute_dtype = dtype
        x, p = torch.tanh(torch.randn(1, 9, H, W, device=dev)), torch.tanh(torch.randn(1, 6, H, W, device=dev))
        with torch.no_grad():
            for _ in range(3):
                g(x, p, None, None, None, None, False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g(x, p, None, None, None, None, False)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        extra["north_star_generator_forward_512x1024"] = {"ms": round(ms, 3), "TFLOPs": round(6.632 / ms * 1e3, 1),
                                                          "frac_of_peak": round(6.632 / ms * 1e3 / PEAK_TFLOPS, 4),
                                                          "algorithmic_TFLOP": 6.632, "issue": "eager (host-bound)"}
        try:    # the same forward replayed from a HIP graph (ir2rgb_amd.graphs): one launch, GPU-bound
            from ir2rgb_amd.graphs import GraphedForward
            gf = GraphedForward(lambda a, b: g(a, b, None, None, None, None, False)[:4], x, p)
            for _ in range(2):
                gf(x, p)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                gf(x, p)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            extra["north_star_generator_forward_512x1024_hip_graph"] = {
                "ms": round(ms, 3), "TFLOPs": round(6.632 / ms * 1e3, 1), "frac_of_peak": round(6.632 / ms * 1e3 / PEAK_TFLOPS, 4)}
        except Exception as e:  # noqa: BLE001  an extra measurement must never take the bench line down
            extra["north_star_generator_forward_512x1024_hip_graph"] = {"error": f"{type(e).__name__}: {e}"}
        del g
        torch.cuda.empty_cache()
        for name, fn in (("config5_two_scale_forward_1024x2048_f16", config5_forward), ("config3_flownet2_operators", config3_operators)):
            try:
                extra[name] = fn(dev)
            except Exception as e:  # noqa: BLE001
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    frames = world * args.steps * 1  # n_frames_load = 1 frame per window per rank
    line = {
        "metric": "generator+discriminator frames/sec at 512x1024", "value": round(frames / elapsed, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "vid2vid training window 512x1024: 2-scale generator (composite ngf128 @256x512 + "
                               "composite-local ngf64 @512x1024, tG=3) + FlowNet2 reference flow + image D (num_D=2) + "
                               "2 temporal D, 3 backward passes + Adam; batch 1 frame per GPU",
                   "parallelism": f"dp{world} (frame-parallel sequences, RCCL gradient all-reduce)"},
        "roofline": roofline, "extra": extra,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(usable_cpus())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 ran the extra forward-only measurement: leave the job together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
