"""Root-cause probe for the two-stream first-forward hazard (replaces tools/branch_race*.py).

A FRESH composite-local generator runs its first forward with the two branches on two HIP streams; the result is compared
with the one-stream forward of a deep copy.  Modes (argv[1]):

  base    nothing is kept alive or added: after the forward (device idle) every packed-weight buffer the forward created is
          compared with a re-pack of the same fp32 parameter -- tells a buffer that is WRONG IN MEMORY from a buffer that
          was right in memory but read wrongly.
  trace   every kernel-level call of the forward (pack, convolution, BatchNorm finalize / apply) keeps references to its
          operands and results; afterwards each call is re-executed alone from the recorded operands and compared with what
          the two-stream forward produced -- names the first kernel whose output does not follow from its inputs.

    python tools/stream_hazard.py base|trace [iterations] [H W]
"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ir2rgb_amd import autograd as A  # noqa: E402,F401
from ir2rgb_amd import conv as C  # noqa: E402
from ir2rgb_amd import layers as L  # noqa: E402
from ir2rgb_amd import networks as N  # noqa: E402

mode = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 2048)
dev = torch.device("cuda:0")
opt = dict(gen_blocks=9, n_blocks_local=3, fg=False, no_flow=False, n_local_enhancers=1, feat_num=3)
gen = torch.Generator().manual_seed(1)
Ain, P = torch.rand(1, 9, H, W, generator=gen).to(dev), torch.rand(1, 6, H, W, generator=gen).to(dev)
mk = lambda: torch.rand(1, 128, H // 2, W // 2, generator=gen).to(dev).half().contiguous(memory_format=torch.channels_last)  # noqa: E731
fi, ff = mk(), mk()
names6 = ["final", "flow", "weight", "raw", "img_feat", "flow_feat"]
main_stream = torch.cuda.current_stream(dev).cuda_stream

REC = None


def where(t):
    return "main" if torch.cuda.current_stream(dev).cuda_stream == main_stream else "side"


orig = dict(pack=C.pack_weight, conv=C.conv2d_fwd, fin=L.bn_finalize, finapp=L.bn_finalize_apply, app=L.bn_apply)


def pack_spy(desc, weight, adjoint=False):
    out = orig["pack"](desc, weight, adjoint)
    if REC is not None:
        REC.append(("pack", where(out), (C.ConvDesc.from_buffer_copy(desc), weight, adjoint), (out,)))
    return out


def conv_spy(desc, x, wpacked, bias=None, want_stats=False, out=None):
    y, st = orig["conv"](desc, x, wpacked, bias, want_stats, out)
    if REC is not None:
        REC.append(("conv", where(y), (C.ConvDesc.from_buffer_copy(desc), x, wpacked, bias, want_stats), (y, st)))
    return y, st


def fin_spy(stats, count, bn, training=True, conv_bias=None, outs=None):
    r = orig["fin"](stats, count, bn, training, conv_bias, outs)
    if REC is not None:
        REC.append(("bn_finalize", where(r[0]), (stats, count, bn, training, conv_bias), tuple(r)))
    return r


def finapp_spy(stats, count, bn, y, act, res1=None, res2=None, conv_bias=None, out=None, outs=None):
    r = orig["finapp"](stats, count, bn, y, act, res1, res2, conv_bias, out, outs)
    if REC is not None:
        REC.append(("bn_finalize_apply", where(r[0]), (stats, count, bn, y, act, res1, res2, conv_bias), tuple(r)))
    return r


def app_spy(x, scale, shift, act=L.ACT_NONE, res1=None, res2=None, out=None):
    r = orig["app"](x, scale, shift, act, res1, res2, out)
    if REC is not None:
        REC.append(("bn_apply", where(r), (x, scale, shift, act, res1, res2), (r,)))
    return r


if mode == "trace":
    C.pack_weight, C.conv2d_fwd = pack_spy, conv_spy
    L.bn_finalize, L.bn_finalize_apply, L.bn_apply = fin_spy, finapp_spy, app_spy


def describe(a, b):
    ne = (a != b)
    idx = ne.nonzero()
    d = (a.float() - b.float()).abs()
    ch = idx[:, 1] if a.dim() == 4 else idx[:, -1]
    return "%d of %d differ, max |d| %.4g, channel/last-axis range [%d, %d], first %s last %s" % (
        int(ne.sum()), a.numel(), float(d.max()), int(ch.min()), int(ch.max()), idx[0].tolist(), idx[-1].tolist())


def replay(rec):
    """Re-execute one recorded call on the (idle) device from its recorded operands; -> list of (label, got, again)."""
    kind, _, args, outs = rec
    if kind == "pack":
        desc, w, adj = args
        return [("packed", outs[0], orig["pack"](desc, w, adj))]
    if kind == "conv":
        desc, x, wp, bias, ws = args
        y2, st2 = orig["conv"](desc, x, wp, bias, ws)
        r = [("y", outs[0], y2)]
        if st2 is not None:
            r.append(("stats", outs[1], st2))
        return r
    if kind == "bn_finalize":
        stats, count, bn, tr, cb = args
        r2 = orig["fin"](stats, count, copy.deepcopy(bn) if isinstance(bn, torch.nn.Module) else bn, tr, cb)
        return [(n, a, b) for n, a, b in zip(("scale", "shift", "mean", "invstd"), outs, r2)]
    if kind == "bn_finalize_apply":
        stats, count, bn, y, act, r1, r2_, cb = args
        r2 = orig["finapp"](stats, count, copy.deepcopy(bn) if isinstance(bn, torch.nn.Module) else bn, y, act, r1, r2_, cb)
        return [(n, a, b) for n, a, b in zip(("z", "scale", "shift", "mean", "invstd"), outs, r2)]
    if kind == "bn_apply":
        x, sc, sh, act, r1, r2_ = args
        return [("z", outs[0], orig["app"](x, sc, sh, act, r1, r2_))]
    return []


print("mode", mode, "size", H, W, "env", {k: v for k, v in os.environ.items() if k.startswith(("AMD_", "HIP_FORCE", "DEBUG_CLR", "GPU_"))},
      flush=True)
bad_iters = 0
for it in range(iters):
    torch.manual_seed(0)
    ga = N.build_generator_module(9, 3, 6, 64, "composite-local", 3, "batch", 1, **opt).to(dev).train()
    gb = copy.deepcopy(ga)
    ga.compute_dtype = gb.compute_dtype = torch.float16
    names = {m: n for n, m in ga.named_modules()}
    torch.cuda.synchronize()
    REC = [] if mode == "trace" else None
    N.BRANCH_STREAMS = "1"
    with torch.no_grad():
        oa = [t for t in ga(Ain, P, None, fi, ff, None, False)[:6]]
    torch.cuda.synchronize()
    rec, REC = REC, None
    N.BRANCH_STREAMS = "0"
    with torch.no_grad():
        ob = [t for t in gb(Ain, P, None, fi, ff, None, False)[:6]]
    torch.cuda.synchronize()
    diff = {n: int((a != b).sum()) for n, a, b in zip(names6, oa, ob) if not torch.equal(a, b)}
    bad_iters += bool(diff)
    print("iter", it, "two-stream vs one-stream, elements differing:", diff, flush=True)
    # (1) packed buffers in memory, against a re-pack now and against the one-stream twin's buffers
    twin = dict(gb.named_modules())
    nbuf = nbad = 0
    for m, n in names.items():
        for tag, hit in m.__dict__.get("_ir2rgb_packed", {}).items():
            other = twin[n].__dict__.get("_ir2rgb_packed", {}).get(tag)
            if other is None:
                continue
            nbuf += 1
            if not torch.equal(hit[1], other[1]):
                nbad += 1
                print("   PACKED BUFFER WRONG IN MEMORY:", n, tag, describe(hit[1].view(1, -1), other[1].view(1, -1)), flush=True)
    print("   packed buffers checked:", nbuf, "wrong in memory:", nbad, flush=True)
    # (2) replay of every recorded call
    if rec is not None:
        nwrong = 0
        for i, r in enumerate(rec):
            for label, got, again in replay(r):
                if got is None or again is None:
                    continue
                if not torch.equal(got, again):
                    nwrong += 1
                    if nwrong <= 12:
                        who = ""
                        if r[0] in ("bn_finalize", "bn_finalize_apply"):
                            who = names.get(r[2][2], "?")
                        print("   call %d %-18s on %s %s: output '%s' does not follow from its recorded inputs: %s" % (
                            i, r[0], r[1], who, label, describe(got, again)), flush=True)
        torch.cuda.synchronize()
        print("   calls recorded:", len(rec), "outputs that do not replay:", nwrong, flush=True)
    del ga, gb, oa, ob, rec
print("iterations with a mismatch:", bad_iters, "of", iters)
