"""Debug check of cross-stream ordering for the buffers this package CACHES across calls (IR2RGB_STREAM_CHECK=1).

The trainer runs on up to three HIP streams (main; FlowNet2 ahead of it; the generators' second branch; optionally weight
gradients), and everything that outlives one call -- packed weights, gather plans, unit constants, split-K workspaces,
BatchNorm running statistics, frame histories, kept reference flows -- is written on one stream and may be read on another.
torch's allocator protects a tensor's MEMORY across streams (record_stream); nothing protects its CONTENTS: that is the
hand-placed wait_stream / wait_event calls.  This module tracks them.

Happens-before by vector clocks, on the host, from the calls this process makes (no device queries):
  * every stream has a production counter and a clock {producer stream: highest production it is ordered after};
  * ``produced(t, what)``: the kernel just queued on the current stream writes cached buffer ``t`` -> tag (stream, count);
  * ``consumed(t)``: the kernel about to be queued on the current stream reads ``t``; if its tag belongs to another stream
    and the current stream's clock has not reached it, raise StreamOrderError;
  * Stream.wait_stream / Stream.wait_event / Event.record / synchronize calls (patched while the check is on) move the clocks.
A HIP graph replay is one node on the stream it is launched on; tags made during its capture carry the capture stream.

Off (the default) every entry point is a constant-false test.  ``enable()`` / ``disable()`` switch it at run time
(tests/test_streams_gpu.py runs sixteen training windows under it, and plants a violation to see it caught).
"""
import os

import torch

ENABLED = False
_seq = {}        # stream id -> productions so far
_clock = {}      # stream id -> {stream id: count seen}
_tags = {}       # (device index, data_ptr) -> (stream id, count, what)
_patched = []
STATS = {"produced": 0, "consumed": 0, "cross_stream": 0}


class StreamOrderError(RuntimeError):
    pass


def _sid(stream):
    return (stream.device.index if hasattr(stream, "device") else torch.cuda.current_device(), int(stream.cuda_stream))


def _cur(device=None):
    return _sid(torch.cuda.current_stream(device))


def _merge(dst, src):
    for k, v in src.items():
        if dst.get(k, 0) < v:
            dst[k] = v


def _view_of(sid):
    """What a stream has seen: its clock, what the host had waited for, and its own productions."""
    c = dict(_HOST)
    _merge(c, _clock.get(sid, {}))
    c[sid] = _seq.get(sid, 0)
    return c


def produced(t, what=""):
    if not ENABLED or t is None or not t.is_cuda:
        return
    sid = _cur(t.device)
    _seq[sid] = _seq.get(sid, 0) + 1
    _tags[(t.device.index, t.data_ptr())] = (sid, _seq[sid], what)
    STATS["produced"] += 1


def consumed(t, what=""):
    if not ENABLED or t is None or not t.is_cuda:
        return
    tag = _tags.get((t.device.index, t.data_ptr()))
    if tag is None:
        return
    STATS["consumed"] += 1
    sid = _cur(t.device)
    if tag[0] == sid:
        return
    STATS["cross_stream"] += 1
    seen = max(_clock.get(sid, {}).get(tag[0], 0), _HOST.get(tag[0], 0))
    if seen < tag[1]:
        raise StreamOrderError(
            f"un-ordered cross-stream use of a cached buffer: '{tag[2] or what}' was written on stream {tag[0]} "
            f"(production {tag[1]}) and is read on stream {sid}, which is only ordered after production "
            f"{seen} of that stream (no wait_stream / wait_event in between)")


def forget(t):
    if ENABLED and t is not None and t.is_cuda:
        _tags.pop((t.device.index, t.data_ptr()), None)


def _patch():
    S, E = torch.cuda.Stream, torch.cuda.Event
    o_wait_stream, o_wait_event, o_record, o_ssync, o_esync, o_sync = (S.wait_stream, S.wait_event, E.record, S.synchronize,
                                                                       E.synchronize, torch.cuda.synchronize)

    def wait_stream(self, other):
        _merge(_clock.setdefault(_sid(self), {}), _view_of(_sid(other)))
        return o_wait_stream(self, other)

    def wait_event(self, event):
        _merge(_clock.setdefault(_sid(self), {}), getattr(event, "_ir2rgb_clock", {}))
        return o_wait_event(self, event)

    def record(self, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream()
        self._ir2rgb_clock = _view_of(_sid(st))
        return o_record(self, st) if stream is not None else o_record(self)

    def host_waited(view):
        # the host has waited: everything it queues from now on, on any stream, is ordered after ``view``
        for sid in set(_seq) | set(_clock):
            _merge(_clock.setdefault(sid, {}), view)
        _HOST.update({k: max(v, _HOST.get(k, 0)) for k, v in view.items()})

    def ssync(self):
        r = o_ssync(self)
        host_waited(_view_of(_sid(self)))
        return r

    def esync(self):
        r = o_esync(self)
        host_waited(getattr(self, "_ir2rgb_clock", {}))
        return r

    def sync(device=None):
        r = o_sync(device)
        allv = {}
        for sid in set(_seq) | set(_clock):
            _merge(allv, _view_of(sid))
        host_waited(allv)
        return r

    S.wait_stream, S.wait_event, E.record, S.synchronize, E.synchronize, torch.cuda.synchronize = (
        wait_stream, wait_event, record, ssync, esync, sync)
    _patched[:] = [(S, "wait_stream", o_wait_stream), (S, "wait_event", o_wait_event), (E, "record", o_record),
                   (S, "synchronize", o_ssync), (E, "synchronize", o_esync), (torch.cuda, "synchronize", o_sync)]


_HOST = {}       # what the host has waited for: a stream created later starts from here


def enable():
    global ENABLED
    if not ENABLED:
        ENABLED = True
        _seq.clear(); _clock.clear(); _tags.clear(); _HOST.clear()
        for k in STATS:
            STATS[k] = 0
        _patch()


def disable():
    global ENABLED
    if ENABLED:
        ENABLED = False
        for obj, name, fn in _patched:
            setattr(obj, name, fn)
        _patched.clear()


if os.environ.get("IR2RGB_STREAM_CHECK", "0") != "0":
    enable()
