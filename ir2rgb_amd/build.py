"""Build recipe for libir2rgb_hip.so (hipcc, gfx950 only, no torch involved).

    python -m ir2rgb_amd.build            # incremental
    python -m ir2rgb_amd.build --force

Each csrc/*.hip is compiled to an object under ir2rgb_amd/lib/obj/ (in parallel) and linked
into ir2rgb_amd/lib/libir2rgb_hip.so.  The library is built in-tree so that it travels to
the GPU box with the repository snapshot; it is git-ignored.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libir2rgb_hip.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++20", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
            "-I", INCLUDE]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libir2rgb_hip.so cannot be built")
    return exe


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    hdrs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, obj, extra):
    cmd = [hipcc(), *CXXFLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {os.path.basename(src)}:\n{r.stderr}")
    return r.stderr


def build(force=False, verbose=False, jobs=None):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hmt = _deps_mtime()
    todo, objs = [], []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJDIR, f[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hmt):
            todo.append((src, obj))
    extra = ["-Rpass-analysis=kernel-resource-usage"] if verbose else []
    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs or min(6, len(todo))) as ex:
            for (src, _), log in zip(todo, ex.map(lambda so: _compile(so[0], so[1], extra), todo)):
                if verbose:
                    sys.stderr.write(log)
    if todo or not os.path.exists(LIB) or force:
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    from . import fastbind        # the CPython fastcall bindings of the same entry points (gcc, host code only)
    fastbind.build(force=force)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
