"""Static check of the LDS-DMA pipelines' write-after-read safety (the two-stream hazard).

The kernels restage an LDS ring stage one barrier after its last fragment read.  That is only safe when the reading wave
has WAITED for those reads (s_waitcnt lgkmcnt) before it arrives at the barrier: a raw ``s_barrier`` is not a fence, and
hipcc is free to sink the wait (and the MFMAs that need the data) below it -- the read is then still in flight when another
wave's ``buffer_load ... lds`` overwrites the stage.  (It did exactly that in the 2-stage form of conv_igemm_kernel and in
the plain form of conv3x3_patch_kernel; the stale operands showed only with a second stream's LDS-heavy kernel resident on
the same CU, which slows the reads enough to lose the race.)

This walks the gfx950 code of every kernel that uses LDS-DMA and reports each ``s_barrier`` a wave can reach with LDS
reads outstanding.  Input: the objects the build left under ir2rgb_amd/lib/obj (default), or assembly / object files.

    python tools/check_lds_war.py [file.o | file.s ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
LABEL = re.compile(r"^(?:[0-9a-f]+ <)?(_Z[\w$.]+)>?:")


def scan_lines(lines):
    """-> (number of LDS-DMA kernels, {kernel: [(line number, reads outstanding), ...]})."""
    kernels, findings = 0, {}
    name, uses_dma, out, hits = None, False, 0, []

    def close():
        nonlocal kernels
        if name is not None and uses_dma:
            kernels += 1
            if hits:
                findings[name] = hits

    for ln, line in enumerate(lines, 1):
        m = LABEL.match(line)
        if m:
            close()
            name, uses_dma, out, hits = m.group(1), False, 0, []
            continue
        s = line.strip()
        if name is None or not s or s[0] in ";.":
            continue
        s = s.split("//")[0].strip()
        op = s.split()[0] if s else ""
        if op.startswith(("ds_read", "ds_load")):
            out += 1
        elif op == "s_waitcnt":
            m = LGKM.search(s)
            if m:
                out = min(out, int(m.group(1)))
            elif "cnt" not in s:
                out = 0     # raw immediate form: treat as a full wait
        elif op.startswith(("buffer_load", "global_load")) and s.endswith("lds"):
            uses_dma = True
        elif op == "s_barrier":
            if out > 0:
                hits.append((ln, out))
        elif op == "s_endpgm":
            out = 0
    close()
    return kernels, findings


def device_asm(obj, tmp):
    """Disassembly of the gfx950 code object bundled in a host object built by hipcc."""
    base = os.path.join(tmp, os.path.basename(obj))
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, base + ".fat"])
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={base}.fat",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={base}.co"], capture_output=True, text=True)
    if r.returncode != 0:
        if "Can't find bundles" in r.stderr:      # a source file without device code
            return []
        raise RuntimeError(r.stderr)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", base + ".co"], check=True,
                          capture_output=True, text=True).stdout.splitlines()


def check(paths=None):
    """-> list of (file, kernels with LDS-DMA, {kernel: hits})."""
    if not paths:
        objdir = os.path.join(ROOT, "ir2rgb_amd", "lib", "obj")
        paths = sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith(".o"))
    res = []
    with tempfile.TemporaryDirectory(prefix="ldswar") as tmp:
        for p in paths:
            lines = open(p).read().splitlines() if p.endswith(".s") else device_asm(p, tmp)
            k, f = scan_lines(lines)
            res.append((os.path.basename(p), k, f))
    return res


def main(paths):
    bad = 0
    for fname, k, f in check(paths):
        if k:
            print(f"{fname}: {k} LDS-DMA kernels, {len(f)} with LDS reads outstanding at a barrier")
        for name, hs in sorted(f.items()):
            bad += 1
            print("   ", name[:150], "->", ", ".join(f"line {ln} ({n} reads)" for ln, n in hs[:6]), "..." if len(hs) > 6 else "")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
