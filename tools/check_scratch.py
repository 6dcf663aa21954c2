"""Compile every .hip file of ir2rgb_amd/csrc for gfx950 with -save-temps and list the kernels that use a
private segment (scratch) or spill VGPRs.  A kernel with scratch pays for it at every dispatch (DESIGN.md,
"Small kernels are latency-shaped"); the library is meant to have none.  CPU only (hipcc cross-compiles)."""
import glob, os, re, subprocess, sys, tempfile

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ir2rgb_amd", "csrc")
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for src in sorted(glob.glob(os.path.join(root, "*.hip"))):
        name = os.path.splitext(os.path.basename(src))[0]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o",
                        os.path.join(tmp, name + ".o"), "-save-temps=obj"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        n = 0
        for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)(?:.|\n)*?\.vgpr_count:\s+(\d+)\n"
                             r"\s+\.vgpr_spill_count:\s+(\d+)", asm):
            n += 1
            if int(m.group(2)) or int(m.group(4)):
                bad += 1
                print("%s: %s scratch %s B, %s VGPRs, %s spilled" % (name, m.group(1)[:80], m.group(2), m.group(3), m.group(4)))
        print("%-16s %3d kernels checked" % (name, n), flush=True)
print("kernels with scratch:", bad)
sys.exit(1 if bad else 0)
