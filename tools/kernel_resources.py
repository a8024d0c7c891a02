#!/usr/bin/env python3
"""Register / LDS / occupancy summary of the kernels of one source file (compiler view, gfx950; ISA left in /tmp/<file>.s):
   python tools/kernel_resources.py remo3d_amd/csrc/patch.hip ['regex on the demangled name']"""
import os, re, subprocess, sys
src = os.path.abspath(sys.argv[1]); filt = sys.argv[2] if len(sys.argv) > 2 else "."
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", src,
                    "-o", "/tmp/%s.s" % os.path.basename(src), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=os.path.dirname(src))
KEYS = [("vgpr", r" VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"TotalSGPRs"), ("occ", r"waves/SIMD\]"), ("sgpr-spill", r"SGPRs Spill"),
        ("vgpr-spill", r"VGPRs Spill"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("lds", r"LDS Size \[bytes/block\]")]
# scratch: bytes of private memory per lane.  Not only spills: an array in registers chosen by a run-time index lands there too (round 4:
# the patch kernel stored 48 bytes per lane for its K sums - 70 MB per application - with "vgpr-spill 0")
for b in re.split(r"remark: Function Name: ", r.stderr)[1:]:
    d = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    d = d.replace("remo::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    d = re.sub(r"\(.*", "", d)[:70]
    if d.startswith("rocprim"):
        continue
    if not re.search(filt, d):
        continue
    vals = []
    for label, key in KEYS:
        m = re.search(key + r": (\d+)", b)
        vals.append("%s %s" % (label, m.group(1) if m else "?"))
    print("%-50s %s" % (d, "  ".join(vals)))
if r.returncode != 0:
    sys.stderr.write(r.stderr[-3000:])
