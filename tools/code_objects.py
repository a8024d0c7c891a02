#!/usr/bin/env python3
"""What the BUILT library holds, kernel by kernel (registers, LDS, scratch): the gfx950 code objects are cut out of the library's
.hip_fatbin section (clang offload bundles, one per translation unit) and their metadata notes read with llvm-readelf.
   python tools/code_objects.py [remo3d_amd/libremo3d_hip.so] [regex on the demangled name]
tests/test_abi_cpu.py uses kernels() to assert that no kernel of ours touches scratch memory: round 4 found the patch kernel storing
48 bytes per lane there (an array of K sums chosen by a run-time index) - 70 MB per application at the headline size, with the
compiler's summary saying "vgpr-spill 0"."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name):
    out = subprocess.run([READELF, "-S", "-W", path], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        m = re.search(r"\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
        if m and m.group(1) == name:
            return int(m.group(3), 16), int(m.group(4), 16)
    raise RuntimeError("no section %s in %s" % (name, path))


def code_objects(path):
    """[(target triple, bytes of the code object)] of every bundle in the library's .hip_fatbin section"""
    off, size = _section(path, ".hip_fatbin")
    with open(path, "rb") as f:
        f.seek(off)
        blob = f.read(size)
    found = []
    at = blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            o, s, t = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + t].decode()
            p += 24 + t
            if s and "amdgcn" in triple:
                found.append((triple, blob[at + o:at + o + s]))
        at = blob.find(MAGIC, at + len(MAGIC))
    return found


def kernels(path):
    """[{name, vgpr, sgpr, lds, scratch, triple}] of every kernel of every gfx950 code object in the library"""
    rows = []
    for triple, data in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(data)
            tmp.flush()
            notes = subprocess.run([READELF, "--notes", tmp.name], capture_output=True, text=True, check=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            def num(key):
                m = re.search(r"\.%s:\s+(\d+)" % key, block)
                return int(m.group(1)) if m else -1
            m = re.search(r"\.name:\s+(\S+)", block)
            if m:
                rows.append(dict(name=m.group(1), vgpr=num("vgpr_count"), sgpr=num("sgpr_count"), lds=num("group_segment_fixed_size"),
                                 scratch=num("private_segment_fixed_size"), triple=triple))
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", d.replace("void ", "").replace("(anonymous namespace)::", "")) for d in out]


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "remo3d_amd", "libremo3d_hip.so")
    filt = sys.argv[2] if len(sys.argv) > 2 else "."
    ks = kernels(lib)
    for k, d in zip(ks, demangle([k["name"] for k in ks])):
        if re.search(filt, d):
            print("%-90s vgpr %3d  sgpr %3d  lds %6d  scratch %4d" % (d[:90], k["vgpr"], k["sgpr"], k["lds"], k["scratch"]))
    print("%d kernels, %d with scratch" % (len(ks), sum(1 for k in ks if k["scratch"] > 0)))
