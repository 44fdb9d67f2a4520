#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in the built library, read from the code objects' metadata (no GPU needed).
usage: tools/kernel_resources.py [regex]      columns: vgpr agpr sgpr scratch-bytes static-LDS-bytes max-workgroup name"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_rows(lib=None):
    """[(demangled name, vgpr, agpr, sgpr, scratch bytes, static LDS bytes, max workgroup size)] of every kernel in the library"""
    lib = lib or os.path.join(ROOT, "gtsam-vslam_amd", "libvslam_hip.so")
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(MAGIC, blob)] + [len(blob)]
        rows = []
        for i in range(len(starts) - 1):
            part = os.path.join(td, "b%d.bin" % i); co = os.path.join(td, "b%d.co" % i)
            open(part, "wb").write(blob[starts[i]:starts[i + 1]])
            r = subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co, "--unbundle"],
                               capture_output=True)
            if r.returncode or not os.path.exists(co):
                continue
            notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                def g(k):
                    m = re.search(r"\." + k + r":\s*(\S+)", blk)
                    return m.group(1) if m else "?"
                name = g("name")
                try:
                    name = subprocess.run([LLVM + "llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
                except Exception:      # noqa: BLE001
                    pass
                rows.append((name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("max_flat_workgroup_size")))
        return sorted(rows)


def main():
    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    print("%5s %5s %5s %8s %8s %6s  %s" % ("vgpr", "agpr", "sgpr", "scratch", "lds", "wg", "kernel"))
    for name, v, a, s, p, l, w in kernel_rows():
        if pat is None or pat.search(name):
            print("%5s %5s %5s %8s %8s %6s  %s" % (v, a, s, p, l, w, name))


if __name__ == "__main__":
    main()
