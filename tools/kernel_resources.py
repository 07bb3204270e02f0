#!/usr/bin/env python3
"""VGPRs / spills / scratch / LDS of the kernels in the built library, from the code objects' metadata notes.
    python tools/kernel_resources.py [substring of the demangled kernel name ...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_code_audit import device_code_objects, LIB  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    pats = sys.argv[1:]
    for i, (triple, blob) in enumerate(device_code_objects(LIB)):
        if "gfx950" not in triple or not blob:
            continue
        path = "/tmp/kernel_resources_%d.co" % i
        open(path, "wb").write(blob)
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem)
            if pats and not any(p in dem for p in pats):
                continue
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
            print("%-70s vgpr %3d  agpr %s  spilled %3d  scratch %5d B  lds %6d B" % (
                dem[:70], g("vgpr_count"), blk.split()[0], g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
