#!/usr/bin/env python3
"""Register / LDS / scratch report of the device code of one kernel source: tools/kregs.py ss2d_chan.hip [filter-regex].
Cross-compiles the device side only (no GPU needed) and reads the metadata notes of the code object."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    src = sys.argv[1]
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "xfmamba_amd", "csrc")
    extra = ["-DXFM_CHAN_N16", "-mllvm", "-amdgpu-mfma-vgpr-form"] if src in ("ss2d_chan.hip", "ss2d_chan1.hip") else []
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", *extra,
                        "--cuda-device-only", "-S", src, "-o", out], cwd=here, check=True)
        txt = open(out).read()
    keys = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        val = {k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1] for k in keys + ("name",)}
        name = subprocess.run(["c++filt", val["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        if pat.search(name):
            print("vgpr %4s agpr %4s sgpr %4s spill %4s scratch %5s lds %6s  %s" % (*[val[k] for k in keys], name))


if __name__ == "__main__":
    main()
