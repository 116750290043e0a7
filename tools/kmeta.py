#!/usr/bin/env python3
"""Registers / spills / scratch / LDS of the kernels of a device assembly listing: tools/kmeta.py k.s [name-substring]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", txt, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} sgpr {g('sgpr_count'):>4s}")
