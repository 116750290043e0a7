#!/usr/bin/env python3
"""Which s_waitcnt vmcnt(N) sit inside loops of the kernels of a device assembly listing?  (a vmcnt(0) in a loop that also
issues loads for a LATER iteration means the prefetch is drained every trip)   tools/vmwaits.py /tmp/k.s [name-regex]"""
import collections
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)^\.Lfunc_end', txt, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        if not pat.search(name):
            continue
        inloop = False
        waits = collections.Counter()
        loads = 0
        for l in m.group(2).split('\n'):
            s = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', s):
                inloop = 'in Loop' in l or 'Loop Header' in l
            elif inloop and s.startswith('s_waitcnt') and 'vmcnt' in s:
                waits[re.search(r'vmcnt\((\d+)\)', s).group(1)] += 1
            elif inloop and s.startswith(('global_load', 'buffer_load')):
                loads += 1
        if waits:
            print(f"{name[:90]:90s} loads in loops {loads:3d}  vmcnt waits in loops {dict(sorted(waits.items(), key=lambda kv: int(kv[0])))}")


if __name__ == "__main__":
    main()
