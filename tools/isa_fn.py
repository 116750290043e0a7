#!/usr/bin/env python3
"""Instruction mix of one kernel of an ISA listing: tools/isa_fn.py k.s <mangled-name-substring> [grep-regex]
(k.s from `hipcc ... --cuda-device-only -S`)."""
import re
import sys
from collections import Counter


def body(txt, name):
    m = re.search(r"^(_Z\w*%s\w*):.*?\n(.*?)\n\.Lfunc_end" % re.escape(name), txt, re.S | re.M)
    return m.group(2).split("\n")


def main():
    txt = open(sys.argv[1]).read()
    lines = body(txt, sys.argv[2])
    ins = [l.strip().split()[0] for l in lines if re.match(r"\s+[a-z]\w+", l) and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    print(len(ins), "instructions")
    for k, v in c.most_common(40):
        print(f"{v:6d} {k}")
    if len(sys.argv) > 3:
        pat = re.compile(sys.argv[3])
        for l in lines:
            if pat.search(l):
                print(l.rstrip())


if __name__ == "__main__":
    main()
