#!/usr/bin/env python3
"""Instruction mix of the big basic blocks of one kernel in a device assembly listing:
   tools/loopmix.py /tmp/k.s <mangled-name-regex> [min-block-size]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    pat = sys.argv[2]
    lim = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    for m in re.finditer(r'^(' + pat + r'):.*?\n(.*?)^\.Lfunc_end', txt, re.S | re.M):
        print(m.group(1))
        blocks = []
        cur = ('entry', [])
        for l in m.group(2).split('\n'):
            s = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', s):
                blocks.append(cur)
                cur = (s.split(':')[0], [])
            elif s and not s.startswith(';') and not s.startswith('.'):
                cur[1].append(s)
        blocks.append(cur)
        for name, ins in blocks:
            if len(ins) < lim:
                continue
            c = collections.Counter()
            for i in ins:
                op = i.split()[0]
                if re.match(r'v_(exp|rcp|log|sqrt|rsq|sin|cos)', op): c['trans'] += 1
                elif op.startswith('v_pk'): c['v_pk'] += 1
                elif op.startswith('v_mfma'): c['mfma'] += 1
                elif op.startswith('v_accvgpr'): c['accvgpr'] += 1
                elif op.startswith('v_'): c['valu'] += 1
                elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
                elif op.startswith('s_nop'): c['nop'] += 1
                elif op.startswith('s_'): c['salu'] += 1
                elif op.startswith('ds_'): c['lds'] += 1
                elif op.startswith(('global', 'buffer', 'scratch', 'flat')): c['vmem'] += 1
                else: c[op] += 1
            print(f"  {name:12s} {len(ins):5d}  {dict(c)}")


if __name__ == "__main__":
    main()
