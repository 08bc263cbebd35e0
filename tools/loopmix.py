#!/usr/bin/env python3
"""Instruction mix of the hottest basic block (most MFMAs) of one kernel in an assembly listing.
usage: tools/loopmix.py file.s <mangled-name-substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
print(m.group(1))
blocks = re.split(r'\n(\.LBB\d+_\d+):', m.group(2))
best = max(((blocks[i + 1].count('v_mfma'), blocks[i], blocks[i + 1]) for i in range(1, len(blocks), 2)))
c = collections.Counter()
for l in best[2].split('\n'):
    l = l.strip()
    if l and not l.startswith(';') and not l.startswith('.'):
        c[l.split()[0]] += 1
valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
print(best[1], 'mfma', best[0], 'total', sum(c.values()), 'valu(non-mfma)', valu, 'salu', sum(v for k, v in c.items() if k.startswith('s_')))
print(' '.join(f'{k}:{v}' for k, v in c.most_common(40)))
