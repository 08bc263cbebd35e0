#!/usr/bin/env python3
"""Register / LDS use of every kernel in a gfx950 assembly file (hipcc -S --cuda-device-only)."""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
for b in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn).replace('void mq::', '')
    print(f"{dn[:60]:60s} vgpr={g('vgpr_count'):4s} spill={g('vgpr_spill_count')} sgpr={g('sgpr_count'):3s} lds={g('group_segment_fixed_size')}")
