"""Compact per-instruction trace of the MFMA main loop of one kernel in a gfx950 .s file.

  hipcc -std=c++17 -O3 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/h2.s mqgan_amd/csrc/igemm_h2.hip
  python tools/loop_trace.py /tmp/h2.s conv_h2_kernelILi4ELi2ELi64ELb0ELb1E

M = MFMA, r = ds_read, D = buffer_load (LDS-DMA), [..] = s_waitcnt, |B| = s_barrier, v / s = other vector / scalar work.
"""
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_Z\w*' + re.escape(key) + r'\w*):.*?\n(.*?)\n\s*s_endpgm', text, re.S | re.M)
body = [l.strip() for l in m.group(2).split('\n')]
blocks, cur = [], []
for l in body:
    if not l or l.startswith(';') or (l.startswith('.') and not l.startswith('.LBB')):
        continue
    if l.startswith('.LBB') or l.endswith(':'):
        blocks.append(cur)
        cur = []
    else:
        cur.append(l)
blocks.append(cur)
best = max(blocks, key=lambda b: sum('mfma' in x for x in b))
out = []
for l in best:
    op = l.split()[0]
    if 'mfma' in op: out.append('M')
    elif op.startswith('ds_read'): out.append('r')
    elif op.startswith('ds_'): out.append('w')
    elif op.startswith('buffer_load'): out.append('D')
    elif op.startswith('s_waitcnt'): out.append('[' + l.split(None, 1)[1] + ']')
    elif op.startswith('s_barrier'): out.append('|B|')
    elif op.startswith('s_nop'): out.append('n')
    elif op.startswith('v_'): out.append('v')
    elif op.startswith('s_'): out.append('s')
    else: out.append('?' + op)
print(m.group(1))
print(len(best), 'instructions,', sum('mfma' in x for x in best), 'MFMAs')
print(''.join(out))
