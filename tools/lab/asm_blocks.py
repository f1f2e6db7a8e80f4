#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a --save-temps .s file: where the MFMAs sit, what shares their blocks.
usage: asm_blocks.py file.s <substring of the kernel's label> [--dump LBBx_y]"""
import re
import sys

s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(s) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l][0]
end = [i for i, l in enumerate(s) if i > start and 's_endpgm' in l][0]
k = s[start:end]
dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == '--dump' else None
cur = 'entry'
cnt, order = {}, []
keys = ('n', 'mfma', 'ds', 'glds', 'vmem', 'bar', 'wait', 'br', 'lane', 'scr', 'salu', 'valu')
for l in k:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1)
    if cur not in cnt:
        cnt[cur] = dict.fromkeys(keys, 0); order.append(cur)
    if m:
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    if dump == cur:
        print(t)
    c = cnt[cur]; c['n'] += 1
    if 'v_mfma' in t: c['mfma'] += 1
    elif t.startswith('ds_'): c['ds'] += 1
    elif 'global_load_lds' in t: c['glds'] += 1
    elif t.startswith('global_') or t.startswith('buffer_'): c['vmem'] += 1
    elif t.startswith('s_barrier'): c['bar'] += 1
    elif t.startswith('s_waitcnt'): c['wait'] += 1
    elif t.startswith('s_cbranch') or t.startswith('s_branch'): c['br'] += 1
    elif 'readlane' in t or 'writelane' in t: c['lane'] += 1
    elif t.startswith('scratch_'): c['scr'] += 1
    elif t.startswith('s_'): c['salu'] += 1
    elif t.startswith('v_'): c['valu'] += 1
if not dump:
    for b in order:
        c = cnt[b]
        if c['n']:
            print(f"{b:12s} " + ' '.join(f"{k_}={c[k_]}" for k_ in keys if c[k_]))
