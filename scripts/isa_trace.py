#!/usr/bin/env python
"""Print the order of memory / MFMA / wait / barrier instructions of selected kernels in a hipcc -S listing
(used to check that every global load of a decode kernel is issued before its first wait).
usage: isa_trace.py file.s substr [substr...]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
pats = sys.argv[2:]
name = None
seq = {}
for l in txt:
    m = re.match(r'^(_ZN3wlx\w+):', l)
    if m:
        name = m.group(1); seq[name] = []; continue
    if name is None: continue
    if l.startswith('\t.section') or l.startswith('.Lfunc_end'):
        name = None; continue
    s = l.strip()
    m = re.match(r'(global_load\w+|global_store\w+|global_atomic\w+|s_load\w+|v_mfma\w+|s_barrier|ds_read\w+|ds_write\w+|ds_bpermute\w+|s_cbranch\w+|s_endpgm|scratch_\w+)', s)
    if m: seq[name].append(m.group(1)); continue
    if s.startswith('s_waitcnt'):
        seq[name].append(s.split(';')[0].strip().replace('s_waitcnt ', 'W:'))
for n, sq in seq.items():
    if not any(p in n for p in pats): continue
    out = []; prev = None; cnt = 0
    for x in sq:
        if x == prev: cnt += 1
        else:
            if prev: out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
            prev = x; cnt = 1
    if prev: out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
    print(n); print('  ' + ' | '.join(out)); print()
