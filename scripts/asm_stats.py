#!/usr/bin/env python3
"""Per-function instruction statistics of a gfx950 assembly listing (hipcc --cuda-device-only -S): instructions, VALU, LDS, scratch
(spill) accesses, and the spill accesses that sit inside loops (between a label and a backward branch to it).
usage: scripts/asm_stats.py file.s"""
import re, sys
cur = None; fn = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+|\w+):\s*(;.*)?$', line)
    if m and not line.startswith('.L'):
        cur = m.group(1); fn[cur] = []; continue
    if cur is None: continue
    s = line.strip()
    if s.startswith('.L') and s.endswith(':') or re.match(r'^\.LBB\S+:', s): fn[cur].append(('label', s.split(':')[0])); continue
    if not s or s.startswith(('.', ';')): continue
    fn[cur].append(('ins', s))
for name, items in fn.items():
    ins = [s for k, s in items if k == 'ins']
    if len(ins) < 100: continue
    labels = {}; loops = []
    for idx, (k, s) in enumerate(items):
        if k == 'label': labels[s] = idx
        elif s.startswith(('s_cbranch', 's_branch')):
            t = s.split()[-1]
            if t in labels: loops.append((labels[t], idx))
    inloop = set()
    for a, b in loops: inloop.update(range(a, b + 1))
    scr = [i for i, (k, s) in enumerate(items) if k == 'ins' and 'scratch_' in s]
    print(f"{name[:64]:66s} ins {len(ins):6d}  valu {sum(1 for s in ins if s.startswith('v_')):6d}  ds {sum(1 for s in ins if s.startswith('ds_')):4d}  scratch {len(scr):4d}  in loops {sum(1 for i in scr if i in inloop):4d}")

# --loops NAME: list the loops of the functions whose name contains NAME
if len(sys.argv) > 3 and sys.argv[2] == '--loops':
    for name, items in fn.items():
        if sys.argv[3] not in name: continue
        labels = {}
        for idx, (k, s) in enumerate(items):
            if k == 'label': labels[s] = idx
            elif s.startswith(('s_cbranch', 's_branch')):
                t = s.split()[-1]
                if t in labels:
                    body = [x for kk, x in items[labels[t]:idx + 1] if kk == 'ins']
                    print(f"  loop {t:12s} ins {len(body):5d} valu {sum(1 for x in body if x.startswith('v_')):5d} f64 {sum(1 for x in body if '_f64' in x):4d} ds {sum(1 for x in body if x.startswith('ds_')):3d} "
                          f"global {sum(1 for x in body if x.startswith('global_')):3d} scratch {sum(1 for x in body if 'scratch_' in x):3d} waitcnt {sum(1 for x in body if x.startswith('s_waitcnt')):3d}")
