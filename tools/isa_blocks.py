#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S assembly file (tools/isa_stats.py with KEEP_S=1 prints its path).
usage: tools/isa_blocks.py k.s <name pattern> [min instructions]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read(); pat = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for m in re.finditer(r'^(_Z\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    if pat not in m.group(1): continue
    blk, cur = [], ['entry', []]
    for l in m.group(2).split('\n'):
        t = l.strip()
        if not t or t.startswith((';', '.p2align', '.set', '.size', '.section', '.type')): continue
        if re.match(r'^\.?[A-Za-z_0-9$]+:', t):
            blk.append(cur); cur = [t.split(':')[0], []]; continue
        if t.startswith('.'): continue
        cur[1].append(t.split()[0])
    blk.append(cur)
    for name, ins in blk:
        if len(ins) < mn: continue
        c = Counter(ins)
        g = lambda *ks: sum(v for k, v in c.items() if any(k.startswith(x) for x in ks))
        print(f"{name:12s} n={len(ins):5d} mfma={g('v_mfma'):3d} exp={g('v_exp'):3d} acc_mov={g('v_accvgpr'):4d} scratch={g('scratch_'):3d} ds_read={g('ds_read'):3d} "
              f"ds_write={g('ds_write'):3d} valu={g('v_') - g('v_mfma', 'v_accvgpr'):4d} s_nop={g('s_nop'):3d} waitcnt={g('s_waitcnt'):3d} buf={g('buffer_'):2d} barrier={g('s_barrier')}")
    break
