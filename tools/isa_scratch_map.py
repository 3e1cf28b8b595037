"""Where a kernel's scratch instructions sit relative to its loops (CPU: reads hipcc -S output).  For every backward branch
(= a loop [target label, branch]) that contains MFMAs: lines, MFMA count, scratch instructions inside; innermost loops first.
usage: python tools/isa_scratch_map.py file.s kernel_substring"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
for i, l in enumerate(txt):
    m = re.match(r'^(_ZN\S*):', l)
    if not m or key not in m.group(1):
        continue
    end = next(j for j in range(i, len(txt)) if txt[j].startswith('.Lfunc_end'))
    body = txt[i:end]
    scr = [k for k, l2 in enumerate(body) if 'scratch_' in l2]
    mf = [k for k, l2 in enumerate(body) if 'v_mfma' in l2]
    labels = {}
    for k, l2 in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l2)
        if mm:
            labels[mm.group(1)] = k
    loops = []
    for k, l2 in enumerate(body):
        mm = re.search(r'\bs_c?branch\S*\s+(\.LBB\d+_\d+)', l2)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            loops.append((labels[mm.group(1)], k))
    print(m.group(1)[-60:], 'lines', len(body), 'scratch', len(scr), 'mfma', len(mf))
    for a, b in sorted(loops, key=lambda ab: ab[1] - ab[0]):
        nm = sum(1 for k in mf if a <= k <= b)
        if nm == 0:
            continue
        ns = sum(1 for k in scr if a <= k <= b)
        print(f'   loop lines {a}-{b} ({b - a} lines): {nm} mfma, {ns} scratch')
