#!/usr/bin/env python3
"""One-letter-per-instruction trace of the k-loop of a kernel in a hipcc -S listing (M mfma, v VALU, r ds_read, w ds_write,
G buffer_load, W s_waitcnt, B s_barrier, S scratch, s other scalar): shows how the compiler interleaved the matrix instructions.
usage: loop_trace.py <listing.s> <mangled kernel name prefix>"""
import re
import sys
t = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
i0 = [i for i, l in enumerate(t) if l.startswith(name) and l.rstrip().split(':')[0].startswith(name) and ':' in l][0]
i1 = [i for i in range(i0, len(t)) if 's_endpgm' in t[i]][0]
body = t[i0:i1]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        n = sum('v_mfma' in x for x in body[a:i])
        if n >= 6 and (best is None or i - a < best[1] - best[0]):
            best = (a, i)
a, b = best
out = []
for l in body[a:b]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    op = l.split()[0]
    k = ('M' if 'mfma' in op else 'r' if op.startswith('ds_read') else 'w' if op.startswith('ds_write') else
         'G' if op.startswith('buffer_load') else 'S' if op.startswith('scratch') else 'B' if op == 's_barrier' else
         'v' if op.startswith('v_') else 'W' if 'waitcnt' in op else 's')
    out.append(k)
print(''.join(out))
print(len(out), 'instructions;', out.count('M'), 'mfma,', out.count('v'), 'valu,', out.count('S'), 'scratch')
