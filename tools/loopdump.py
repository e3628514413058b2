#!/usr/bin/env python3
"""loopdump.py — print the basic blocks of one loop (label .. last backward branch) of a kernel.
usage: loopdump.py <file.s> <kernel-substring> <label> [-b]   (-b: block summary only)"""
import re, sys
path, kern, lab = sys.argv[1:4]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and kern in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start + 1:end]
a = next(i for i, l in enumerate(body) if l.startswith(lab + ':'))
last = a
for i in range(a, len(body)):
    t = body[i].strip()
    if (t.startswith('s_cbranch') or t.startswith('s_branch')) and t.split(';')[0].split()[-1] == lab: last = i
blk, n = lab, 0
out = []
for l in body[a:last + 1]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    t = l.strip()
    if m:
        out.append((blk, n)); blk, n = m.group(1), 0
        if '-b' not in sys.argv: print(l)
        continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    n += 1
    if '-b' not in sys.argv: print('   ', t.split(';')[0].rstrip())
out.append((blk, n))
if '-b' in sys.argv:
    for b, n in out: print(b, n)
