#!/usr/bin/env python3
"""loopstat.py — instruction mix of the loops of one kernel in hipcc's -save-temps assembly.
A loop = the span between a label and the last backward branch to it (inner cold blocks included).
usage: loopstat.py <file.s> <kernel-substring> [min_instr] [-v]"""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
min_instr = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 60
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and kern in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
ins, labels = [], {}
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    ins.append(t.split(';')[0].strip())
loops = {}
for i, t in enumerate(ins):
    if t.startswith('s_cbranch') or t.startswith('s_branch'):
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] <= i: loops[tgt] = i
for tgt, last in sorted(loops.items(), key=lambda kv: labels[kv[0]]):
    body = ins[labels[tgt]:last + 1]
    if len(body) < min_instr: continue
    ops = collections.Counter(i.split()[0] for i in body)
    pre = lambda p: sum(v for k, v in ops.items() if k.startswith(p))
    print(f"{tgt}: n={len(body)} valu={pre('v_')} salu={pre('s_') - pre('s_waitcnt') - pre('s_cbranch') - pre('s_nop') - pre('s_branch')} cmp={pre('v_cmp')} "
          f"cndmask={pre('v_cndmask')} mad24={ops.get('v_mad_i32_i24', 0)} mul_lo={ops.get('v_mul_lo_u32', 0)} sad={ops.get('v_sad_u32', 0)} "
          f"ds={pre('ds_')} glb={pre('global_')} waitcnt={ops.get('s_waitcnt', 0)} branch={pre('s_cbranch') + pre('s_branch')} swappc={pre('s_swappc')}")
    if '-v' in sys.argv:
        for k, v in ops.most_common(): print(f"    {k:28s} {v}")
