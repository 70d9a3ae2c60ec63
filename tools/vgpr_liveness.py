#!/usr/bin/env python3
"""Backward liveness of VGPRs over a kernel's gfx950 assembly (from `hipcc -save-temps -g1`): reports the
program points with the highest number of simultaneously live VGPRs and the source lines they map to.
Development aid for the register diet."""
import collections
import re
import sys

path, kname = sys.argv[1], sys.argv[2]
txt = open(path).read().split('\n')
start = next(i for i, l in enumerate(txt) if l.startswith(kname + ':'))
end = next(i for i in range(start, len(txt)) if 's_endpgm' in txt[i])
L = txt[start:end + 1]


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


NODEF = ('ds_write', 'global_store', 's_', 'v_cmp', 'buffer_store', 'scratch_store', 'v_readlane', 'v_readfirstlane', 'global_load_lds', 'ds_bpermute_b32x')
ins = []      # (text, srcline, defs, uses, label_before, branch_target, is_uncond, partial)
labels = {}
cur = None
for l in L:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    m = re.match(r'\s*\.loc\s+\d+\s+(\d+)', l)
    if m:
        cur = int(m.group(1))
        continue
    if not l.startswith('\t') or l.startswith('\t;') or l.startswith('\t.'):
        continue
    parts = l.strip().split(None, 1)
    op = parts[0]
    ops = parts[1].split(',') if len(parts) > 1 else []
    nodef = op.startswith(NODEF)
    d = set() if (nodef or not ops) else regs(ops[0])
    u = set()
    for o in (ops if nodef else ops[1:]):
        u |= regs(o)
    # instructions that only partially overwrite (v_writelane, DPP with old, v_cndmask is full) keep the old value live
    partial = op.startswith(('v_writelane', 'v_mov_b32_dpp', 'v_fmac', 'v_mac', 'v_pk_fmac')) or 'dpp' in l
    if partial:
        u |= d
    tgt = None
    m = re.search(r'(\.LBB\d+_\d+)', parts[1]) if len(parts) > 1 and op.startswith(('s_cbranch', 's_branch')) else None
    if m:
        tgt = m.group(1)
    ins.append((l.strip(), cur, d, u, tgt, op == 's_branch'))
n = len(ins)
succ = [[] for _ in range(n)]
for i, it in enumerate(ins):
    if it[4] is not None and it[4] in labels:
        succ[i].append(labels[it[4]])
    if not it[5] and i + 1 < n:
        succ[i].append(i + 1)
live_in = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for i in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[i]:
            out |= live_in[s_]
        # exec-masked writes do not kill (divergent control flow): be conservative only for full-exec? assume kill
        new = (out - ins[i][2]) | ins[i][3]
        if new != live_in[i]:
            live_in[i] = new
            changed = True
sizes = [len(s_) for s_ in live_in]
print('instructions', n, 'max live VGPRs', max(sizes))
by_line = collections.defaultdict(int)
for i, sz in enumerate(sizes):
    by_line[ins[i][1]] = max(by_line[ins[i][1]], sz)
for line, sz in sorted(by_line.items(), key=lambda x: -x[1])[:25]:
    print('src line', line, 'max live', sz)
imax = max(range(n), key=lambda i: sizes[i])
live = live_in[imax]
print('max at instr', imax, ins[imax][0][:60], 'src', ins[imax][1])
defline = {}
for v in live:
    j = imax - 1
    while j >= 0 and v not in ins[j][2]:
        j -= 1
    defline[v] = (ins[j][1], ins[j][0].split()[0]) if j >= 0 else (None, 'entry')
useline = {}
for v in live:
    j = imax
    while j < n and v not in ins[j][3]:
        j += 1
    useline[v] = ins[j][1] if j < n else None
cnt = collections.Counter((defline[v][0], defline[v][1], useline[v]) for v in live)
for (dl, op, ul), c in sorted(cnt.items(), key=lambda x: -x[1])[:40]:
    print(f'  {c:3d} VGPRs: def line {dl} ({op}) -> next use line {ul}')
