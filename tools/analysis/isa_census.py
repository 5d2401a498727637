"""Developer analysis: static instruction census of a kernel in a hipcc -S listing (blocks, loops, barriers)."""
import re, sys
from collections import Counter
path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks = []; cur = ['entry', []]
for l in lines[start + 1:end]:
    ls = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', ls)
    if m:
        blocks.append(cur); cur = [m.group(1), []]; continue
    if not ls or ls.startswith(';') or ls.startswith('.'): continue
    cur[1].append(ls)
blocks.append(cur)
tot = sum(len(b[1]) for b in blocks)
print('blocks', len(blocks), 'instructions', tot)
c = Counter()
def kind(i):
    op = i.split()[0]
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane-move'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'vmem'
    if op in ('s_waitcnt',): return 'waitcnt'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op == 's_nop': return 'nop'
    if op.startswith('s_load'): return 'smem'
    return 'salu'
for b in blocks:
    for i in b[1]: c[kind(i)] += 1
print(dict(c))
idx = 0
pos = {}
for b in blocks:
    pos[b[0]] = idx
    idx += len(b[1])
idx = 0
prev = 0
for b in blocks:
    for i in b[1]:
        idx += 1
        if i.split()[0] == 's_barrier':
            print(f'barrier at instr {idx} (+{idx - prev})  block {b[0]}'); prev = idx
# backward branches = loops
idx = 0
for b in blocks:
    for i in b[1]:
        idx += 1
        m = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', i)
        if m and pos.get(m.group(1), 1 << 30) < idx:
            print(f'loop: instr {pos[m.group(1)]}..{idx} ({idx - pos[m.group(1)]} instrs) {b[0]} -> {m.group(1)}')
