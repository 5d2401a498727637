"""Developer analysis: static instruction counts of one kernel in a `hipcc -S -gline-tables-only` listing, per source
line (.loc), with the source file's own region markers: tools/analysis/isa_lines.py <listing.s> <mangled kernel> [min]"""
import re, sys, collections
path, name = sys.argv[1], sys.argv[2]
minimum = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split('\n')
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m: files[int(m.group(1))] = m.group(3)
start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
cur = (0, 0)
valu = collections.Counter(); allc = collections.Counter()
for l in lines[start + 1:end]:
    ls = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', ls)
    if m:
        cur = (int(m.group(1)), int(m.group(2))); continue
    if not ls or ls.startswith(';') or ls.startswith('.') or ls.endswith(':'): continue
    op = ls.split()[0]
    allc[cur] += 1
    if op.startswith('v_') and not op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): valu[cur] += 1
print('total', sum(allc.values()), 'valu', sum(valu.values()))
byfile = collections.Counter()
for (f, ln), v in valu.items(): byfile[files.get(f, '?')] += v
print('valu by file', dict(byfile))
for f in sorted(set(k[0] for k in valu)):
    rows = sorted((ln, v, allc[(f, ln)]) for (ff, ln), v in valu.items() if ff == f)
    print('==', files.get(f, '?'))
    for ln, v, a in rows:
        if v >= minimum: print('  line %5d  valu %4d  all %4d' % (ln, v, a))
