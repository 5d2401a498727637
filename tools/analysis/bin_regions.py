"""Developer analysis: static instructions of pm_bin_kernel<false,4> per region of pm_bin_rows.h (hipcc -S -gline-tables-only listing)."""
import re, collections, sys
path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/isa/bin4.s'
detail = sys.argv[2] if len(sys.argv) > 2 else 'votes'
lines = open(path).read().split('\n')
cur = None; cnt = collections.Counter()
for l in lines:
    ls = l.strip()
    if ls.startswith('.loc'):
        chain = re.findall(r'(\S+?):(\d+):\d+', ls.split(';', 1)[1]) if ';' in ls else []
        key = None
        for f, ln in chain:
            if 'pm_bin_rows.h' in f:
                key = int(ln); break
        cur = key
        continue
    if not ls or ls.startswith(';') or ls.startswith('.') or ls.endswith(':'): continue
    cnt[cur] += 1
print('total', sum(cnt.values()))
src = open('/root/repo/piet_metal_amd/csrc/pm_bin_rows.h').read().split('\n')
def find(s): return next(i + 1 for i, l in enumerate(src) if s in l)
marks = [('prologue', 1), ('item scan', find('for (uint32_t ib = 0;; ib += kBatch)')), ('headers', find('---- candidate headers')),
         ('sup/chunk tests', find('---- super-chunk stream')), ('votes', find('---- segment votes')), ('piece alloc', find("---- the tiles' pieces of this record")),
         ('cand pass', find('---- candidates pass.')), ('tail wave', find('// the tail wave: where the pieces went')), ('tail hdrs', find('---- the tail wave: piece headers')),
         ('entries', find('---- candidate entries, the same way')), ('scatter', find('---- scatter: every relevant')), ('end', find('cursor += n_slots;'))]
rt0, rt1 = find('auto RowTailIssue'), find('// Records hold up to kBatch CANDIDATES')
b = collections.Counter()
def region(ln):
    if ln is None: return '?'
    if rt0 <= ln < rt1: return 'rowtail'
    c = [n for n, s0 in marks if s0 <= ln]
    return c[-1] if c else '?'
for ln, c in cnt.items(): b[region(ln)] += c
for k, v in b.items(): print(f'{k:18s}{v}')
for ln in sorted(k for k in cnt if k and region(k) == detail):
    if cnt[ln] >= 5: print(ln, cnt[ln], src[ln - 1].strip()[:120])
