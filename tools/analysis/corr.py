import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bin_work2
# reuse: monkeypatch main to return K
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin_work2.py')).read().replace('    act = K["cands"] > 0', '    return K\n    act = K["cands"] > 0')
ns = {'__file__': os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin_work2.py'), '__name__': 'bw2'}
exec(compile(src.replace('if __name__ == "__main__":', 'if False:'), 'bw2', 'exec'), ns)
sys.argv = ['x', sys.argv[1] if len(sys.argv) > 1 else 'config3']
K = ns['main']()
act = K['cands'] > 0
t = K['c4 tests'][act].astype(float); s = K['c4 slots tight'][act].astype(float); su = K['c4 super tests'][act].astype(float); ca = K['c4 chunk tests after super'][act].astype(float)
print('corr tests~slots', np.corrcoef(t, s)[0, 1], 'corr super~slots', np.corrcoef(su, s)[0, 1], 'corr after-super~slots', np.corrcoef(ca, s)[0, 1])
order = np.argsort(-s)[:40]
print('top-40 rows by slots: their rank by tests:', sorted(np.argsort(np.argsort(-t))[order])[:40])
w = su + ca + 2 * s
order = np.argsort(-w)[:40]
print('top-40 by work estimate: rank by tests', sorted(np.argsort(np.argsort(-t))[order]))
