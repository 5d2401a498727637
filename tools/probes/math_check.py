import numpy as np, sys
f = open(sys.argv[1], "rb"); n = int(np.frombuffer(f.read(4), np.uint32)[0])
a = np.frombuffer(f.read(4 * n), np.uint32).view(np.float32); b = np.frombuffer(f.read(4 * n), np.uint32).view(np.float32)
s = np.frombuffer(f.read(4 * n), np.uint32); d = np.frombuffer(f.read(4 * n), np.uint32); h = np.frombuffer(f.read(2 * n), np.uint16)
with np.errstate(all="ignore"):
    rs = np.sqrt(np.abs(a)).view(np.uint32); rd = (a / b).view(np.uint32); rh = a.astype(np.float16).view(np.uint16)
def cmp(name, got, ref, isnan):
    bad = (got != ref) & ~isnan
    print(f"{name}: {int(bad.sum())} of {n} differ", end="")
    if bad.any():
        i = np.nonzero(bad)[0][:5]; print(" e.g.", [(hex(a.view(np.uint32)[k]), hex(b.view(np.uint32)[k]), hex(got[k]), hex(ref[k])) for k in i])
    else: print()
with np.errstate(all="ignore"):
    cmp("sqrt", s, rs, np.isnan(np.sqrt(np.abs(a)))); cmp("div", d, rd, np.isnan(a / b)); cmp("f32->f16", h, rh, np.isnan(a))
