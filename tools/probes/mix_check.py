import numpy as np, sys
f = open(sys.argv[1], "rb"); n = int(np.frombuffer(f.read(4), np.uint32)[0])
a = np.frombuffer(f.read(4 * n), np.uint32).view(np.float32); b = np.frombuffer(f.read(4 * n), np.uint32).view(np.float32)
o = np.frombuffer(f.read(2 * n), np.uint16); o2 = np.frombuffer(f.read(2 * n), np.uint16)
two = (a * b).astype(np.float16).view(np.uint16)                                    # round to f32, then to f16 (what the source says)
one = (a.astype(np.float64) * b.astype(np.float64)).astype(np.float16).view(np.uint16)  # exact product rounded once
print("half(a*b) as compiled: differs from two-step rounding:", int((o != two).sum()), "; from single rounding:", int((o != one).sum()), "; cases where the two differ:", int((one != two).sum()))
print("with the product pinned in a register: differs from two-step:", int((o2 != two).sum()))
