"""Robustness fuzz of the SVG front-end (CPU only): mutated documents must parse or be rejected
with an error code -- never crash or hang.  python tests/dev/fuzz_svg.py [seed] [count]"""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import piet_metal_amd as pm
base = [open(os.path.join(ROOT, 'tests', 'data', 'shapes.svg'), 'rb').read()]
tiger = open(os.path.join(ROOT, 'piet_metal_amd', 'assets', 'Ghostscript_Tiger.svg'), 'rb').read()
base.append(tiger[:6000] + b'</g></svg>')
def run(seed=1, n=3000):
    rng = random.Random(seed)
    tokens = [b'<', b'>', b'/', b'"', b"'", b'=', b'url(#', b')', b'<use href="#leaf"/>', b'<style>', b'</style>', b'{', b'}', b'%', b'e99', b'-', b'M', b'z', b'A', b'<![CDATA[', b']]>', b'<!--', b'-->', b'#', b'\x00', b'\xff', b'<g>', b'</g>', b'<svg', b'viewBox="', b'transform="rotate(', b'id="leaf"', b'href="#dot"']
    ok = err = 0
    t0 = time.time()
    for i in range(n):
        b = bytearray(rng.choice(base))
        for _ in range(rng.randint(1, 8)):
            k = rng.random()
            pos = rng.randrange(len(b))
            if k < 0.3:
                del b[pos:pos + rng.randint(1, 40)]
            elif k < 0.6:
                b[pos:pos] = rng.choice(tokens)
            elif k < 0.8:
                b[pos] = rng.randrange(256)
            else:
                q = rng.randrange(len(b)); b[pos:pos] = b[q:q + rng.randint(1, 60)]
        try:
            for flags in ((False, False), (True, True)):
                pm.PathSet.from_svg(bytes(b), spec_defaults=flags[0], flat_gradients=flags[1])
            ok += 1
        except pm.PietMetalError:
            err += 1

    return ok, err, time.time() - t0


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    ok, err, dt = run(seed, n)
    print(f"{n} mutated documents from seed {seed}: {ok} parsed, {err} rejected, {dt:.1f} s")
