"""One-off soak of the descriptor matcher (a9 / K7): random set sizes around the tile (128), block (16) and workgroup (1 024) borders,
planted duplicates and near-duplicates, against the oracle's best / second-best scan.  usage: fuzz_best2.py [cases] [seed]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pilotguru_amd as pg
from oracle import orb_oracle as oracle
oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ext = pg.ORBextractor(100, 1.2, 8, 20, 7, max_width=320, max_height=240)
bad = 0; t0 = time.time(); nd = 0
edges = [1, 15, 16, 17, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2047, 2048]
for it in range(N):
    na = int(rng.choice(edges)) if rng.randint(0, 3) == 0 else int(rng.randint(1, 3200))
    nb = int(rng.choice(edges)) if rng.randint(0, 3) == 0 else int(rng.randint(0, 3200))
    a = rng.randint(0, 256, (na, 32)).astype(np.uint8)
    b = rng.randint(0, 256, (nb, 32)).astype(np.uint8)
    if nb:
        for i in rng.randint(0, na, min(na, 40)):
            j = int(rng.randint(0, nb)); b[j] = a[i]
            k = int(rng.randint(0, nb))
            if k != j:
                b[k] = a[i]
                if rng.randint(0, 2): b[k, rng.randint(0, 32)] ^= 1 << rng.randint(0, 8)
    ext.set_option("match_mode", int(rng.randint(-1, 3)))
    bi, b1, b2 = ext.hamming_best2(a, b)
    obi, ob1, ob2 = oracle.hamming_best2(a, b)
    nd += na * nb
    if not (np.array_equal(bi, obi) and np.array_equal(b1, ob1) and np.array_equal(b2, ob2)):
        print("MISMATCH", it, na, nb, flush=True); bad += 1
print("cases", N, "distances", nd, "mismatches", bad, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
