"""One-off soak: many random configurations, HIP path vs oracle, bit for bit."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pilotguru_amd as pg
from oracle import orb_oracle
from pilotguru_amd.synth import synth_scene, synth_scene_road
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; t0 = time.time(); nkp = 0
for it in range(N):
    w = int(rng.randint(90, 1000)); h = int(rng.randint(90, 800))
    scale = float(rng.choice([1.2, 1.2, 1.1, 1.25, 1.33, 1.5, 1.7, 2.0]))
    nlev = int(rng.randint(1, 9)); nf = int(rng.randint(40, 2500))
    ini = int(rng.choice([20, 20, 12, 30, 40, 8])); mn = min(int(rng.choice([7, 7, 5, 10, 3, 1])), ini)
    kind = rng.randint(0, 6)
    img = synth_scene(1000 + it, w, h)
    if kind == 1: img = (96 + (img.astype(np.int32) - 128) // 6).clip(0, 255).astype(np.uint8)
    elif kind == 2: img = (128 + rng.randint(-12, 13, (h, w))).astype(np.uint8)
    elif kind == 3:
        img = np.full((h, w), 90, np.uint8); p = min(h, w) // 3
        img[h // 4:h // 4 + p, w // 3:w // 3 + p] = synth_scene(it, p, p)
    elif kind == 4: img = rng.randint(0, 256, (h, w)).astype(np.uint8)
    elif kind == 5: img = synth_scene_road(2000 + it, w, h)
    try:
        okp, od = orb_oracle.OrbOracle(nf, scale, nlev, ini, mn).extract(img)
    except Exception as e:
        okp = None
    err = ""
    try:
        ext = pg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w, max_height=h)
        form = int(rng.randint(0, 3))                    # K2: cell form, or (developer builds) the block form with a random tile shape
        try:
            ext.set_option("fast_kernel", 1 if form else 0)
        except Exception:
            form = 0                                     # the product library has the cell form only
        if form:
            ext.set_option("fast_block_cx", int(rng.randint(1, 5))); ext.set_option("fast_block_cy", int(rng.randint(1, 5)))
        kp, d = ext(img)
    except Exception as e:
        kp = None; err = str(e)
    if (okp is None) != (kp is None):
        print("MISMATCH (error state)", it, w, h, scale, nlev, nf, ini, mn, kind, okp is None, kp is None, err[:80]); bad += 1; continue
    if okp is None: continue
    nkp += len(okp)
    if len(kp) != len(okp) or kp.tobytes() != okp.tobytes() or not np.array_equal(d, od):
        print("MISMATCH", it, w, h, scale, nlev, nf, ini, mn, kind, len(kp), len(okp)); bad += 1
print("cases", N, "keypoints", nkp, "mismatches", bad, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
