import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pilotguru_amd as pg
from oracle import orb_oracle
from pilotguru_amd.synth import synth_scene
rng = np.random.RandomState(5)
bad = 0; n = 0
for it in range(80):
    w = int(rng.randint(300, 1400)); h = int(rng.randint(250, 900))
    scale = float(rng.choice([1.05, 1.08, 1.1, 1.15, 1.2]))
    nlev = int(rng.randint(9, 17)); nf = int(rng.randint(200, 4000))
    img = synth_scene(2000 + it, w, h)
    try: okp, od = orb_oracle.OrbOracle(nf, scale, nlev, 20, 7).extract(img)
    except Exception as e: okp = None
    err = ""
    try:
        kp, d = pg.ORBextractor(nf, scale, nlev, 20, 7, max_width=w, max_height=h)(img)
    except Exception as e: kp = None; err = str(e)[:100]
    if (okp is None) != (kp is None): print("ERRSTATE", it, w, h, scale, nlev, nf, okp is None, kp is None, err); bad += 1; continue
    if okp is None: continue
    n += len(okp)
    if kp.tobytes() != okp.tobytes() or not np.array_equal(d, od): print("MISMATCH", it, w, h, scale, nlev, nf, len(kp), len(okp)); bad += 1
print("keypoints", n, "bad", bad)
