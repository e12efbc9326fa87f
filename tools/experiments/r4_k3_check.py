"""quick K3 check: a few frame shapes through the extractor against the oracle (run under `timeout`)"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
import pilotguru_amd as pg
from oracle import orb_oracle
from pilotguru_amd.synth import synth_scene
for (w, h, nf) in [(320, 240, 500), (640, 480, 1000), (1920, 1080, 2000)]:
    img = synth_scene(3, w, h)
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    kp, d = ext(img)
    okp, od = orb_oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(img)
    print(w, h, nf, len(kp), len(okp), kp.tobytes() == okp.tobytes(), np.array_equal(d, od), flush=True)
