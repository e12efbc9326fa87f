import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import pilotguru_amd as pg
from oracle import orb_oracle as oracle
from pilotguru_amd.synth import synth_ride
from test_frame_matcher import _synthetic_map_points
oracle.build()
w, h, nf = 640, 480, 1200
ride = synth_ride(4, w, h, 2, dx=7, dy=3)
ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
F1, F2 = pg.Frame(ext, ride[0]), pg.Frame(ext, ride[1])
for th, ratio in ((5.0, 0.6), (3.0, 0.8), (8.0, 0.9)):
    rng = np.random.RandomState(5)
    sel, valid, px, py, lvl, vc, pd, obs = _synthetic_map_points(F1.mvKeys, F1.mDescriptors, (7, 3), rng)
    has = (rng.uniform(size=F2.N) > 0.9).astype(np.uint8)
    sf = ext.GetScaleFactors()
    onm, oasg = oracle.search_by_projection_points(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, has, valid, px, py, lvl, vc, pd, obs, th, ratio)
    nm, asg = pg.ORBmatcher(ratio, True).SearchByProjection(F2, pg.MapPoints(valid, px, py, lvl, vc, pd, obs), th, has)
    bad = np.nonzero(asg != oasg)[0]
    print("th", th, "ratio", ratio, "nm", nm, onm, "mismatching keypoints", len(bad))
    g = oracle.frame_grid(F2.mvKeys, F2.bounds)
    for k in bad[:12]:
        qs = [int(asg[k]), int(oasg[k])]
        info = []
        for q in qs:
            if q < 0: info.append(None); continue
            r = (2.5 if vc[q] > 0.998 else 4.0) * th * sf[lvl[q]]
            cnt = len(oracle.features_in_area(F2.mvKeys, g, F2.bounds, float(px[q]), float(py[q]), float(r), int(lvl[q]) - 1, int(lvl[q])))
            info.append((q, cnt, int(obs[q])))
        print("  keypoint", int(k), "gpu query / oracle query (index, list length, has obs):", info)
