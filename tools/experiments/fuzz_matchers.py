"""One-off soak of the rows either side of the extractor: random frame pairs and parameters through the Frame grid,
SearchForInitialization, both SearchByProjection forms, the BoW transform and SearchByBoW -- HIP path vs oracle, bit for bit.
usage: fuzz_matchers.py [cases] [seed]"""
import os, sys, tempfile, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pilotguru_amd as pg
from pilotguru_amd import vocab as V
from oracle import orb_oracle as oracle
from pilotguru_amd.synth import synth_ride

oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp()
bad = 0; t0 = time.time(); checks = 0
stats = {"sfi": 0, "proj": 0, "last": 0, "bow": 0, "tf": 0}


def map_points(K, D, shift, drop, dup, jit):
    n = len(K)
    sel = rng.permutation(n)[: max(1, int(n * (1 - drop)))]
    sel = np.concatenate([sel, sel[: int(n * dup)]])
    px = (K["x"][sel] - shift[0] + rng.uniform(-jit, jit, len(sel))).astype(np.float32)
    py = (K["y"][sel] - shift[1] + rng.uniform(-jit, jit, len(sel))).astype(np.float32)
    valid = (rng.uniform(size=len(sel)) > 0.05).astype(np.uint8)
    vc = np.where(rng.uniform(size=len(sel)) > 0.5, 0.9995, 0.9).astype(np.float32)
    obs = (rng.uniform(size=len(sel)) > 0.1).astype(np.uint8)
    return sel, valid, px, py, K["octave"][sel].astype(np.int32), vc, D[sel], obs


def report(what, it, cfg):
    global bad
    bad += 1
    print("MISMATCH", what, "case", it, cfg, flush=True)


for it in range(N):
    w = int(rng.randint(160, 900)); h = int(rng.randint(120, 700))
    nf = int(rng.randint(150, 3000)); nlev = int(rng.randint(1, 9))
    scale = float(rng.choice([1.2, 1.2, 1.2, 1.1, 1.5]))
    dx, dy = int(rng.randint(0, 13)), int(rng.randint(0, 9))
    ride = synth_ride(5000 + it, w, h, 2, dx=dx, dy=dy)
    if rng.randint(0, 2):                                        # the camera moves the other way
        ride = np.ascontiguousarray(ride[::-1]); dx, dy = -dx, -dy
    cfg = dict(w=w, h=h, nf=nf, nlev=nlev, scale=scale, dx=dx, dy=dy)
    if rng.randint(0, 5) == 0:                                   # low contrast: few keypoints, empty windows
        ride = (100 + (ride.astype(np.int32) - 128) // 5).clip(0, 255).astype(np.uint8)
    try:
        ext = pg.ORBextractor(nf, scale, nlev, 20, 7, max_width=w, max_height=h)
        F1, F2 = pg.Frame(ext, ride[0]), pg.Frame(ext, ride[1])
    except Exception as e:
        print("skip", it, cfg, str(e)[:80]); continue
    if F1.N < 4 or F2.N < 4:
        continue
    # Frame grid
    for F in (F1, F2):
        start, idx = oracle.frame_grid(F.mvKeys, F.bounds)
        checks += 1
        if not (np.array_equal(F.grid_start, start) and np.array_equal(F.grid_idx[:len(idx)], idx)): report("grid", it, cfg)
    # SearchForInitialization, twice (the second call starts from the updated vbPrevMatched)
    ratio = float(rng.choice([0.9, 0.9, 0.7, 0.6, 1.0])); ori = bool(rng.randint(0, 2)); win = int(rng.choice([100, 100, 40, 8, 250, 3]))
    prev = np.stack([F1.mvKeys["x"], F1.mvKeys["y"]], 1).astype(np.float32)
    oprev = prev.copy()
    m = pg.ORBmatcher(ratio, ori)
    for rep in range(2):
        onm, om12, oprev = oracle.search_for_initialization(F1.mvKeys, F1.mDescriptors, F2.mvKeys, F2.mDescriptors, F2.bounds, oprev, win, ratio, ori)
        nm, m12 = m.SearchForInitialization(F1, F2, prev, win)
        checks += 1; stats["sfi"] += onm
        if not (nm == onm and np.array_equal(m12, om12) and prev.tobytes() == oprev.tobytes()):
            report("SearchForInitialization", it, dict(cfg, ratio=ratio, ori=ori, win=win, rep=rep)); break
    # SearchByProjection (map points) and (last frame)
    sf = ext.GetScaleFactors()
    drop, dup, jit = float(rng.uniform(0, 0.6)), float(rng.uniform(0, 0.5)), float(rng.choice([0.5, 1.5, 4.0]))
    sel, valid, px, py, lvl, vc, pd, obs = map_points(F1.mvKeys, F1.mDescriptors, (dx, dy), drop, dup, jit)
    th = float(rng.choice([1.0, 3.0, 5.0, 8.0])); ratio = float(rng.choice([0.8, 0.6, 0.9]))
    has = (rng.uniform(size=F2.N) > rng.choice([0.9, 0.5, 1.1])).astype(np.uint8)
    onm, oasg = oracle.search_by_projection_points(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, has, valid, px, py, lvl, vc, pd, obs, th, ratio)
    nm, asg = pg.ORBmatcher(ratio, True).SearchByProjection(F2, pg.MapPoints(valid, px, py, lvl, vc, pd, obs), th, has)
    checks += 1; stats["proj"] += onm
    if not (nm == onm and np.array_equal(asg, oasg)): report("SearchByProjection(points)", it, dict(cfg, th=th, ratio=ratio, drop=drop, dup=dup, jit=jit))
    th = float(rng.choice([7.0, 15.0, 30.0, 3.0])); ori = bool(rng.randint(0, 2))
    ang = F1.mvKeys["angle"][sel].copy()
    ang[::7] = (ang[::7] + 100.0) % 360.0
    onm, oasg = oracle.search_by_projection_frame(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, None, valid, px, py, lvl, ang, pd, obs, th, ori)
    nm, asg = pg.ORBmatcher(0.9, ori).SearchByProjectionLastFrame(F2, valid, px, py, lvl, ang, pd, obs, th)
    checks += 1; stats["last"] += onm
    if not (nm == onm and np.array_equal(asg, oasg)): report("SearchByProjection(last frame)", it, dict(cfg, th=th, ori=ori))
    # vocabulary: transform both frames, SearchByBoW
    k = int(rng.randint(2, 11)); L = int(rng.randint(1, 5 if k > 6 else 6)); levelsup = int(rng.randint(0, L + 2))
    desc, weight, parent = V.synth_vocabulary(k, L, seed=int(rng.randint(1 << 30)))
    path = os.path.join(tmp, "voc.txt")
    V.write_vocabulary_text(path, k, L, desc, weight, parent)
    voc = V.ORBVocabulary(text_file=path)
    voc.upload(ext)
    bvK, fvK = voc.transform(F1.mDescriptors, levelsup)
    bvF, fvF = voc.transform(F2.mDescriptors, levelsup)
    ora = oracle.VocabOracle(path)
    for Fx, bv, fv in ((F1, bvK, fvK), (F2, bvF, fvF)):
        obv, ofv = ora.transform(Fx.mDescriptors, levelsup)
        checks += 1; stats["tf"] += len(obv[0])
        same = np.array_equal(bv[0], obv[0]) and bv[1].tobytes() == obv[1].tobytes() and all(np.array_equal(a, b) for a, b in zip(fv, ofv))
        if not same: report("BoW transform", it, dict(cfg, k=k, L=L, levelsup=levelsup))
    del ora
    ratio = float(rng.choice([0.7, 0.9, 0.6])); ori = bool(rng.randint(0, 2))
    validK = (rng.uniform(size=F1.N) > rng.choice([0.3, 0.0, 0.8])).astype(np.uint8)
    onm, om = oracle.search_by_bow(F1.mDescriptors, F1.mvKeys["angle"], validK, fvK, F2.mDescriptors, F2.mvKeys["angle"], fvF, ratio, ori)
    nm, mt = pg.ORBmatcher(ratio, ori).SearchByBoW(ext, F1.mDescriptors, F1.mvKeys["angle"], validK, fvK, F2, fvF)
    checks += 1; stats["bow"] += onm
    if not (nm == onm and np.array_equal(mt, om)): report("SearchByBoW", it, dict(cfg, k=k, L=L, levelsup=levelsup, ratio=ratio, ori=ori))
    del ext
print("cases", N, "checks", checks, "matches compared", stats, "mismatches", bad, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
