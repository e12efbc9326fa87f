"""Re-run fuzz_parity case `it` (seed 31) and compare stage taps with the oracle."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import pilotguru_amd as pg
from oracle import orb_oracle
from pilotguru_amd.synth import synth_scene, synth_scene_road
target = int(sys.argv[1])
rng = np.random.RandomState(31)
for it in range(target + 1):
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
    form = int(rng.randint(0, 3))
    # (product build: the block form is refused, so its two shape draws never happen)
print("case", target, w, h, scale, nlev, nf, ini, mn, kind)
ora = orb_oracle.OrbOracle(nf, scale, nlev, ini, mn)
okp, od = ora.extract(img)
ext = pg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w, max_height=h)
kp, d = ext(img)
print("keypoints", len(kp), len(okp))
for l in range(nlev):
    x, y, r = ext.debug_level_candidates(0, l)
    oc = ora.level_candidates(l)
    a = sorted(zip(y.tolist(), x.tolist(), r.tolist())); b = sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["response"].tolist()))
    lw, lh = ext.debug_level_size(l) if hasattr(ext, "debug_level_size") else (0, 0)
    print("level", l, "size", lw, lh, "candidates", len(a), len(b), "equal" if a == b else "DIFFER", "kps", ext.debug_level_keypoints(0, l), ora.level_keypoints(l))
    if a != b:
        sa, sb = set(a), set(b)
        print("   only gpu", sorted(sa - sb)[:10], "only oracle", sorted(sb - sa)[:10])
