"""One-off soak of the ingest row (f3): random frame sizes (odd widths, widths that do / do not take the dword kernels),
channel counts, channel orders, rotations and flips through pgorb_extract_batch_ingest_device; the grey level-0 plane and
every pyramid level against the oracle, keypoints and descriptors of the whole frame for one case in four.
usage: fuzz_ingest.py [cases] [seed]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
import pilotguru_amd as pg
from oracle import orb_oracle as oracle
from pilotguru_amd.synth import synth_scene

oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; t0 = time.time(); planes = 0; full = 0
for it in range(N):
    w = int(rng.randint(97, 1300)); h = int(rng.randint(97, 900))
    if rng.randint(0, 3) == 0: w = (w + 3) & ~3
    if rng.randint(0, 6) == 0: w = (w + 15) & ~15
    cn = int(rng.choice([1, 3, 3, 4])); rgb = bool(rng.randint(0, 2)); rot = int(rng.choice([0, 0, 90, 180, 270]))
    vf, hf = bool(rng.randint(0, 2)), bool(rng.randint(0, 2))
    B = int(rng.randint(1, 4)); nlev = int(rng.randint(1, 9)); nf = int(rng.randint(100, 1500))
    cfg = dict(w=w, h=h, cn=cn, rgb=rgb, rot=rot, vf=vf, hf=hf, B=B, nlev=nlev, nf=nf)
    if rng.randint(0, 2):
        src = rng.randint(0, 256, (B, h, w) if cn == 1 else (B, h, w, cn)).astype(np.uint8)
    else:
        g = np.stack([synth_scene(9000 + 7 * it + b, w, h) for b in range(B)])
        src = g if cn == 1 else np.ascontiguousarray(np.stack([g, np.roll(g, 5, axis=2), 255 - g] + ([np.full_like(g, 200)] if cn == 4 else []), axis=3))
    ow, oh = (h, w) if rot in (90, 270) else (w, h)
    try:
        ext = pg.ORBextractor(nf, 1.2, nlev, 20, 7, max_width=ow, max_height=oh, max_batch=B)
        kps, desc, n = ext.extract_batch_ingest_device(torch.from_numpy(src).cuda(), rgb_order=rgb, rotate_degrees=rot, vertical_flip=vf, horizontal_flip=hf)
        torch.cuda.synchronize(); ext.check_async()
    except Exception as e:
        print("skip", it, cfg, str(e)[:70]); continue
    ora = oracle.OrbOracle(nf, 1.2, nlev, 20, 7)
    for b in range(B):
        up = oracle.ingest_geometry(src[b], rot, vf, hf)
        if cn > 1:
            up = oracle.rgb_to_gray(np.ascontiguousarray(up[:, :, :3] if rgb else up[:, :, 2::-1]))
        planes += 1
        if not np.array_equal(ext.debug_level_image(b, 0), up):
            print("MISMATCH level 0", it, b, cfg, flush=True); bad += 1; continue
        if b == B - 1 and it % 4 == 0:
            try:
                okp, od = ora.extract(up)
            except Exception:
                continue
            full += 1
            m = int(n[b])
            if m != len(okp) or kps[b, :m].cpu().numpy().tobytes() != okp.tobytes() or not np.array_equal(desc[b, :m].cpu().numpy(), od):
                print("MISMATCH keypoints", it, b, cfg, m, len(okp), flush=True); bad += 1
            for l in range(1, nlev):
                if not np.array_equal(ext.debug_level_image(b, l), ora.level_image(l)):
                    print("MISMATCH pyramid level", l, it, b, cfg, flush=True); bad += 1
    del ext
print("cases", N, "planes", planes, "full extractions", full, "mismatches", bad, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
