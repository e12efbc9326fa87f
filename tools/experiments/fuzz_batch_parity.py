"""One-off soak, batch paths: random configurations, a batch of 2-6 frames of MIXED scene kinds through
(a) pgorb_extract_batch, (b) the streamed ingest (pgorb_stream_*, ragged batches, depth 2-3) with its front-end stage
(SearchForInitialization of every frame against its predecessor, random window / ratio / orientation check), plus the best-2
Hamming match of consecutive frames -- HIP path vs oracle, bit for bit.  tools/experiments/fuzz_parity.py is the single-frame soak.
usage: fuzz_batch_parity.py [cases] [seed]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pilotguru_amd as pg
from oracle import orb_oracle
from pilotguru_amd.synth import synth_scene, synth_scene_road
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; t0 = time.time(); nkp = 0; nframes = 0


def scene(kind, seed, w, h):
    img = synth_scene(seed, w, h)
    if kind == 1: return (96 + (img.astype(np.int32) - 128) // 6).clip(0, 255).astype(np.uint8)
    if kind == 2: return (128 + rng.randint(-12, 13, (h, w))).astype(np.uint8)
    if kind == 3:
        out = np.full((h, w), 90, np.uint8); p = min(h, w) // 3
        out[h // 4:h // 4 + p, w // 3:w // 3 + p] = synth_scene(seed, p, p)
        return out
    if kind == 4: return rng.randint(0, 256, (h, w)).astype(np.uint8)
    if kind == 5: return synth_scene_road(seed, w, h)
    if kind == 6: return np.roll(img, (int(rng.randint(0, 5)), int(rng.randint(0, 9))), (0, 1))      # a shifted copy: matches
    return img


for it in range(N):
    w = int(rng.randint(120, 900)); h = int(rng.randint(120, 700))
    scale = float(rng.choice([1.2, 1.2, 1.2, 1.1, 1.25, 1.5]))
    nlev = int(rng.randint(1, 9)); nf = int(rng.randint(60, 2200))
    ini = int(rng.choice([20, 20, 12, 30])); mn = min(int(rng.choice([7, 7, 5, 3])), ini)
    B = int(rng.randint(2, 7))
    base = 5000 + 7 * it
    frames = [scene(int(rng.choice([0, 0, 6, 6, 1, 2, 3, 4, 5])), base if k else base, w, h) if k == 0 else
              scene(int(rng.choice([0, 6, 6, 6, 1, 2, 3, 4, 5])), base, w, h) for k in range(B)]
    try:
        ora = orb_oracle.OrbOracle(nf, scale, nlev, ini, mn)
        want = [ora.extract(f) for f in frames]
    except Exception as e:
        continue                                        # (geometry the reference cannot run: covered by fuzz_parity.py)
    ext = pg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w, max_height=h, max_batch=B)
    got = ext.extract_batch(frames)
    for k in range(B):
        if got[k][0].tobytes() != want[k][0].tobytes() or not np.array_equal(got[k][1], want[k][1]):
            print("MISMATCH batch", it, k, w, h, scale, nlev, nf, ini, mn, len(got[k][0]), len(want[k][0])); bad += 1
    # streamed ingest with a ragged tail
    sb = int(rng.randint(1, B + 1)); depth = int(rng.randint(2, 4))
    st = pg.FrameStream(ext, w, h, sb, depth)
    win = int(rng.choice([100, 100, 30, 250])); ratio = float(rng.choice([0.9, 0.7])); ori = bool(rng.randint(0, 2))
    bounds = (0.0, float(w), 0.0, float(h))
    st.frontend(bounds, win, ratio, ori, -1)
    res = {}; inflight = []; chunks = [(b0, min(sb, B - b0)) for b0 in range(0, B, sb)]
    def collect(j, s0):
        out = [np.array(a) for a in st.wait(s0)]
        fe = st.frontend_results(s0, out[1].shape[0], out[1].shape[1])
        res[j] = out + [np.array(fe[0]), np.array(fe[1])]
    for i, (b0, nb) in enumerate(chunks):
        slot = i % depth
        if len(inflight) == depth: collect(*inflight.pop(0))
        st.input(slot)[:nb] = np.stack(frames[b0:b0 + nb]); st.submit(slot, nb); inflight.append((i, slot))
    for j, s0 in inflight: collect(j, s0)
    st.close()
    for i, (b0, nb) in enumerate(chunks):
        n, kps, desc, bi, b1, b2, m12, nm = res[i]
        for k in range(nb):
            f = b0 + k
            okp, od = want[f]
            if n[k] != len(okp) or kps[k, :n[k]].tobytes() != okp.tobytes() or not np.array_equal(desc[k, :n[k]], od):
                print("MISMATCH stream", it, f, w, h, scale, nlev, nf, ini, mn); bad += 1; continue
            if f == 0: continue
            pk, pdd = want[f - 1]
            onm, om12, _ = orb_oracle.search_for_initialization(pk, pdd, okp, od, bounds, np.stack([pk["x"], pk["y"]], 1).astype(np.float32), win, ratio, ori)
            if nm[k] != onm or not np.array_equal(m12[k, :len(pk)], om12):
                print("MISMATCH init-match", it, f, w, h, nf, win, ratio, ori, int(nm[k]), onm); bad += 1
            if n[k] == 0: continue
            pd = want[f - 1][1]
            if len(pd) == 0:
                ok = np.all(bi[k, :n[k]] == -1)
            else:
                obi, ob1, ob2 = orb_oracle.hamming_best2(od, pd)
                ok = np.array_equal(bi[k, :n[k]], obi) and np.array_equal(b1[k, :n[k]], ob1) and np.array_equal(b2[k, :n[k]], ob2)
            if not ok: print("MISMATCH match", it, f, w, h, nf, len(od), len(pd)); bad += 1
    nkp += sum(len(x[0]) for x in want); nframes += B
print("cases", N, "frames", nframes, "keypoints", nkp, "mismatches", bad, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
