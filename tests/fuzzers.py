"""Randomised soaks of the HIP path against the oracle (test infrastructure: this module imports oracle/).

Each fuzz_*(cases, seed) runs `cases` random configurations and returns (mismatches, summary).  A fixed-seed slice of every
fuzzer runs in the GPU suite (tests/test_gpu_fuzz.py -- the committed suite of round 3 had missed a K2 bug that only the soak
found); tools/experiments/fuzz_*.py are command-line wrappers for long soaks (profiles/r0x_fuzz_soak.txt)."""
import os
import tempfile
import time

import numpy as np


def _say(log, *a):
    if log:
        print(*a, flush=True)


def _scene(rng, kind, seed, w, h):
    from pilotguru_amd.synth import synth_scene, synth_scene_road
    img = synth_scene(seed, w, h)
    if kind == 1: return (96 + (img.astype(np.int32) - 128) // 6).clip(0, 255).astype(np.uint8)          # low contrast: the minThFAST retry
    if kind == 2: return (128 + rng.randint(-12, 13, (h, w))).astype(np.uint8)                           # faint noise
    if kind == 3:                                                                                        # a textured patch on a flat frame
        out = np.full((h, w), 90, np.uint8); p = min(h, w) // 3
        out[h // 4:h // 4 + p, w // 3:w // 3 + p] = synth_scene(seed, p, p)
        return out
    if kind == 4: return rng.randint(0, 256, (h, w)).astype(np.uint8)                                    # pure noise: list overflow
    if kind == 5: return synth_scene_road(seed, w, h)
    if kind == 6: return np.roll(img, (int(rng.randint(0, 5)), int(rng.randint(0, 9))), (0, 1))         # a shifted copy: matches
    return img


def fuzz_parity(cases=200, seed=1, log=True, size_range=((90, 1000), (90, 800)), nf_range=(40, 2500)):
    """Single frames: random size, scale factor, level count, feature count, thresholds, scene kind and K2 tile shape;
    keypoints and descriptors bit for bit, and the same error state where the reference cannot run."""
    import pilotguru_amd as pg
    from oracle import orb_oracle
    rng = np.random.RandomState(seed)
    bad = 0; t0 = time.time(); nkp = 0
    for it in range(cases):
        w = int(rng.randint(*size_range[0])); h = int(rng.randint(*size_range[1]))
        scale = float(rng.choice([1.2, 1.2, 1.1, 1.25, 1.33, 1.5, 1.7, 2.0]))
        nlev = int(rng.randint(1, 9)); nf = int(rng.randint(*nf_range))
        ini = int(rng.choice([20, 20, 12, 30, 40, 8])); mn = min(int(rng.choice([7, 7, 5, 10, 3, 1])), ini)
        kind = int(rng.randint(0, 6))
        img = _scene(rng, kind, (2000 if kind == 5 else 1000) + it if kind != 3 else it, w, h)
        try:
            okp, od = orb_oracle.OrbOracle(nf, scale, nlev, ini, mn).extract(img)
        except Exception:
            okp = None
        err = ""
        try:
            ext = pg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w, max_height=h)
            form = int(rng.randint(0, 4))                    # K2 tile shape: shipped, or a random LDS pitch / 4 cells per workgroup
            if form == 1: ext.set_option("fast_tile_pitch", int(rng.choice([48, 64, 80, 96, 112, 128])))
            if form == 2: ext.set_option("fast_waves_per_block", 4)
            if form == 3: ext.set_option("fast_cells_per_wave", int(rng.choice([2, 3, 7])))
            ext.set_option("quadtree_split", it % 3)           # K3's pass inside the quadtree kernel / as its own launch / chosen by the library
            ext.set_option("quadtree_threads", (0, 256, 512, 1024)[(it // 3) % 4])
            ext.set_option("fused_levels", 0 if form else it % 2)      # the K2 tile shapes need K2 on every level; else alternate
            kp, d = ext(img)
        except Exception as e:
            kp = None; err = str(e)
        if (okp is None) != (kp is None):
            _say(log, "MISMATCH (error state)", it, w, h, scale, nlev, nf, ini, mn, kind, okp is None, kp is None, err[:80]); bad += 1; continue
        if okp is None: continue
        nkp += len(okp)
        if len(kp) != len(okp) or kp.tobytes() != okp.tobytes() or not np.array_equal(d, od):
            _say(log, "MISMATCH", it, w, h, scale, nlev, nf, ini, mn, kind, len(kp), len(okp)); bad += 1
    return bad, "cases %d keypoints %d mismatches %d seconds %.1f" % (cases, nkp, bad, time.time() - t0)


def fuzz_levels(cases=80, seed=5, log=True):
    """9-16 pyramid levels with small scale factors (the reference runs 8 x 1.2; deep pyramids take K3's generic paths)."""
    import pilotguru_amd as pg
    from oracle import orb_oracle
    from pilotguru_amd.synth import synth_scene
    rng = np.random.RandomState(seed)
    bad = 0; n = 0; t0 = time.time()
    for it in range(cases):
        w = int(rng.randint(300, 1400)); h = int(rng.randint(250, 900))
        scale = float(rng.choice([1.05, 1.08, 1.1, 1.15, 1.2]))
        nlev = int(rng.randint(9, 17)); nf = int(rng.randint(200, 4000))
        img = synth_scene(2000 + it, w, h)
        try: okp, od = orb_oracle.OrbOracle(nf, scale, nlev, 20, 7).extract(img)
        except Exception: okp = None
        err = ""
        try: kp, d = pg.ORBextractor(nf, scale, nlev, 20, 7, max_width=w, max_height=h)(img)
        except Exception as e: kp = None; err = str(e)[:100]
        if (okp is None) != (kp is None): _say(log, "ERRSTATE", it, w, h, scale, nlev, nf, okp is None, kp is None, err); bad += 1; continue
        if okp is None: continue
        n += len(okp)
        if kp.tobytes() != okp.tobytes() or not np.array_equal(d, od): _say(log, "MISMATCH", it, w, h, scale, nlev, nf, len(kp), len(okp)); bad += 1
    return bad, "cases %d keypoints %d mismatches %d seconds %.1f" % (cases, n, bad, time.time() - t0)


def fuzz_batch_parity(cases=100, seed=1, log=True, size_range=((120, 900), (120, 700)), nf_range=(60, 2200)):
    """Batch paths: 2-6 frames of MIXED scene kinds through (a) pgorb_extract_batch, (b) the streamed ingest (ragged batches,
    depth 2-3) with its front-end stage (SearchForInitialization of every frame against its predecessor, random window / ratio /
    orientation check), plus the best-2 Hamming match of consecutive frames."""
    import pilotguru_amd as pg
    from oracle import orb_oracle
    rng = np.random.RandomState(seed)
    bad = 0; t0 = time.time(); nkp = 0; nframes = 0
    for it in range(cases):
        w = int(rng.randint(*size_range[0])); h = int(rng.randint(*size_range[1]))
        scale = float(rng.choice([1.2, 1.2, 1.2, 1.1, 1.25, 1.5]))
        nlev = int(rng.randint(1, 9)); nf = int(rng.randint(*nf_range))
        ini = int(rng.choice([20, 20, 12, 30])); mn = min(int(rng.choice([7, 7, 5, 3])), ini)
        B = int(rng.randint(2, 7))
        base = 5000 + 7 * it
        frames = [_scene(rng, int(rng.choice([0, 0, 6, 6, 1, 2, 3, 4, 5])), base, w, h) if k == 0 else
                  _scene(rng, int(rng.choice([0, 6, 6, 6, 1, 2, 3, 4, 5])), base, w, h) for k in range(B)]
        try:
            ora = orb_oracle.OrbOracle(nf, scale, nlev, ini, mn)
            want = [ora.extract(f) for f in frames]
        except Exception:
            continue                                        # (geometry the reference cannot run: covered by fuzz_parity)
        ext = pg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w, max_height=h, max_batch=B)
        ext.set_option("quadtree_split", it % 3)
        ext.set_option("quadtree_threads", (0, 256, 512, 1024)[(it // 3) % 4])
        ext.set_option("fused_levels", (it // 2) % 2)
        got = ext.extract_batch(frames)
        for k in range(B):
            if got[k][0].tobytes() != want[k][0].tobytes() or not np.array_equal(got[k][1], want[k][1]):
                _say(log, "MISMATCH batch", it, k, w, h, scale, nlev, nf, ini, mn, len(got[k][0]), len(want[k][0])); bad += 1
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
                    _say(log, "MISMATCH stream", it, f, w, h, scale, nlev, nf, ini, mn); bad += 1; continue
                if f == 0: continue
                pk, pdd = want[f - 1]
                onm, om12, _ = orb_oracle.search_for_initialization(pk, pdd, okp, od, bounds, np.stack([pk["x"], pk["y"]], 1).astype(np.float32), win, ratio, ori)
                if nm[k] != onm or not np.array_equal(m12[k, :len(pk)], om12):
                    _say(log, "MISMATCH init-match", it, f, w, h, nf, win, ratio, ori, int(nm[k]), onm); bad += 1
                if n[k] == 0: continue
                pd = want[f - 1][1]
                if len(pd) == 0:
                    ok = np.all(bi[k, :n[k]] == -1)
                else:
                    obi, ob1, ob2 = orb_oracle.hamming_best2(od, pd)
                    ok = np.array_equal(bi[k, :n[k]], obi) and np.array_equal(b1[k, :n[k]], ob1) and np.array_equal(b2[k, :n[k]], ob2)
                if not ok: _say(log, "MISMATCH match", it, f, w, h, nf, len(od), len(pd)); bad += 1
        nkp += sum(len(x[0]) for x in want); nframes += B
        ext.close()
    return bad, "cases %d frames %d keypoints %d mismatches %d seconds %.1f" % (cases, nframes, nkp, bad, time.time() - t0)


def fuzz_best2(cases=200, seed=1, log=True):
    """Descriptor matcher (a9 / K7): set sizes around the tile (128), block (16) and workgroup (1 024) borders, planted
    duplicates and near-duplicates, every match_mode and the popcount kernels, against the oracle's best / second-best scan."""
    import pilotguru_amd as pg
    from oracle import orb_oracle as oracle
    oracle.build()
    rng = np.random.RandomState(seed)
    ext = pg.ORBextractor(100, 1.2, 8, 20, 7, max_width=320, max_height=240)
    bad = 0; t0 = time.time(); nd = 0
    edges = [1, 15, 16, 17, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2047, 2048]
    for it in range(cases):
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
        ext.set_option("matcher", int(rng.randint(0, 5) == 0))
        bi, b1, b2 = ext.hamming_best2(a, b)
        obi, ob1, ob2 = oracle.hamming_best2(a, b)
        nd += na * nb
        if not (np.array_equal(bi, obi) and np.array_equal(b1, ob1) and np.array_equal(b2, ob2)):
            _say(log, "MISMATCH", it, na, nb); bad += 1
    ext.close()
    return bad, "cases %d distances %d mismatches %d seconds %.1f" % (cases, nd, bad, time.time() - t0)


def fuzz_ingest(cases=100, seed=1, log=True):
    """Ingest row (f3): random frame sizes (odd widths, widths that do / do not take the dword kernels), channel counts and
    orders, rotations and flips through pgorb_extract_batch_ingest_device; the grey level-0 plane of every frame against the
    oracle, keypoints, descriptors and every pyramid level for one case in four."""
    import torch
    import pilotguru_amd as pg
    from oracle import orb_oracle as oracle
    from pilotguru_amd.synth import synth_scene
    oracle.build()
    rng = np.random.RandomState(seed)
    bad = 0; t0 = time.time(); planes = 0; full = 0
    for it in range(cases):
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
            _say(log, "skip", it, cfg, str(e)[:70]); continue
        ora = oracle.OrbOracle(nf, 1.2, nlev, 20, 7)
        for b in range(B):
            up = oracle.ingest_geometry(src[b], rot, vf, hf)
            if cn > 1:
                up = oracle.rgb_to_gray(np.ascontiguousarray(up[:, :, :3] if rgb else up[:, :, 2::-1]))
            planes += 1
            if not np.array_equal(ext.debug_level_image(b, 0), up):
                _say(log, "MISMATCH level 0", it, b, cfg); bad += 1; continue
            if b == B - 1 and it % 4 == 0:
                try:
                    okp, od = ora.extract(up)
                except Exception:
                    continue
                full += 1
                m = int(n[b])
                if m != len(okp) or kps[b, :m].cpu().numpy().tobytes() != okp.tobytes() or not np.array_equal(desc[b, :m].cpu().numpy(), od):
                    _say(log, "MISMATCH keypoints", it, b, cfg, m, len(okp)); bad += 1
                for l in range(1, nlev):
                    if not np.array_equal(ext.debug_level_image(b, l), ora.level_image(l)):
                        _say(log, "MISMATCH pyramid level", l, it, b, cfg); bad += 1
        ext.close()
    return bad, "cases %d planes %d full extractions %d mismatches %d seconds %.1f" % (cases, planes, full, bad, time.time() - t0)


def fuzz_matchers(cases=100, seed=1, log=True, nf_range=(150, 3000), size_range=((160, 900), (120, 700))):
    """Rows either side of the extractor: random frame pairs and parameters through the Frame grid, SearchForInitialization,
    the SearchByProjection forms (map points, last frame, key frame / relocalisation), the BoW transform and SearchByBoW.
    nf_range / size_range: e.g. (4000, 4001) and ((1920, 1921), (1080, 1081)) for the initialisation workload's dense windows."""
    import pilotguru_amd as pg
    from pilotguru_amd import vocab as V
    from oracle import orb_oracle as oracle
    from pilotguru_amd.synth import synth_ride
    oracle.build()
    rng = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    state = {"bad": 0}
    t0 = time.time(); checks = 0
    stats = {"sfi": 0, "proj": 0, "last": 0, "reloc": 0, "bow": 0, "tf": 0}

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
        state["bad"] += 1
        _say(log, "MISMATCH", what, "case", it, cfg)

    for it in range(cases):
        w = int(rng.randint(*size_range[0])); h = int(rng.randint(*size_range[1]))
        nf = int(rng.randint(*nf_range)); nlev = int(rng.randint(1, 9))
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
            _say(log, "skip", it, cfg, str(e)[:80]); continue
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
        th = float(rng.choice([1.0, 3.0, 5.0, 8.0, 16.0, 45.0])); ratio = float(rng.choice([0.8, 0.6, 0.9]))     # (16 / 45: dense windows, lists beyond 64 candidates)
        has = (rng.uniform(size=F2.N) > rng.choice([0.9, 0.5, 1.1])).astype(np.uint8)
        onm, oasg = oracle.search_by_projection_points(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, has, valid, px, py, lvl, vc, pd, obs, th, ratio)
        nm, asg = pg.ORBmatcher(ratio, True).SearchByProjection(F2, pg.MapPoints(valid, px, py, lvl, vc, pd, obs), th, has)
        checks += 1; stats["proj"] += onm
        if not (nm == onm and np.array_equal(asg, oasg)): report("SearchByProjection(points)", it, dict(cfg, th=th, ratio=ratio, drop=drop, dup=dup, jit=jit))
        th = float(rng.choice([7.0, 15.0, 30.0, 3.0, 100.0])); ori = bool(rng.randint(0, 2))
        ang = F1.mvKeys["angle"][sel].copy()
        ang[::7] = (ang[::7] + 100.0) % 360.0
        onm, oasg = oracle.search_by_projection_frame(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, None, valid, px, py, lvl, ang, pd, obs, th, ori)
        nm, asg = pg.ORBmatcher(0.9, ori).SearchByProjectionLastFrame(F2, valid, px, py, lvl, ang, pd, obs, th)
        checks += 1; stats["last"] += onm
        if not (nm == onm and np.array_equal(asg, oasg)): report("SearchByProjection(last frame)", it, dict(cfg, th=th, ori=ori))
        # SearchByProjection (key frame, relocalisation): PredictScale level windows, ORBdist, sAlreadyFound
        if hasattr(oracle, "search_by_projection_keyframe"):
            th, orbdist = [(10.0, 100), (3.0, 64), (10.0, 256), (5.0, 40), (80.0, 100)][int(rng.randint(0, 5))]
            ori = bool(rng.randint(0, 2))
            nq = len(sel)
            dist3d = (rng.uniform(0.5, 30.0, nq)).astype(np.float32)
            # max distance such that the predicted level is near the keypoint's own level (sometimes far off / out of range)
            lf = oracle.log_f(sf[1])
            if np.float32(lf).view(np.uint32) != np.float32(ext.log_scale_factor()).view(np.uint32): report("log_scale_factor", it, cfg)
            maxd = (dist3d * np.power(np.float32(scale), lvl.astype(np.float32) + rng.uniform(-1.5, 1.5, nq).astype(np.float32))).astype(np.float32)
            mind = (maxd / np.float32(scale) ** np.float32(nlev - 1) * rng.choice([1.0, 1.0, 1.3], nq)).astype(np.float32)
            found = (rng.uniform(size=nq) > 0.85).astype(np.uint8)
            has0 = (rng.uniform(size=F2.N) > 0.9).astype(np.uint8)
            onm, oasg = oracle.search_by_projection_keyframe(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, has0, valid, found, px, py, dist3d, mind, maxd, lf,
                                                             F1.mvKeys["angle"][sel], pd, th, orbdist, ori)
            nm, asg = pg.ORBmatcher(0.9, ori).SearchByProjectionKeyFrame(F2, valid, found, px, py, dist3d, mind, maxd, F1.mvKeys["angle"][sel], pd, th, orbdist, has0)
            checks += 1; stats["reloc"] += onm
            if not (nm == onm and np.array_equal(asg, oasg)): report("SearchByProjection(key frame)", it, dict(cfg, th=th, orbdist=orbdist, ori=ori))
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
        ext.close()
    return state["bad"], "cases %d checks %d matches compared %s mismatches %d seconds %.1f" % (cases, checks, stats, state["bad"], time.time() - t0)


FUZZERS = {"parity": fuzz_parity, "levels": fuzz_levels, "batch_parity": fuzz_batch_parity, "best2": fuzz_best2,
           "ingest": fuzz_ingest, "matchers": fuzz_matchers}


def main(name, argv):
    cases = int(argv[1]) if len(argv) > 1 else None
    seed = int(argv[2]) if len(argv) > 2 else None
    kw = {}
    if cases is not None: kw["cases"] = cases
    if seed is not None: kw["seed"] = seed
    bad, summary = FUZZERS[name](**kw)
    print(summary, flush=True)
    return 1 if bad else 0
