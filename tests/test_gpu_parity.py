"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bit-exact is the bar: keypoint position / octave / angle / response bytes and
256-bit descriptors identical, Hamming scores integer-equal."""
import numpy as np
import pytest

from pilotguru_amd.synth import synth_ride, synth_scene

pytestmark = pytest.mark.gpu

CASES = [
    # (w, h, nfeatures, seed)
    (320, 240, 500, 1),
    (640, 480, 1000, 0),
    (641, 479, 1000, 3),       # odd sizes: unaligned pitches, w & 3 != 0 blur tail
    (1280, 720, 1500, 2),
]


def _make(nfeatures, w, h, batch=1, **kw):
    import pilotguru_amd as pg
    return pg.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=batch, **kw)


@pytest.mark.parametrize("w,h,nf,seed", CASES)
def test_stages_and_output_bit_exact(oracle, w, h, nf, seed):
    img = synth_scene(seed, w, h)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    okp, odesc = ora.extract(img)
    ext = _make(nf, w, h)
    kp, desc = ext(img)
    # constructor tables
    assert np.array_equal(ext.GetScaleFactors().view(np.uint32), ora.scale_factors.view(np.uint32))
    assert np.array_equal(ext.GetInverseScaleFactors().view(np.uint32), ora.inv_scale_factors.view(np.uint32))
    assert np.array_equal(ext.GetScaleSigmaSquares().view(np.uint32), ora.level_sigma2.view(np.uint32))
    assert np.array_equal(ext.GetInverseScaleSigmaSquares().view(np.uint32), ora.inv_level_sigma2.view(np.uint32))
    assert np.array_equal(ext.features_per_level(), ora.features_per_level)
    for l in range(8):
        # K1 pyramid
        assert ext.debug_level_size(l) == ora.level_size(l)
        assert np.array_equal(ext.debug_level_image(0, l), ora.level_image(l)), "pyramid level %d" % l
        # K2 candidates: same set (device order is arbitrary)
        x, y, r = ext.debug_level_candidates(0, l)
        oc = ora.level_candidates(l)
        got = sorted(zip(y.tolist(), x.tolist(), r.tolist()))
        exp = sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["response"].tolist()))
        assert got == exp, "FAST candidates level %d" % l
        # K3 count
        assert ext.debug_level_keypoints(0, l) == ora.level_keypoints(l), "quadtree count level %d" % l
    # K3 order + K4 angle + output assembly
    assert len(kp) == len(okp)
    assert kp.tobytes() == okp.tobytes()
    # K5/K6 descriptors
    assert np.array_equal(desc, odesc)


def test_batch_device_resident_matches_oracle(oracle):
    import torch
    w, h, nf, B = 640, 480, 1000, 4
    ride = synth_ride(5, w, h, B)
    ext = _make(nf, w, h, batch=B)
    frames = torch.from_numpy(ride).cuda()
    kps, desc, n = ext.extract_batch_device(frames)
    ext.check_async()
    torch.cuda.synchronize()
    n = n.cpu().numpy()
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    kps_h = kps.cpu().numpy()
    desc_h = desc.cpu().numpy()
    descs = []
    for f in range(B):
        okp, odesc = ora.extract(ride[f])
        assert n[f] == len(okp)
        assert kps_h[f, :n[f]].tobytes() == okp.tobytes()
        assert np.array_equal(desc_h[f, :n[f]], odesc)
        descs.append(odesc)
    # K7: consecutive-frame best-2 match, integer-equal
    pq = torch.arange(1, B, dtype=torch.int32, device="cuda")
    pt = torch.arange(0, B - 1, dtype=torch.int32, device="cuda")
    bi, b1, b2 = ext.match_batch_device(desc, torch.from_numpy(n).cuda(), pq, pt)
    torch.cuda.synchronize()
    for p in range(B - 1):
        obi, ob1, ob2 = oracle.hamming_best2(descs[p + 1], descs[p])
        m = len(descs[p + 1])
        assert np.array_equal(bi[p, :m].cpu().numpy(), obi)
        assert np.array_equal(b1[p, :m].cpu().numpy().view(np.uint16), ob1)
        assert np.array_equal(b2[p, :m].cpu().numpy().view(np.uint16), ob2)


def test_unaligned_device_input_copy_path(oracle):
    import torch
    w, h, nf = 322, 242, 300
    img = synth_scene(11, w, h)
    buf = torch.zeros(1 + w * h, dtype=torch.uint8, device="cuda")
    view = buf[1:].view(1, h, w)                      # base pointer % 4 == 1 -> copy path
    view.copy_(torch.from_numpy(img).cuda())
    ext = _make(nf, w, h)
    kps, desc, n = ext.extract_batch_device(view)
    ext.check_async()
    okp, odesc = oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(img)
    n0 = int(n[0])
    assert n0 == len(okp)
    assert kps[0, :n0].cpu().numpy().tobytes() == okp.tobytes()
    assert np.array_equal(desc[0, :n0].cpu().numpy(), odesc)


def test_hamming_matrix_and_best2(oracle):
    rng = np.random.RandomState(7)
    a = rng.randint(0, 256, (777, 32)).astype(np.uint8)
    b = rng.randint(0, 256, (1033, 32)).astype(np.uint8)
    b[5] = a[3]
    b[700] = a[3]                                    # duplicate minimum: first index must win
    ext = _make(100, 320, 240)
    assert np.array_equal(ext.hamming_matrix(a, b), oracle.hamming_matrix(a, b))
    bi, b1, b2 = ext.hamming_best2(a, b)
    obi, ob1, ob2 = oracle.hamming_best2(a, b)
    assert np.array_equal(bi, obi) and np.array_equal(b1, ob1) and np.array_equal(b2, ob2)
    assert bi[3] == 5 and b1[3] == 0 and b2[3] == 0
    # edge cases: single train descriptor, empty train set
    bi, b1, b2 = ext.hamming_best2(a[:10], b[:1])
    assert np.all(bi == 0) and np.all(b2 == 65535)
    bi, b1, b2 = ext.hamming_best2(a[:10], b[:0])
    assert np.all(bi == -1) and np.all(b1 == 65535)


@pytest.mark.parametrize("na,nb", [(300, 8191), (130, 8192), (70, 9000), (1, 17), (257, 16), (64, 15),
                                   (1024, 128), (1025, 129), (2100, 2005), (3000, 127), (960, 1)])   # 16-wave workgroups: 1 024 queries share a staged tile of 128
@pytest.mark.parametrize("mode", [-1, 0, 1, 2])
def test_best2_index_field_limits(oracle, na, nb, mode):
    """The MFMA matcher packs the train index into 13 key bits (nb < 8192); larger train sets take
    the popcount kernel.  Both sides of the switch, block-size multiples and tiny sets, with near
    duplicates so that ties and the second best are exercised (ORBmatcher.cc:438-459 semantics)."""
    rng = np.random.RandomState(na * 31 + nb)
    a = rng.randint(0, 256, (na, 32)).astype(np.uint8)
    b = rng.randint(0, 256, (nb, 32)).astype(np.uint8)
    for i in range(0, na, 7):                                    # plant exact and 1-bit-off copies
        j = (i * 97) % nb
        b[j] = a[i]
        k = (j * 13 + 5) % nb
        if k != j:
            b[k] = a[i]
            b[k, i % 32] ^= 1 << (i % 8)
    b[nb - 1] = a[0]
    ext = _make(100, 320, 240)
    ext.set_option("match_mode", mode)       # every way the train descriptors reach the matrix cores (match.hip); -1 = by launch size
    try:
        bi, b1, b2 = ext.hamming_best2(a, b)
    finally:
        ext.set_option("match_mode", -1)
    obi, ob1, ob2 = oracle.hamming_best2(a, b)
    assert np.array_equal(bi, obi) and np.array_equal(b1, ob1) and np.array_equal(b2, ob2)


def test_errors_and_edge_cases():
    import pilotguru_amd as pg
    from pilotguru_amd._lib import PGORB_E_LIMIT, PGORB_E_TOOSMALL, PgorbError
    ext = _make(500, 320, 240)
    k, d = ext(np.zeros((0, 0), np.uint8))           # empty image: silent empty result (:1045)
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = ext(np.full((240, 320), 128, np.uint8))   # flat image: no corners anywhere
    assert len(k) == 0
    with pytest.raises(PgorbError) as e:
        ext(np.zeros((100, 100), np.uint8))          # level 7 would have no 30-px cell
    assert e.value.code == PGORB_E_TOOSMALL
    with pytest.raises(PgorbError) as e:
        ext(np.zeros((480, 640), np.uint8))
    assert e.value.code == PGORB_E_LIMIT
    with pytest.raises(TypeError):
        ext(np.zeros((240, 320), np.float32))
    with pytest.raises(PgorbError) as e:                 # aspect < 0.5: DistributeOctTree's nIni would be 0 (:543)
        _make(100, 320, 240)(np.zeros((180, 100), np.uint8))
    assert e.value.code == PGORB_E_TOOSMALL


@pytest.mark.parametrize("cn,rgb", [(3, True), (3, False), (4, True)])
def test_color_ingest_matches_cvtcolor_then_extract(oracle, cn, rgb):
    """GrabImageMonocular: cvtColor(RGB/BGR(A) -> GRAY) then the extractor (Tracking.cc:247-265)."""
    import torch
    w, h, nf = 324, 244, 300
    rng = np.random.RandomState(cn * 2 + rgb)
    base = synth_scene(21, w, h).astype(np.int32)
    img = np.stack([np.clip(base + rng.randint(-20, 21, base.shape), 0, 255) for _ in range(cn)], 2).astype(np.uint8)
    rgb_img = img[..., :3] if rgb else img[..., 2::-1]
    gray = oracle.rgb_to_gray(np.ascontiguousarray(rgb_img))
    ext = _make(nf, w, h)
    kps, desc, n = ext.extract_batch_color_device(torch.from_numpy(img[None]).cuda(), rgb_order=rgb)
    ext.check_async()
    assert np.array_equal(ext.debug_level_image(0, 0), gray)
    okp, odesc = oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(gray)
    n0 = int(n[0])
    assert n0 == len(okp) and kps[0, :n0].cpu().numpy().tobytes() == okp.tobytes()
    assert np.array_equal(desc[0, :n0].cpu().numpy(), odesc)


@pytest.mark.parametrize("w,h,nf", [(1920, 1080, 2000), (3840, 2160, 4000)])
def test_full_size_configs_bit_exact(oracle, w, h, nf):
    """BASELINE.json configs[1] and configs[2] frame shapes, one frame each against the oracle."""
    img = synth_scene(2, w, h)
    ext = _make(nf, w, h)
    kp, desc = ext(img)
    okp, odesc = oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(img)
    assert len(kp) == len(okp) >= nf - 1          # acceptance: every level fills its quota
    assert kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_batch_properties_at_bench_size():
    """Size-independent properties on the bench workload (1080p, 2000 kp, batch of a ride):
    run-to-run determinism (no order dependence on atomics / dispatch), batch == single frame,
    octaves ascending, every frame fills its quota, descriptors of consecutive frames match."""
    import torch
    w, h, nf, B = 1920, 1080, 2000, 8
    ride = synth_ride(0, w, h, B)
    ext = _make(nf, w, h, batch=B)
    frames = torch.from_numpy(ride).cuda()
    k1, d1, n1 = [t.clone() for t in ext.extract_batch_device(frames)]
    ext.check_async()
    k2, d2, n2 = ext.extract_batch_device(frames)
    ext.check_async()
    torch.cuda.synchronize()
    n = n1.cpu().numpy()
    assert torch.equal(n1, n2) and n.min() >= 1999
    for f in range(B):
        assert torch.equal(k1[f, :n[f]].view(torch.int32), k2[f, :n[f]].view(torch.int32))   # class_id -1 reads as NaN
        assert torch.equal(d1[f, :n[f]], d2[f, :n[f]])
    ext1 = _make(nf, w, h)
    kp, desc = ext1(ride[3])                                         # single-frame host path == batch slot 3
    assert len(kp) == n[3] and kp.tobytes() == k1[3, :n[3]].cpu().numpy().tobytes()
    assert np.array_equal(desc, d1[3, :n[3]].cpu().numpy())
    assert np.all(np.diff(kp["octave"]) >= 0) and np.all(kp["class_id"] == -1)
    # frame 4 is frame 3 shifted by (2, 1): most level-0 descriptors find an exact twin
    pq = torch.tensor([4], dtype=torch.int32, device="cuda")
    pt = torch.tensor([3], dtype=torch.int32, device="cuda")
    bi, b1, b2 = ext.match_batch_device(d1, n1, pq, pt)
    torch.cuda.synchronize()
    best = b1[0, :n[4]].cpu().numpy().view(np.uint16)
    assert (best == 0).mean() > 0.15 and np.median(best) < 40


@pytest.mark.parametrize("scale,nlevels,ini,mn,nf", [(1.5, 4, 20, 7, 600), (2.0, 3, 15, 5, 400), (1.2, 1, 20, 7, 300),
                                                      (1.1, 6, 30, 10, 700), (1.25, 5, 20, 7, 500)])
def test_other_pyramid_parameters_bit_exact(oracle, scale, nlevels, ini, mn, nf):
    """Non-default ORBextractor parameters: other scale factors take the generic pyramid kernels
    (the 4x4 fast path only covers down-scales <= 1.25), other thresholds, a single level."""
    import pilotguru_amd as pg
    w, h = 612, 452
    img = synth_scene(31, w, h)
    ora = oracle.OrbOracle(nf, scale, nlevels, ini, mn)
    okp, odesc = ora.extract(img)
    ext = pg.ORBextractor(nf, scale, nlevels, ini, mn, max_width=w, max_height=h)
    kp, desc = ext(img)
    for l in range(nlevels):
        assert np.array_equal(ext.debug_level_image(0, l), ora.level_image(l)), "pyramid level %d" % l
    assert len(kp) == len(okp) and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_blur_tie_mode_switch(oracle):
    """blur_tie_mode = 1 (scalar half-up everywhere) against the oracle's tie_mode 1."""
    import pilotguru_amd as pg
    w, h, nf = 400, 300, 500
    img = synth_scene(17, w, h)
    okp, odesc = oracle.OrbOracle(nf, 1.2, 8, 20, 7, blur_tie_mode=1).extract(img)
    kp, desc = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, blur_tie_mode=1)(img)
    assert kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_context_reuse_across_frame_sizes(oracle):
    """One context, several frame sizes in a row (plan rebuild), then back."""
    import pilotguru_amd as pg
    ext = pg.ORBextractor(400, 1.2, 8, 20, 7, max_width=640, max_height=480)
    ora = oracle.OrbOracle(400, 1.2, 8, 20, 7)
    for (w, h, seed) in ((640, 480, 1), (322, 250, 2), (640, 480, 1), (500, 300, 3)):
        img = synth_scene(seed, w, h)
        kp, desc = ext(img)
        okp, odesc = ora.extract(img)
        assert kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


@pytest.mark.parametrize("w,h,nlevels,nf", [(700, 100, 3, 300), (100, 150, 2, 120), (1000, 64, 1, 200)])
def test_extreme_aspect_ratios_and_tall_cells(oracle, w, h, nlevels, nf):
    """Panoramic / narrow frames: many quadtree roots (nIni = round(W/H), ORBextractor.cc:543),
    single cell rows with cells up to 59 px (the kernel's generic staging path and 16-quad rows)."""
    import pilotguru_amd as pg
    img = synth_scene(40 + nlevels, w, h)
    ora = oracle.OrbOracle(nf, 1.2, nlevels, 20, 7)
    okp, odesc = ora.extract(img)
    ext = pg.ORBextractor(nf, 1.2, nlevels, 20, 7, max_width=w, max_height=h)
    kp, desc = ext(img)
    for l in range(nlevels):
        x, y, r = ext.debug_level_candidates(0, l)
        oc = ora.level_candidates(l)
        assert sorted(zip(y.tolist(), x.tolist(), r.tolist())) == sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["response"].tolist()))
        assert ext.debug_level_keypoints(0, l) == ora.level_keypoints(l)
    assert len(kp) == len(okp) > 0 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_noise_image_takes_the_chunked_fast_path(oracle):
    """Pure noise: nearly every pixel passes the quick test at minThFAST in cells that have no
    iniThFAST corner -> the candidate list overflows its LDS capacity -> chunked slow path."""
    import pilotguru_amd as pg
    rng = np.random.RandomState(0)
    w, h, nf = 400, 300, 500
    img = (128 + rng.randint(-9, 10, (h, w))).astype(np.uint8)      # +-9 noise: corners at 7 but none at 20
    okp, odesc = oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(img)
    kp, desc = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)(img)
    assert len(okp) > 100 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def _clustered_frame(w, h, patch, grid):
    """Flat frame with grid x grid textured patches: few candidates, packed tightly."""
    img = np.full((h, w), 90, np.uint8)
    tex = synth_scene(patch + grid, patch, patch)
    for gy in range(grid):
        for gx in range(grid):
            y0 = 40 + gy * (h - 80 - patch) // max(grid - 1, 1) if grid > 1 else h // 3
            x0 = 40 + gx * (w - 80 - patch) // max(grid - 1, 1) if grid > 1 else w // 2 + 17
            img[y0:y0 + patch, x0:x0 + patch] = tex
    return img


@pytest.mark.parametrize("w,h,nf,patch,grid", [(960, 540, 1500, 120, 1), (1280, 720, 2000, 140, 2), (640, 480, 1000, 200, 1)])
def test_clustered_corners_deep_quadtree(oracle, w, h, nf, patch, grid):
    """Corners only inside small textured patches of a flat frame, fewer than the quota: every
    node is split down to single keys, several of which share one depth-5 descendant of a root
    (15 x 17 px at 960 x 540) -- deeper than the kernel's count pyramid, so it leaves pyramid
    mode and continues with per-generation key passes (DistributeOctTree, ORBextractor.cc:594-739)."""
    import pilotguru_amd as pg
    img = _clustered_frame(w, h, patch, grid)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    okp, odesc = ora.extract(img)
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    kp, desc = ext(img)
    for l in range(8):
        assert ext.debug_level_keypoints(0, l) == ora.level_keypoints(l), "quadtree count level %d" % l
    assert len(okp) > 200 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)
    # pigeonhole: level-0 keypoints outnumber the depth-5 descendants the patches can touch
    lvl0 = okp[okp["octave"] == 0]
    leaf_w, leaf_h = w / (round(w / h) * 32.0), h / 32.0
    assert len(lvl0) > grid * grid * (patch / leaf_w + 1) * (patch / leaf_h + 1)


@pytest.mark.parametrize("w,h,nf", [(1600, 200, 800), (2000, 120, 600), (3000, 96, 500)])
def test_many_roots_shallow_count_pyramid(oracle, w, h, nf):
    """nIni = round(W/H) of 8 / 20 / 50+ roots: the count pyramid is only 4 / 3 / 2 levels deep
    for these (LDS budget), so the switch to key passes happens in earlier generations."""
    import pilotguru_amd as pg
    img = synth_scene(77, w, h)
    ora = oracle.OrbOracle(nf, 1.2, 2, 20, 7)
    okp, odesc = ora.extract(img)
    kp, desc = pg.ORBextractor(nf, 1.2, 2, 20, 7, max_width=w, max_height=h)(img)
    assert len(okp) > 100 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def _fuzz_cases():
    rng = np.random.RandomState(20260928)
    cases = []
    for i in range(14):
        w = int(rng.randint(97, 900)); h = int(rng.randint(97, 700))
        if not (0.45 < w / h < 6.0):
            w, h = 320 + i, 240 + 2 * i
        scale = float(rng.choice([1.2, 1.2, 1.2, 1.25, 1.33, 1.5, 1.7, 2.0]))
        nlev = int(rng.randint(1, 9))
        nf = int(rng.randint(60, 1500))
        ini = int(rng.choice([20, 20, 12, 30, 40])); mn = int(rng.choice([7, 7, 5, 10, 3]))
        cases.append((w, h, scale, nlev, nf, ini, min(mn, ini), i))
    return cases


@pytest.mark.parametrize("w,h,scale,nlev,nf,ini,mn,seed", _fuzz_cases())
def test_randomised_configurations_bit_exact(oracle, w, h, scale, nlev, nf, ini, mn, seed):
    """Differential test over random frame sizes / scale factors / level counts / quotas /
    thresholds: keypoints and descriptors of the HIP path equal the oracle's, or both reject the
    frame as too small for the cell grid."""
    import pilotguru_amd as pg
    img = synth_scene(100 + seed, w, h)
    if seed % 3 == 2:                                   # low-contrast variant: most cells need the minTh pass
        img = (96 + (img.astype(np.int32) - 128) // 6).clip(0, 255).astype(np.uint8)
    try:
        okp, odesc = oracle.OrbOracle(nf, scale, nlev, ini, mn).extract(img)
    except Exception:
        okp = None
    ext = pg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w, max_height=h)
    if okp is None:
        with pytest.raises(Exception):
            ext(img)
        return
    kp, desc = ext(img)
    assert len(kp) == len(okp) and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_match_batch_ragged_frame_sizes(oracle):
    """pgorb_match_batch_device on frames of very different sizes (empty, one descriptor, around
    the 16-descriptor MFMA block and the 256-query workgroup), arbitrary pairs incl. a frame
    against itself; near-duplicates planted so that ties and second-best values matter."""
    import torch
    counts = [0, 1, 15, 16, 17, 255, 256, 257, 700, 64]
    cap = 704
    rng = np.random.RandomState(99)
    desc = rng.randint(0, 256, (len(counts), cap, 32)).astype(np.uint8)
    for f in range(1, len(counts)):                       # share some descriptors between frames
        m = min(counts[f], counts[f - 1])
        if m:
            desc[f, :m:3] = desc[f - 1, :m:3]
            desc[f, 1:m:5, 7] ^= 0x10
    pairs = [(1, 0), (0, 1), (2, 3), (3, 2), (4, 4), (8, 5), (5, 8), (6, 7), (7, 6), (9, 8), (8, 9), (1, 1), (8, 8)]
    ext = _make(100, 320, 240)
    d = torch.from_numpy(desc).cuda()
    n = torch.tensor(counts, dtype=torch.int32, device="cuda")
    pq = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device="cuda")
    pt = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device="cuda")
    want = [oracle.hamming_best2(desc[q, :counts[q]], desc[t, :counts[t]]) for q, t in pairs]
    for mode in (-1, 0, 1, 2):                            # every staging form of the MFMA matcher (match.hip)
        ext.set_option("match_mode", mode)
        try:
            bi, b1, b2 = ext.match_batch_device(d, n, pq, pt)
            torch.cuda.synchronize()
        finally:
            ext.set_option("match_mode", -1)
        for k, (q, t) in enumerate(pairs):
            m = counts[q]
            obi, ob1, ob2 = want[k]
            assert np.array_equal(bi[k, :m].cpu().numpy(), obi), (mode, q, t)
            assert np.array_equal(b1[k, :m].cpu().numpy().view(np.uint16), ob1), (mode, q, t)
            assert np.array_equal(b2[k, :m].cpu().numpy().view(np.uint16), ob2), (mode, q, t)


def test_device_entry_points_are_graph_capture_safe(oracle):
    """After the first (allocating) call, pgorb_extract_batch_device + pgorb_match_batch_device
    only enqueue work on the given stream: a hipGraph captured from them replays to the same
    bytes (launch-bound callers can wrap the per-frame launch set in a graph)."""
    import torch
    w, h, nf = 640, 480, 800
    ride = synth_ride(9, w, h, 2)
    ext = _make(nf, w, h, batch=2)
    frames = torch.from_numpy(ride).cuda()
    cap = ext.max_keypoints(w, h)
    kps = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros((2,), dtype=torch.int32, device="cuda")
    pq = torch.tensor([1], dtype=torch.int32, device="cuda")
    pt = torch.tensor([0], dtype=torch.int32, device="cuda")
    mout = (torch.zeros((1, cap), dtype=torch.int32, device="cuda"), torch.zeros((1, cap), dtype=torch.int16, device="cuda"),
            torch.zeros((1, cap), dtype=torch.int16, device="cuda"))
    s = torch.cuda.Stream()

    def step():
        ext.extract_batch_device(frames, kps, desc, n, stream=s.cuda_stream)
        ext.match_batch_device(desc, n, pq, pt, mout, stream=s.cuda_stream)

    with torch.cuda.stream(s):
        step()                                            # warm-up: allocations, table upload
        s.synchronize()
        ref = [t.clone() for t in (kps, desc, n) + mout]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        for t in (kps, desc, n) + mout:
            t.zero_()
        g.replay()
        s.synchronize()
    for a, b in zip(ref, (kps, desc, n) + mout):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    okp, odesc = oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(ride[1])
    assert int(n[1]) == len(okp) and np.array_equal(desc[1, :len(okp)].cpu().numpy(), odesc)


@pytest.mark.parametrize("rot", [0, 90, 180, 270])
@pytest.mark.parametrize("cn,rgb", [(1, True), (3, True), (4, False)])
@pytest.mark.parametrize("w,h", [(333, 251), (336, 252)])     # rows of 336 pixels take the dword-per-channel kernel for rotation 0 / 180
def test_ingest_rotation_flip_colour_on_device(oracle, rot, cn, rgb, w, h):
    """pgorb_extract_batch_ingest_device: frames as decoded (grey / RGB / BGRA) go through the
    reader's rotation and flips and Tracking's grey conversion on the device; keypoints and
    descriptors equal the oracle run on the oracle-ingested frame, for all four flip settings."""
    import torch
    import pilotguru_amd as pg
    nf = 500
    rng = np.random.RandomState(rot + cn)
    base = synth_scene(60 + rot // 90, w, h)
    if cn == 1:
        frame = base
    else:
        frame = np.stack([base, np.roll(base, 3, axis=1), (255 - base)] + ([np.full_like(base, 255)] if cn == 4 else []), axis=2)
        frame = np.ascontiguousarray(frame)
    ow, oh = (h, w) if rot in (90, 270) else (w, h)
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=ow, max_height=oh, max_batch=2)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    dev = torch.from_numpy(np.stack([frame, frame])).cuda()
    for vf in (False, True):
        for hf in (False, True):
            up = oracle.ingest_geometry(frame, rot, vf, hf)
            if cn > 1:
                c3 = np.ascontiguousarray(up[:, :, :3] if rgb else up[:, :, 2::-1])
                up = oracle.rgb_to_gray(c3)
            okp, odesc = ora.extract(up)
            kps, desc, n = ext.extract_batch_ingest_device(dev, rgb_order=rgb, rotate_degrees=rot, vertical_flip=vf, horizontal_flip=hf)
            torch.cuda.synchronize()
            assert np.array_equal(ext.debug_level_image(1, 0), up)
            for f in range(2):
                m = int(n[f])
                assert m == len(okp) > 50
                assert kps[f, :m].cpu().numpy().tobytes() == okp.tobytes()
                assert np.array_equal(desc[f, :m].cpu().numpy(), odesc)
    with pytest.raises(Exception):
        ext.extract_batch_ingest_device(dev, rotate_degrees=45)


@pytest.mark.parametrize("w,h,nf,nlev", [(969, 578, 2461, 1), (640, 480, 3000, 2), (1280, 720, 6000, 3)])
def test_large_per_level_quota_uses_global_node_arrays(oracle, w, h, nf, nlev):
    """Quotas above ~1180 keypoints on one level do not fit the quadtree workgroup's LDS node
    list; those levels run the same kernel on a global slab (k_quadtree<true>).  Same keypoints,
    same order, same descriptors as the oracle."""
    import pilotguru_amd as pg
    img = synth_scene(300 + nlev, w, h)
    ora = oracle.OrbOracle(nf, 1.2, nlev, 20, 7)
    okp, odesc = ora.extract(img)
    ext = pg.ORBextractor(nf, 1.2, nlev, 20, 7, max_width=w, max_height=h)
    assert max(ext.features_per_level()) > 1200
    kp, desc = ext(img)
    for l in range(nlev):
        assert ext.debug_level_keypoints(0, l) == ora.level_keypoints(l), "quadtree count level %d" % l
    assert len(okp) > 1500 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_batch_with_featureless_frames(oracle):
    """Flat (and nearly flat) frames inside a batch: zero candidates on every level for some
    frames, normal work for the others; counts, keypoints and matches stay per frame."""
    import torch
    w, h, nf = 480, 360, 600
    tex = synth_scene(21, w, h)
    flat = np.full((h, w), 77, np.uint8)
    dim = (100 + (tex.astype(np.int32) - 128) // 40).astype(np.uint8)        # contrast below minThFAST everywhere
    frames = np.stack([flat, tex, dim, tex[::-1].copy(), flat])
    ext = _make(nf, w, h, batch=len(frames))
    kps, desc, n = ext.extract_batch_device(torch.from_numpy(frames).cuda())
    ext.check_async()
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    n = n.cpu().numpy()
    descs = []
    for f in range(len(frames)):
        okp, odesc = ora.extract(frames[f])
        assert n[f] == len(okp), f
        assert kps[f, :n[f]].cpu().numpy().tobytes() == okp.tobytes()
        assert np.array_equal(desc[f, :n[f]].cpu().numpy(), odesc)
        descs.append(odesc)
    assert n[0] == 0 and n[4] == 0 and n[1] > 300
    pq = torch.tensor([1, 0, 2, 3, 4], dtype=torch.int32, device="cuda")
    pt = torch.tensor([0, 1, 1, 1, 0], dtype=torch.int32, device="cuda")
    bi, b1, b2 = ext.match_batch_device(desc, torch.from_numpy(n).cuda(), pq, pt)
    torch.cuda.synchronize()
    for p, (q, t) in enumerate(zip(pq.tolist(), pt.tolist())):
        obi, ob1, ob2 = oracle.hamming_best2(descs[q], descs[t])
        m = len(descs[q])
        assert np.array_equal(bi[p, :m].cpu().numpy(), obi)
        assert np.array_equal(b1[p, :m].cpu().numpy().view(np.uint16), ob1)


@pytest.mark.parametrize("rows", [16, 64])
def test_pyramid_tile_height_variants_bit_exact(oracle, rows, monkeypatch):
    """PGORB_PYR_TILE_ROWS (read when a plan is built) selects the 256 x 16 / 256 x 64 shapes of the LDS-staged
    resize used for the tile-size sweep in DESIGN.md: every level and the final output must not change."""
    import pilotguru_amd as pg
    monkeypatch.setenv("PGORB_PYR_TILE_ROWS", str(rows))
    w, h, nf = 1000, 701, 1200
    img = synth_scene(77, w, h)
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    kp, desc = ext(img)
    orc = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    okp, odesc = orc.extract(img)
    for lvl in range(1, 8):
        assert np.array_equal(ext.debug_level_image(0, lvl), orc.level_image(lvl)), (rows, lvl)
    assert kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_popcount_matcher_switch_gives_the_same_matches(tmp_path):
    """PGORB_MATCH_POPCOUNT=1 routes every size through the v_bcnt kernels (the matcher BASELINE.json's north star
    describes); the flag is read once per process, so the comparison runs in two child processes."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import pilotguru_amd as pg;"
            "r = np.random.RandomState(5); a = r.randint(0, 256, (1500, 32)).astype(np.uint8); b = r.randint(0, 256, (1777, 32)).astype(np.uint8);"
            "b[:200] = a[:200]; ext = pg.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240);"
            "i, d1, d2 = ext.hamming_best2(a, b); np.savez(sys.argv[1], i=i, d1=d1, d2=d2)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in (None, "1"):
        env = dict(os.environ)
        env.pop("PGORB_MATCH_POPCOUNT", None)
        if flag:
            env["PGORB_MATCH_POPCOUNT"] = flag
        out = os.path.join(str(tmp_path), "m%s.npz" % (flag or "0"))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env)
        outs.append(np.load(out))
    for k in ("i", "d1", "d2"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert np.all(outs[0]["d1"][:200] == 0)


@pytest.mark.parametrize("w,h,scale,nlev,nf", [(884, 212, 1.33, 4, 1438), (565, 129, 1.2, 4, 2394), (963, 352, 2.0, 3, 1828), (662, 120, 1.33, 2, 1116)])
def test_levels_with_one_tall_cell_row(oracle, w, h, scale, nlev, nf):
    """Wide, low frames: a level whose region is 30..59 px high has ONE cell row of up to 59-px cells (65 window rows) at
    the common cell width -- K2's instantiation with immediate offsets but not the narrow one.  Round 3's straight-line
    window staging first covered 63 rows only and its two-step map clearing 51 (found by tools/experiments/fuzz_parity.py)."""
    import pilotguru_amd as pg
    img = synth_scene(1000 + w, w, h)
    ora = oracle.OrbOracle(nf, scale, nlev, 20, 7)
    okp, odesc = ora.extract(img)
    ext = pg.ORBextractor(nf, scale, nlev, 20, 7, max_width=w, max_height=h)
    kp, desc = ext(img)
    tall = 0
    for l in range(nlev):
        x, y, r = ext.debug_level_candidates(0, l)
        oc = ora.level_candidates(l)
        assert sorted(zip(y.tolist(), x.tolist(), r.tolist())) == sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["response"].tolist())), "level %d" % l
        lw, lh = ora.level_size(l)
        tall += 30 <= lh - 32 < 60 and (lh - 32) > 40
    assert tall >= 1                                    # the case is in the frame
    assert len(kp) == len(okp) > 0 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


@pytest.mark.parametrize("pitch,wpb,cpw", [(0, 1, 1), (48, 1, 1), (64, 1, 1), (80, 1, 1), (96, 1, 1), (128, 1, 1), (0, 4, 1), (64, 4, 1),
                                           (0, 1, 2), (0, 1, 5), (0, 4, 3), (80, 1, 4), (0, 1, 64), (0, 2, 1), (64, 2, 2)])
def test_fast_every_tile_shape_bit_exact(oracle, pitch, wpb, cpw):
    """BASELINE.json configs[2]: "LDS tile-size sweep ... bit-exact at every tile size".  K2's LDS window pitch
    (pgorb_set_option "fast_tile_pitch": 0 = the shipped 48-byte pitch with immediate offsets, 48 ... 128 = the same window in
    an LDS row of that many bytes, run-time pitch) x waves per workgroup (1 | 4) x consecutive cell records per wave (round 5,
    "fast_cells_per_wave": the cells of a wave run one after the other on the same LDS), on the hard scenes: textured, driving-like
    (minThFAST retry per cell), pure noise (candidate list overflow -> row-chunked passes) and panoramic frames (cells up to
    59 px).  Candidate sets and the final output against the oracle.  (Rounds 1-3 swept a block form -- one workgroup
    per block of cells -- that lost on every shape and left the tree in round 4; profiles/r04_k2_tile_sweep_2160p.txt times these.)"""
    import pilotguru_amd as pg
    from pilotguru_amd.synth import synth_scene_road
    rng = np.random.RandomState(1)
    cases = [(synth_scene(7, 641, 479), 1000, 8), (synth_scene_road(2, 640, 480), 1000, 8),
             ((128 + rng.randint(-9, 10, (300, 400))).astype(np.uint8), 500, 8),
             (synth_scene(43, 700, 100), 300, 3), (synth_scene(42, 100, 150), 120, 2)]
    for img, nf, nlevels in cases:
        h, w = img.shape
        ora = oracle.OrbOracle(nf, 1.2, nlevels, 20, 7)
        okp, odesc = ora.extract(img)
        ext = pg.ORBextractor(nf, 1.2, nlevels, 20, 7, max_width=w, max_height=h)
        ext.set_option("fused_levels", 0)                 # every level through K2 (the fused launch of round 6 has one tile shape)
        ext.set_option("fast_tile_pitch", pitch)
        ext.set_option("fast_waves_per_block", wpb)
        ext.set_option("fast_cells_per_wave", cpw)
        assert ext.get_option("fast_tile_pitch") == pitch and ext.get_option("fast_waves_per_block") == wpb
        assert ext.get_option("fast_cells_per_wave") == cpw
        kp, desc = ext(img)
        for l in range(nlevels):
            x, y, r = ext.debug_level_candidates(0, l)
            oc = ora.level_candidates(l)
            assert sorted(zip(y.tolist(), x.tolist(), r.tolist())) == \
                sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["response"].tolist())), "level %d" % l
        assert len(kp) == len(okp) > 0 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


@pytest.mark.parametrize("pitch", [64, 96])
def test_fast_tile_shapes_at_2160p(oracle, pitch):
    """configs[2]'s frame shape (3840 x 2160 / 4000) through two of the swept tile shapes."""
    img = synth_scene(2, 3840, 2160)
    okp, odesc = oracle.OrbOracle(4000, 1.2, 8, 20, 7).extract(img)
    ext = _make(4000, 3840, 2160)
    ext.set_option("fused_levels", 0)
    ext.set_option("fast_tile_pitch", pitch)
    kp, desc = ext(img)
    assert kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)


def test_options_belong_to_a_context(oracle):
    """pgorb_set_option writes the CONTEXT (round 4; "matcher" / "match_mode" were process-wide statics): two contexts with
    different settings in one process keep them, and both produce the oracle's matches.  Also the round-3 advisory: the
    slab form (match_mode 0) selected AFTER a stream exists must size its arena at submit time."""
    import pilotguru_amd as pg
    from pilotguru_amd._lib import PgorbError
    w, h, nf = 640, 480, 1000
    a, b = _make(nf, w, h, batch=4), _make(nf, w, h, batch=4)
    a.set_option("matcher", 1); a.set_option("fast_tile_pitch", 64)
    b.set_option("match_mode", 0)
    assert a.get_option("matcher") == 1 and b.get_option("matcher") == 0
    assert a.get_option("match_mode") == -1 and b.get_option("match_mode") == 0
    assert a.get_option("fast_tile_pitch") == 64 and b.get_option("fast_tile_pitch") == 0
    assert a.matcher_name(2000) == "popcount" and b.matcher_name(2000) == "mfma_fp4"
    with pytest.raises(PgorbError):
        a.set_option("match_mode", 3)
    with pytest.raises(PgorbError):
        a.set_option("fast_tile_pitch", 50)
    rng = np.random.RandomState(3)
    qa = rng.randint(0, 256, (700, 32)).astype(np.uint8)
    tb = rng.randint(0, 256, (900, 32)).astype(np.uint8)
    tb[17] = qa[5]
    exp = oracle.hamming_best2(qa, tb)
    for ext in (a, b):
        got = ext.hamming_best2(qa, tb)
        assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    # a stream created under the default mode, then the slab form switched on: 4 frames x 1000 descriptors need 0.5 MB of slab
    ride = synth_ride(9, w, h, 8)
    c = _make(nf, w, h, batch=4)
    st = pg.FrameStream(c, w, h, 4, 2)
    c.set_option("match_mode", 0)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    prev = None
    for blk in range(2):
        st.input(blk)[:4] = ride[4 * blk:4 * blk + 4]
        st.submit(blk, 4)
        rn, rk, rd, rbi, rb1, rb2 = st.wait(blk)
        for f in range(4):
            okp, odesc = ora.extract(ride[4 * blk + f])
            n = int(rn[f])
            assert n == len(okp) and np.array_equal(rd[f, :n], odesc)
            if prev is not None:
                obi, ob1, ob2 = oracle.hamming_best2(odesc, prev)
                assert np.array_equal(rbi[f, :n], obi) and np.array_equal(rb1[f, :n], ob1) and np.array_equal(rb2[f, :n], ob2)
            prev = odesc
        del rn, rk, rd, rbi, rb1, rb2
    st.close(); c.close()
    st.close()                                          # closing a stream after its extractor: a no-op (round-3 advisory)


@pytest.mark.parametrize("split", [1, 0, 2])
def test_quadtree_in_two_launches_and_in_one(oracle, split):
    """K3's candidate pass as its own launch (k_qt_leaves, round 4: many small workgroups that each own a band of leaf rows, leaf
    counts and best records handed over in HBM, tables built with the plan on the HOST; option quadtree_split 1) and inside
    k_quadtree (0; tables built on the device); 2 = the library chooses per launch (the default).  All equal the oracle -- textured
    frames, a deep tree (clustered corners), many roots (shallow pyramid), a batch run three times over with the frames in
    different slots, and the tall frame without a root."""
    import pilotguru_amd as pg
    from pilotguru_amd._lib import PGORB_E_TOOSMALL, PgorbError
    cases = [(640, 480, 1000, 8, synth_ride(21, 640, 480, 1)[0]), (1283, 721, 2000, 8, synth_ride(22, 1283, 721, 1)[0]),
             (960, 540, 1500, 8, _clustered_frame(960, 540, 120, 1)), (2000, 120, 600, 2, synth_scene(77, 2000, 120))]
    for w, h, nf, nlev, img in cases:
        ora = oracle.OrbOracle(nf, 1.2, nlev, 20, 7)
        okp, odesc = ora.extract(img)
        ext = pg.ORBextractor(nf, 1.2, nlev, 20, 7, max_width=w, max_height=h)
        ext.set_option("quadtree_split", split)
        assert ext.get_option("quadtree_split") == split
        for threads in (0, 256, 512, 1024):                          # threads per K3 workgroup: the per-launch rule, or forced
            ext.set_option("quadtree_threads", threads)
            assert ext.get_option("quadtree_threads") == threads
            kp, desc = ext(img)
            assert kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc), (w, h, threads)
        ext.set_option("quadtree_threads", 0)
        for l in range(nlev):
            assert ext.debug_level_keypoints(0, l) == ora.level_keypoints(l), "quadtree count level %d" % l
        ext.close()
    w, h, nf = 800, 600, 1200
    ride = synth_ride(24, w, h, 6)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    ext = _make(nf, w, h, batch=6)
    ext.set_option("quadtree_split", split)
    exp = [ora.extract(f) for f in ride]
    for rep in range(3):
        order = [(f + rep) % 6 for f in range(6)]                  # a different frame in every slot each time
        res = ext.extract_batch(np.ascontiguousarray(ride[order]))
        for (kp, desc), f in zip(res, order):
            assert kp.tobytes() == exp[f][0].tobytes() and np.array_equal(desc, exp[f][1]), (rep, f)
    ext.close()
    t = pg.ORBextractor(300, 1.2, 8, 20, 7, max_width=320, max_height=720)
    t.set_option("quadtree_split", split)
    with pytest.raises(PgorbError) as e:                            # aspect < 0.5 WITH corners: no root (:543), raised by K3 in either form
        t(np.ascontiguousarray(synth_ride(25, 300, 700, 1)[0]))
    assert e.value.code == PGORB_E_TOOSMALL
    k, d = t(np.full((700, 300), 90, np.uint8))                     # the same shape without corners: an empty result, like the reference
    assert len(k) == 0
    t.close()


def test_host_frame_calls_replay_a_graph(oracle):
    """pgorb_extract (the reference's call shape: one frame per synchronous call, Frame.cc:251-257) launches its kernels directly
    the first time it sees a frame size, captures them into a HIP graph the second time and replays the graph from then on:
    every call against the oracle -- across different frames of one size, a size change and back, a batch-size change, an option
    change (the captured plan is stale) and a call with a stage profile armed (direct launches again)."""
    import pilotguru_amd as pg
    nf = 600
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=3)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)

    def check(img):
        kp, desc = ext(img)
        okp, odesc = ora.extract(img)
        assert len(kp) == len(okp) > 0 and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)
    for seed in range(5):
        check(synth_scene(50 + seed, 640, 480))                  # direct, capture, replay x 3
    for seed in range(3):
        check(synth_scene(60 + seed, 500, 380))                  # a new plan: direct, capture, replay
    check(synth_scene(70, 640, 480))                             # back (a third plan epoch)
    check(synth_scene(71, 640, 480))
    frames = [synth_scene(80 + k, 640, 480) for k in range(3)]
    for rep in range(3):                                         # batch of 3 through the same path
        out = ext.extract_batch(frames)
        for k in range(3):
            okp, odesc = ora.extract(frames[k])
            assert out[k][0].tobytes() == okp.tobytes() and np.array_equal(out[k][1], odesc)
    check(synth_scene(72, 640, 480)); check(synth_scene(73, 640, 480)); check(synth_scene(74, 640, 480))
    ext.set_option("fast_tile_pitch", 64)                        # the captured graph holds the old plan by value
    check(synth_scene(75, 640, 480)); check(synth_scene(76, 640, 480)); check(synth_scene(77, 640, 480))
    ext.profile_begin(2)
    check(synth_scene(78, 640, 480)); check(synth_scene(79, 640, 480))
    ncalls, ms = ext.profile_read()
    assert ncalls == 2 and ms["fast"] > 0
    check(synth_scene(90, 640, 480))
    k, d = ext(np.full((480, 640), 128, np.uint8))               # a flat frame through the replayed graph: no keypoints
    assert len(k) == 0
    ext.close()


def test_device_sincos_equals_the_oracle_for_every_input():
    """SURVEY.md hard part 4: the argument of computeOrbDescriptor's cos / sin (ORBextractor.cc:112-113) is a
    float in [0, 2 pi] -- 1.09e9 bit patterns.  EVERY one of them through the device's pg_sincos_f and the
    oracle's orc_sincos_f, compared as 64-bit checksums of 2^20 inputs each (what either differs from glibc's
    sinf / cosf by is measured by tools/sincos_sweep.c, profiles/r02_sincos_sweep.txt)."""
    import ctypes as C
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _oracle_pool import oracle_sincos_checksums
    from pilotguru_amd import _lib
    L = _lib.lib()
    L.pgorb_debug_sincos_checksum.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    last = int(np.float32(6.2831855).view(np.uint32))
    count = 1 << 20
    nblocks = last // count + 1                       # covers [0, last] and a little beyond
    got = np.zeros(nblocks, np.uint64)
    assert L.pgorb_debug_sincos_checksum(0, count, nblocks, C.c_void_p(got.ctypes.data)) == 0
    want = np.array(oracle_sincos_checksums(0, count, nblocks), np.uint64)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, "blocks of 2^20 inputs that differ: %s" % bad[:10]
    assert nblocks * count > 1.08e9


def test_pyramid_beside_fast_pipeline_is_bit_exact(oracle):
    """pgorb_set_option("pipeline_pyramid", 1): the resize chain on a high-priority side stream, K2 level by level on
    another stream behind the launch that wrote its level.  Same keypoints, descriptors and stage taps, batch of frames,
    repeated calls (the fork / join events are reused)."""
    import torch
    w, h, nf, B = 640, 480, 1000, 6
    ride = synth_ride(9, w, h, B)
    ext = _make(nf, w, h, batch=B)
    ext.set_option("pipeline_pyramid", 1)
    frames = torch.from_numpy(ride).cuda()
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    for rep in range(3):
        kps, desc, n = ext.extract_batch_device(frames)
        ext.check_async()
        torch.cuda.synchronize()
        nh = n.cpu().numpy()
        for f in (0, B - 1, rep + 1):
            okp, odesc = ora.extract(ride[f])
            assert nh[f] == len(okp) and kps[f, :nh[f]].cpu().numpy().tobytes() == okp.tobytes()
            assert np.array_equal(desc[f, :nh[f]].cpu().numpy(), odesc)
    for l in range(8):
        x, y, r = ext.debug_level_candidates(B - 1, l)
        oc = ora.level_candidates(l)                      # (the oracle's last frame is ride[rep + 1] = ride[3]; redo for B - 1)
    ora.extract(ride[B - 1])
    for l in range(8):
        assert np.array_equal(ext.debug_level_image(B - 1, l), ora.level_image(l)) if l else True
        x, y, r = ext.debug_level_candidates(B - 1, l)
        oc = ora.level_candidates(l)
        assert sorted(zip(y.tolist(), x.tolist(), r.tolist())) == sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["response"].tolist()))


@pytest.mark.parametrize("mask", [0b10, 0b1010, 0b11111110])
def test_level_groups_beside_each_other_are_bit_exact(oracle, mask):
    """pgorb_set_option("pipeline_levels", mask): K2 group of levels by group of levels on the caller's stream, K3 and
    K4-6 of each group on side streams behind it (k_quadtree / k_describe launched for a RANGE of levels: a keypoint's
    output position needs the counts of the levels below its own only, the frame's count comes from the last launch).
    Same keypoints and descriptors, batch of frames, repeated calls, and the serial order again afterwards."""
    import torch
    w, h, nf, B = 640, 480, 1000, 5
    ride = synth_ride(19, w, h, B)
    ext = _make(nf, w, h, batch=B)
    frames = torch.from_numpy(ride).cuda()
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    want = [ora.extract(ride[f]) for f in range(B)]
    for m in (mask, mask, 0):
        ext.set_option("pipeline_levels", m)
        kps, desc, n = ext.extract_batch_device(frames)
        ext.check_async()
        torch.cuda.synchronize()
        nh = n.cpu().numpy()
        for f in range(B):
            okp, odesc = want[f]
            assert nh[f] == len(okp) and kps[f, :nh[f]].cpu().numpy().tobytes() == okp.tobytes()
            assert np.array_equal(desc[f, :nh[f]].cpu().numpy(), odesc)


def test_extract_through_the_staged_upload_twice(oracle, monkeypatch):
    """PGORB_EXTRACT_STAGE=1 (the A/B form of the host-frame upload: page-locked staging buffer in chunks) grew its
    buffer with three stray teardown lines that destroyed the stream and the graph in use.  Two frames of one size (the second
    call replays the captured graph), then a larger batch through the same context (the staging buffer grows while the stream
    and the graph exist), then the first frame again: every result equals the oracle."""
    monkeypatch.setenv("PGORB_EXTRACT_STAGE", "1")
    w, h, nf = 640, 480, 1000
    ride = synth_ride(2, w, h, 4)
    ora = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    want = [ora.extract(f) for f in ride]
    ext = _make(nf, w, h, batch=4)
    for i in (0, 1, 1):
        kp, desc = ext(ride[i])
        assert kp.tobytes() == want[i][0].tobytes() and np.array_equal(desc, want[i][1])
    res = ext.extract_batch(list(ride))
    for (kp, desc), (okp, odesc) in zip(res, want):
        assert kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc)
    kp, desc = ext(ride[0])
    assert kp.tobytes() == want[0][0].tobytes() and np.array_equal(desc, want[0][1])


def _stage_check(oracle, ext, img, nf, scale, nlev, ini, mn, label):
    ora = oracle.OrbOracle(nf, scale, nlev, ini, mn)
    okp, odesc = ora.extract(img)
    kp, desc = ext(img)
    for l in range(nlev):
        assert np.array_equal(ext.debug_level_image(0, l), ora.level_image(l)), "%s: pyramid level %d" % (label, l)
        x, y, r = ext.debug_level_candidates(0, l)
        oc = ora.level_candidates(l)
        assert sorted(zip(y.tolist(), x.tolist(), r.tolist())) == \
            sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["response"].tolist())), "%s: FAST candidates level %d" % (label, l)
    assert len(kp) == len(okp) and kp.tobytes() == okp.tobytes() and np.array_equal(desc, odesc), label


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("w,h,nf,scale,nlev", [(640, 480, 1000, 1.2, 8), (641, 479, 1000, 1.2, 8), (1920, 1080, 2000, 1.2, 8),
                                               (1280, 720, 1500, 1.2, 6), (333, 251, 500, 1.2, 5), (800, 600, 1200, 1.25, 4),
                                               (1000, 1000, 1500, 1.1, 8), (700, 100, 300, 1.2, 3), (2048, 96, 600, 1.2, 2),
                                               (1024, 768, 1000, 1.5, 4), (500, 800, 600, 1.2, 4), (3840, 2160, 4000, 1.2, 8)])
def test_fused_resize_and_detect_every_level_bit_exact(oracle, w, h, nf, scale, nlev, fused):
    """fused.hip (round 6): the launch that resizes level l -> l + 1 also detects level l (each level read once), against the
    oracle stage by stage -- every pyramid level, every level's candidate set, the final output -- on textured, driving-like
    (minThFAST retry in >= 40 % of the cells) and noise frames (list overflow -> row-chunked passes inside the fused launch), for
    frame sizes whose tiles end in partial columns / clipped or skipped last cells / edge bands of every height, and for scale
    factors the 4 x 4 resize does not take (those levels fall back to K1 + K2 inside the same call).  fused = 0: the two-launch
    path, on the same frames (ORBextractor.cc:1106-1131, 765-829)."""
    import pilotguru_amd as pg
    from pilotguru_amd.synth import synth_scene_road
    rng = np.random.RandomState(w + h)
    frames = [synth_scene(w + 3 * h, w, h), synth_scene_road(h, w, h), (128 + rng.randint(-9, 10, (h, w))).astype(np.uint8)]
    ext = pg.ORBextractor(nf, scale, nlev, 20, 7, max_width=w, max_height=h)
    ext.set_option("fused_levels", fused)
    assert ext.get_option("fused_levels") == fused
    for k, img in enumerate(frames):
        _stage_check(oracle, ext, img, nf, scale, nlev, 20, 7, "%dx%d frame %d fused %d" % (w, h, k, fused))
    # the host-frame call captures its launches into a graph on the second call of a shape and replays it from the third on: the
    # fused launches (and the status-word memset in front of them) inside a graph give the same bytes
    first = ext(frames[0])
    for _ in range(3):
        again = ext(frames[0])
        assert again[0].tobytes() == first[0].tobytes() and np.array_equal(again[1], first[1])


def test_fused_levels_on_an_aliased_batch_with_odd_pitch(oracle):
    """Level 0 aliases the caller's device buffer: a row pitch that is a multiple of 16 takes the fused launch on the caller's rows
    (band chunks never pass the pitch: the last row of the last frame is the end of the caller's allocation), any other 4-aligned
    pitch sends level 0 through K1 + K2 and the rest through the fused launches -- the same keypoints either way."""
    import torch
    import pilotguru_amd as pg
    w, h, nf, B = 636, 476, 800, 3
    ride = synth_ride(5, w, h, B)
    want = [oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(f) for f in ride]
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    for pitch in (640, 636, 644):
        buf = torch.zeros((B, h, pitch), dtype=torch.uint8, device="cuda")
        buf[:, :, :w] = torch.from_numpy(np.stack(ride)).cuda()
        kps, desc, n = ext.extract_batch_device(buf[:, :, :w])
        ext.check_async()
        torch.cuda.synchronize()
        n = n.cpu().numpy()
        for k in range(B):
            assert n[k] == len(want[k][0]), (pitch, k)
            assert kps[k, :n[k]].cpu().numpy().tobytes() == want[k][0].tobytes(), (pitch, k)
            assert np.array_equal(desc[k, :n[k]].cpu().numpy(), want[k][1]), (pitch, k)


def test_registered_caller_buffer_uploads_directly(oracle):
    """pgorb_host_register: a frame buffer the caller owns and reuses, page-locked once -- pgorb_extract then takes the direct upload
    (one DMA, no staging copy) and returns the same keypoints; unregistered again, the staged path; errors are codes."""
    import ctypes as C
    import pilotguru_amd as pg
    w, h, nf = 640, 480, 800
    ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    L = ext._L
    frames = [np.ascontiguousarray(f) for f in synth_ride(31, w, h, 3)]
    want = [oracle.OrbOracle(nf, 1.2, 8, 20, 7).extract(f) for f in frames]
    buf = np.zeros((h, w), np.uint8)                                   # "the decoder's" buffer, reused for every frame
    assert L.pgorb_host_register(C.c_void_p(buf.ctypes.data), buf.nbytes) == 0
    try:
        for k in range(3):
            buf[:] = frames[k]
            kp, desc = ext(buf)
            assert kp.tobytes() == want[k][0].tobytes() and np.array_equal(desc, want[k][1]), k
    finally:
        assert L.pgorb_host_unregister(C.c_void_p(buf.ctypes.data)) == 0
    buf[:] = frames[1]
    kp, desc = ext(buf)
    assert kp.tobytes() == want[1][0].tobytes()
    assert L.pgorb_host_register(None, 16) != 0 and L.pgorb_host_unregister(None) != 0
