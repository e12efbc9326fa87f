"""Device time of the kernels either side of the hot path on the benchmark's batch:
128 frames of 1920x1080 / 2000 keypoints resident in HBM.  HIP events on the stream the calls are issued on, median of 9;
next to each, the oracle's CPU time for the same work (1 thread, a few frames, scaled to the batch).

  ingest       RGB -> grey (Tracking.cc:247-260) + horizontal flip (image_sequence_reader.cc:53-58) in front of K1
  undistort    Frame::UndistortKeyPoints (Frame.cc:408-438), k1 != 0
  grid         Frame::AssignFeaturesToGrid (Frame.cc:234-249), 64 x 48 cells, CSR
  init-match   ORBmatcher::SearchForInitialization (ORBmatcher.cc:407-522), window 100, every frame vs its predecessor
  bow          ORBVocabulary::transform (TemplatedVocabulary.h:1126-1259) on an ORBvoc-sized tree (k = 10, L = 6)

usage: python tools/next_tier_bench.py [--batch 128] [--out profiles/r02_next_tier.txt]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pilotguru_amd as pg
from pilotguru_amd import vocab as V
from pilotguru_amd.synth import synth_ride
from oracle import orb_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--out", default="")
ap.add_argument("--features", type=int, default=2000, help="4000 = the initialisation extractor (2 * nFeatures, Tracking.cc:143)")
a = ap.parse_args()
w, h, nf, B = 1920, 1080, a.features, a.batch
ride = synth_ride(0, w, h, B)
ext = pg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
fr = torch.from_numpy(ride).cuda()
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))


rows = []
t_plain = timed(lambda: ext.extract_batch_device(fr))
rgb = fr.flip(2).unsqueeze(-1).expand(B, h, w, 3).contiguous()          # grey value in all channels, mirrored: ingest undoes it
t_ing = timed(lambda: ext.extract_batch_ingest_device(rgb, True, 0, False, True))
k2, d2, n2 = ext.extract_batch_ingest_device(rgb, True, 0, False, True)
kps, desc, n = ext.extract_batch_device(fr)
ext.check_async(); torch.cuda.synchronize()
assert torch.equal(n, n2) and all(torch.equal(d2[f, :n[f]], desc[f, :n[f]]) and torch.equal(k2[f, :n[f]].view(torch.int32), kps[f, :n[f]].view(torch.int32)) for f in range(B))   # (4899 + 9617 + 1868) v + 8192 >> 14 == v
kps, desc, n = kps.clone(), desc.clone(), n.clone()
cap = kps.shape[1]
nh = n.cpu().numpy()
r0 = rgb[0].cpu().numpy()
orb_oracle.ingest_geometry(orb_oracle.rgb_to_gray(r0), 0, False, True)          # (first call loads the library)
t0 = time.time(); g = orb_oracle.ingest_geometry(orb_oracle.rgb_to_gray(r0), 0, False, True); c_ing = (time.time() - t0) * B * 1e3
assert np.array_equal(g, ride[0])
rows.append(("ingest (RGB->grey + hflip), beyond plain extraction", t_ing - t_plain, 3 * w * h * B, c_ing))

cam = (C.c_float * 4)(1400.0, 1400.0, 960.0, 540.0); dist = (C.c_float * 5)(-0.28, 0.07, 0.0002, 0.00002, 0.0)
und = torch.empty_like(kps)
t_und = timed(lambda: ext._check(ext._L.pgorb_undistort_keypoints_batch_device(ext._h, p(kps), p(n), B, cap, cam, dist, p(und), s)))
kh = kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
k0 = kh[0, :nh[0]].copy().view(orb_oracle.KEYPOINT_DTYPE).reshape(-1)
t0 = time.time(); orb_oracle.undistort_keypoints(k0, list(cam), list(dist)); c_und = (time.time() - t0) * B * 1e3
rows.append(("undistort keypoints", t_und, int(nh.sum()) * 56, c_und))

gs = torch.empty((B, 3073), dtype=torch.int32, device="cuda"); gi = torch.empty((B, cap), dtype=torch.int32, device="cuda")
t_grid = timed(lambda: ext._check(ext._L.pgorb_frame_grid_batch_device(ext._h, p(kps), p(n), B, cap, 0.0, float(w), 0.0, float(h), p(gs), p(gi), s)))
t0 = time.time(); orb_oracle.frame_grid(k0, (0.0, float(w), 0.0, float(h))); c_grid = (time.time() - t0) * B * 1e3
rows.append(("frame grid 64x48", t_grid, int(nh.sum()) * 28 + B * 3073 * 4, c_grid))

f1 = torch.arange(0, B - 1, dtype=torch.int32, device="cuda"); f2 = torch.arange(1, B, dtype=torch.int32, device="cuda")
prev0 = kps[:B - 1, :, :2].contiguous()
prev = prev0.clone(); m12 = torch.empty((B - 1, cap), dtype=torch.int32, device="cuda"); nm = torch.empty(B - 1, dtype=torch.int32, device="cuda")
def init_match():
    prev.copy_(prev0)
    ext._check(ext._L.pgorb_search_for_initialization_batch_device(ext._h, p(kps), p(desc), p(n), cap, p(gs), p(gi), p(f1), p(f2), B - 1,
               0.0, float(w), 0.0, float(h), p(prev), p(m12), p(nm), 100, C.c_float(0.9), 1, s))
t_copy = timed(lambda: prev.copy_(prev0))
t_sfi = timed(init_match) - t_copy
dh = desc.cpu().numpy()
k1 = kh[1, :nh[1]].copy().view(orb_oracle.KEYPOINT_DTYPE).reshape(-1)
t0 = time.time()
onm, _, _ = orb_oracle.search_for_initialization(k0, dh[0, :nh[0]], k1, dh[1, :nh[1]], (0.0, float(w), 0.0, float(h)), np.stack([k0["x"], k0["y"]], 1))
c_sfi = (time.time() - t0) * (B - 1) * 1e3
assert int(nm[0]) == onm
rows.append(("SearchForInitialization (%d pairs, %d matches/pair)" % (B - 1, int(nm.float().mean())), t_sfi, int(nh.sum()) * (28 + 32) * 2, c_sfi))

dsc, wgt, par = V.synth_vocabulary_fast(10, 6, seed=7)
voc = V.ORBVocabulary(blob=V.pack_vocabulary(10, 6, dsc, wgt, par))
voc.upload(ext)
flat = torch.cat([desc[f, :nh[f]] for f in range(B)]).contiguous()
nd = flat.shape[0]
word = torch.empty(nd, dtype=torch.int32, device="cuda"); wt = torch.empty(nd, dtype=torch.float64, device="cuda"); node = torch.empty(nd, dtype=torch.int32, device="cuda")
t_bow = timed(lambda: ext._check(ext._L.pgorb_bow_transform_device(ext._h, p(flat), nd, 4, p(word), p(wt), p(node), s)))
rows.append(("BoW transform, k=10 L=6 (%d descriptors)" % nd, t_bow, nd * (32 + 6 * 10 * 32), float("nan")))

# ---- the SLAM-state matchers (a11): single host calls (H2D + kernel + D2H, wall clock), median of 9, one frame pair ----------
def wall(fn, reps=9):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


F1, F2 = pg.Frame(ext, ride[0]), pg.Frame(ext, ride[1])
rng = np.random.RandomState(5)
nk = F1.N
sel = np.concatenate([rng.permutation(nk)[: int(nk * 0.8)]] * 1)
px = (F1.mvKeys["x"][sel] - 2 + rng.uniform(-1.5, 1.5, len(sel))).astype(np.float32)       # the ride moves (2, 1) px per frame
py = (F1.mvKeys["y"][sel] - 1 + rng.uniform(-1.5, 1.5, len(sel))).astype(np.float32)
valid = (rng.uniform(size=len(sel)) > 0.05).astype(np.uint8); obs = (rng.uniform(size=len(sel)) > 0.1).astype(np.uint8)
lvl = F1.mvKeys["octave"][sel].astype(np.int32); vc = np.full(len(sel), 0.9995, np.float32); pdsc = F1.mDescriptors[sel]
mp = pg.MapPoints(valid, px, py, lvl, vc, pdsc, obs)
sf = ext.GetScaleFactors()
host_rows = []
m = pg.ORBmatcher(0.8, True)
g_ms = wall(lambda: m.SearchByProjection(F2, mp, 3.0, None))
t0 = time.perf_counter(); onm, _ = orb_oracle.search_by_projection_points(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, None, valid, px, py, lvl, vc, pdsc, obs, 3.0, 0.8); c_ms = (time.perf_counter() - t0) * 1e3
host_rows.append(("SearchByProjection(Frame, MapPoints, th 3), %d points -> %d" % (len(sel), onm), g_ms, c_ms))
m = pg.ORBmatcher(0.9, True)
ang = F1.mvKeys["angle"][sel].copy()
g_ms = wall(lambda: m.SearchByProjectionLastFrame(F2, valid, px, py, lvl, ang, pdsc, obs, 15.0))
t0 = time.perf_counter(); onm, _ = orb_oracle.search_by_projection_frame(F2.mvKeys, F2.mDescriptors, F2.bounds, sf, None, valid, px, py, lvl, ang, pdsc, obs, 15.0, True); c_ms = (time.perf_counter() - t0) * 1e3
host_rows.append(("SearchByProjection(Frame, LastFrame, th 15), %d points -> %d" % (len(sel), onm), g_ms, c_ms))
_, fvK = voc.transform(F1.mDescriptors, 4); _, fvF = voc.transform(F2.mDescriptors, 4)
kvalid = (rng.uniform(size=F1.N) > 0.3).astype(np.uint8)
m = pg.ORBmatcher(0.7, True)
g_ms = wall(lambda: m.SearchByBoW(ext, F1.mDescriptors, F1.mvKeys["angle"], kvalid, fvK, F2, fvF))
t0 = time.perf_counter(); onm, _ = orb_oracle.search_by_bow(F1.mDescriptors, F1.mvKeys["angle"], kvalid, fvK, F2.mDescriptors, F2.mvKeys["angle"], fvF, 0.7, True); c_ms = (time.perf_counter() - t0) * 1e3
host_rows.append(("SearchByBoW(KeyFrame, Frame), %d nodes -> %d" % (len(fvK[0]), onm), g_ms, c_ms))
sm = pg.ORBmatcher(0.9, True)
prevm = np.stack([F1.mvKeys["x"], F1.mvKeys["y"]], 1).astype(np.float32)
g_ms = wall(lambda: sm.SearchForInitialization(F1, F2, prevm.copy(), 100))
host_rows.append(("SearchForInitialization(F1, F2, 100), one pair through the host API", g_ms, c_sfi / (B - 1)))

# ---- round 3: the same three matchers as BATCHED, RESIDENT calls: every frame against its predecessor, all pairs in one launch ----
npairs = B - 1
qn = min(int(nk * 0.8), 2000)
qcap = qn
def rep(a, dt):                                      # the same query set for every pair (frame f's keypoints projected into f + 1)
    return torch.from_numpy(np.ascontiguousarray(np.broadcast_to(np.asarray(a[:qn], dt), (npairs,) + np.asarray(a[:qn]).shape))).cuda()
qsel = []
vq, xq, yq, lq, cq, aq, dq, oq = [], [], [], [], [], [], [], []
for f in range(npairs):
    kf = kh[f, :nh[f]].copy().view(orb_oracle.KEYPOINT_DTYPE).reshape(-1)
    ss = rng.permutation(nh[f])[:qn]
    m = len(ss)
    pad = lambda a, dt: np.concatenate([np.asarray(a, dt), np.zeros((qcap - m,) + np.asarray(a).shape[1:], dt)])
    vq.append(pad((rng.uniform(size=m) > 0.05), np.uint8)); xq.append(pad(kf["x"][ss] - 2 + rng.uniform(-1.5, 1.5, m), np.float32))
    yq.append(pad(kf["y"][ss] - 1 + rng.uniform(-1.5, 1.5, m), np.float32)); lq.append(pad(kf["octave"][ss], np.int32))
    cq.append(pad(np.full(m, 0.9995), np.float32)); aq.append(pad(kf["angle"][ss], np.float32)); dq.append(pad(dh[f, ss], np.uint8))
    oq.append(pad((rng.uniform(size=m) > 0.1), np.uint8)); qsel.append(m)
T = lambda L_: torch.from_numpy(np.stack(L_)).cuda()
vq, xq, yq, lq, cq, aq, dq, oq = T(vq), T(xq), T(yq), T(lq), T(cq), T(aq), T(dq), T(oq)
nqd = torch.tensor(qsel, dtype=torch.int32, device="cuda")
pairF = torch.arange(1, B, dtype=torch.int32, device="cuda"); pairK = torch.arange(0, B - 1, dtype=torch.int32, device="cuda")
asg = torch.empty((npairs, cap), dtype=torch.int32, device="cuda"); nmb = torch.empty(npairs, dtype=torch.int32, device="cuda")
bnd = (0.0, float(w), 0.0, float(h))
batch_rows = []
t_pp = timed(lambda: ext._check(ext._L.pgorb_search_by_projection_points_batch_device(ext._h, p(kps), p(desc), p(n), cap, p(gs), p(gi), p(pairF), npairs, *bnd,
             None, qcap, p(nqd), p(vq), p(xq), p(yq), p(lq), p(cq), p(dq), p(oq), 3.0, 0.8, p(asg), p(nmb), s)))
batch_rows.append(("SearchByProjection(Frame, MapPoints, th 3): %d pairs x %d points -> %d" % (npairs, qn, int(nmb.float().mean())), t_pp, host_rows[0][2]))
t_pf = timed(lambda: ext._check(ext._L.pgorb_search_by_projection_frame_batch_device(ext._h, p(kps), p(desc), p(n), cap, p(gs), p(gi), p(pairF), npairs, *bnd,
             None, qcap, p(nqd), p(vq), p(xq), p(yq), p(lq), p(aq), p(dq), p(oq), 15.0, 1, p(asg), p(nmb), s)))
batch_rows.append(("SearchByProjection(Frame, LastFrame, th 15): %d pairs x %d points -> %d" % (npairs, qn, int(nmb.float().mean())), t_pf, host_rows[1][2]))
wordb = torch.empty((B, cap), dtype=torch.int32, device="cuda"); wtb = torch.empty((B, cap), dtype=torch.float64, device="cuda"); nodeb = torch.empty((B, cap), dtype=torch.int32, device="cuda")
ext._check(ext._L.pgorb_bow_transform_device(ext._h, p(desc), B * cap, 4, p(wordb), p(wtb), p(nodeb), s))
fvn = torch.empty((B, cap), dtype=torch.int32, device="cuda"); fvs = torch.empty((B, cap + 1), dtype=torch.int32, device="cuda")
fvf = torch.empty((B, cap), dtype=torch.int32, device="cuda"); nfvd = torch.empty(B, dtype=torch.int32, device="cuda")
t_fv = timed(lambda: ext._check(ext._L.pgorb_feature_vectors_batch_device(ext._h, p(nodeb), p(n), B, cap, p(fvn), p(fvs), p(fvf), p(nfvd), s)))
batch_rows.append(("FeatureVector of %d frames (CSR by node, on the device)" % B, t_fv, float("nan")))
kfvd = torch.from_numpy((rng.uniform(size=(npairs, cap)) > 0.3).astype(np.uint8)).cuda()
t_bw = timed(lambda: ext._check(ext._L.pgorb_search_by_bow_batch_device(ext._h, p(kps), p(desc), p(n), cap, p(fvn), p(fvs), p(fvf), p(nfvd), p(pairK), p(pairF), npairs,
             p(kfvd), 0.7, 1, p(asg), p(nmb), s)))
batch_rows.append(("SearchByBoW(KeyFrame, Frame): %d pairs -> %d" % (npairs, int(nmb.float().mean())), t_bw, host_rows[2][2]))

lines = ["# python tools/next_tier_bench.py --batch %d --features %d   (MI355X; ms per %d-frame 1080p batch; CPU = oracle, 1 thread, one frame or pair scaled to the batch)" % (B, nf, B),
         "# extraction alone (K1-K6): %.3f ms" % t_plain,
         "%-58s %10s %12s %12s %10s" % ("kernel", "GPU ms", "us / frame", "GB/s (alg.)", "CPU ms")]
for name, ms, byts, cpu in rows:
    lines.append("%-58s %10.3f %12.2f %12.1f %10.0f" % (name, ms, ms * 1e3 / B, byts / ms / 1e6, cpu))
lines.append("# single host calls on one 1080p frame pair (upload + kernel + download, wall clock ms) next to the oracle's")
lines.append("%-72s %10s %10s" % ("call", "GPU ms", "CPU ms"))
for name, g, cc in host_rows:
    lines.append("%-72s %10.3f %10.2f" % (name, g, cc))
lines.append("# round 3: batched, resident forms (every frame vs its predecessor, one launch for all pairs; GPU ms per batch and per pair) next to the oracle's one-core ms per pair")
lines.append("%-86s %10s %12s %12s" % ("call", "GPU ms", "GPU ms/pair", "CPU ms/pair"))
for name, g, cc in batch_rows:
    lines.append("%-86s %10.3f %12.4f %12.2f" % (name, g, g / npairs, cc))
print("\n".join(lines))
if a.out:
    open(a.out, "w").write("\n".join(lines) + "\n")
