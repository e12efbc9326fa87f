"""The reference's call shape, measured: ORBextractor::operator() is called synchronously, one frame per call, from a
pageable cv::Mat (thirdparty/orb-slam2/src/Frame.cc:251-257, Tracking.cc:262-266) -- INTEGRATION.md section 1's adaptor does
exactly that with pgorb_extract.  Wall clock per call (p50 / p99 / mean over `calls` calls after a warm-up), the kernel stage
times of the same calls (HIP events) and the host phases the library reports; then the calls Tracking makes per frame
around it: + Frame grid + SearchForInitialization (before initialisation), + the BoW transform (after).
usage: python tools/single_frame_bench.py [--calls 2000] [--width 1920 --height 1080 --features 2000] [--out file]
bench.py's `single_frame` leg imports run()."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _pct(ts):
    a = np.sort(np.asarray(ts, np.float64)) / 1e3
    return {"p50_us": round(float(a[len(a) // 2]), 1), "p99_us": round(float(a[int(len(a) * 0.99)]), 1), "mean_us": round(float(a.mean()), 1),
            "min_us": round(float(a[0]), 1)}


def run(width=1920, height=1080, features=2000, calls=2000, warmup=100, with_frontend=True, options=()):
    import pilotguru_amd as pg
    from pilotguru_amd.synth import synth_ride
    from pilotguru_amd import vocab as V
    ride = synth_ride(0, width, height, 2)                       # two frames of a ride: consecutive frames match
    ext = pg.ORBextractor(features, 1.2, 8, 20, 7, max_width=width, max_height=height, max_batch=1)
    L, h = ext._L, ext._h
    for kv in options:
        k, v = kv.split("="); ext.set_option(k, int(v))
    cap = ext.max_keypoints(width, height)
    kps = np.zeros(cap, pg.orb.KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int(0)
    frames = [np.ascontiguousarray(ride[0]), np.ascontiguousarray(ride[1])]          # pageable host memory, like a cv::Mat
    pk, pd = C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data)
    pf = [C.c_void_p(f.ctypes.data) for f in frames]

    def call(i):
        rc = L.pgorb_extract(h, pf[i & 1], width, height, width, pk, pd, cap, C.byref(n))
        if rc: ext._check(rc)
    for i in range(warmup): call(i)
    out = {"workload": "%dx%d grayscale, %d features, ONE frame per synchronous call from pageable host memory (pgorb_extract)" % (width, height, features),
           "calls": calls, "keypoints": int(n.value)}
    has_host = hasattr(L, "pgorb_profile_host")
    if has_host:
        L.pgorb_profile_host(h, None, 1)
    ts = []
    for i in range(calls):                                       # the timed loop: nothing armed, the library replays its graph
        t0 = time.perf_counter_ns(); call(i); ts.append(time.perf_counter_ns() - t0)
    out["extract"] = _pct(ts)
    if has_host:
        us = (C.c_double * 8)()
        k = L.pgorb_profile_host(h, us, 1)
        names = ["upload_issue", "kernel_launch_issue", "wait_for_gpu", "copy_out"]
        out["extract"]["host_phase_us"] = {nm: round(us[i] / max(k, 1), 1) for i, nm in enumerate(names)}
    # kernel stage times from a second, shorter loop with the stage profile armed (HIP events between the kernels: the library
    # launches directly then, so the wall clock of THESE calls is not the number above)
    pc = min(calls, 300)
    ext.profile_begin(pc)
    ts = []
    for i in range(pc):
        t0 = time.perf_counter_ns(); call(i); ts.append(time.perf_counter_ns() - t0)
    ncalls, ms = ext.profile_read()
    out["extract"]["kernel_stage_us"] = {k: round(v * 1e3, 1) for k, v in ms.items() if k != "match"}
    out["extract"]["kernel_sum_us"] = round(sum(v for k, v in ms.items() if k != "match") * 1e3, 1)
    out["extract"]["p50_us_direct_launches_profiled"] = _pct(ts)["p50_us"]
    # the same calls with the caller's two frame buffers page-locked once (pgorb_host_register: what a decoder that reuses its buffers
    # would do): the upload is ONE DMA out of the caller's pages, the host thread's 2 MB staging memcpy is gone
    regs = [L.pgorb_host_register(pf[i], frames[i].nbytes) for i in range(2)]
    if all(r == 0 for r in regs):
        for i in range(warmup): call(i)
        if has_host:
            L.pgorb_profile_host(h, None, 1)
        ts = []
        for i in range(calls):
            t0 = time.perf_counter_ns(); call(i); ts.append(time.perf_counter_ns() - t0)
        out["extract_registered_caller_buffer"] = _pct(ts)
        if has_host:
            us = (C.c_double * 8)()
            k = L.pgorb_profile_host(h, us, 1)
            out["extract_registered_caller_buffer"]["host_phase_us"] = {nm: round(us[i] / max(k, 1), 1) for i, nm in enumerate(["upload_issue", "kernel_launch_issue", "wait_for_gpu", "copy_out"])}
    for i in range(2):
        if regs[i] == 0: L.pgorb_host_unregister(pf[i])
    if with_frontend:
        # what Tracking does with the frame before the map is initialised: Frame::Frame = extract + undistort (identity here) + grid
        # (Frame.cc:178-232), then ORBmatcher(0.9, true).SearchForInitialization(mInitialFrame, mCurrentFrame, ..., 100) (Tracking.cc:596-597)
        F1 = pg.Frame(ext, frames[0])
        prev0 = np.stack([F1.mvKeys["x"], F1.mvKeys["y"]], 1).astype(np.float32)
        m = pg.ORBmatcher(0.9, True)
        ts = []
        for i in range(max(calls // 4, 50) + 20):
            t0 = time.perf_counter_ns()
            F2 = pg.Frame(ext, frames[1])
            nm, m12 = m.SearchForInitialization(F1, F2, prev0.copy(), 100)
            if i >= 20: ts.append(time.perf_counter_ns() - t0)
        out["extract_grid_search_for_initialization"] = dict(_pct(ts), matches=int(nm), note="python wrappers included (pg.Frame + ORBmatcher)")
        # after initialisation every frame gets its BoW vectors (Frame::ComputeBoW, Frame.cc:399-406) on a k = 10, L = 6 tree
        dsc, wgt, par = V.synth_vocabulary_fast(10, 6, seed=7)
        voc = V.ORBVocabulary(blob=V.pack_vocabulary(10, 6, dsc, wgt, par))
        voc.upload(ext)
        ts = []
        for i in range(max(calls // 4, 50) + 20):
            t0 = time.perf_counter_ns()
            F2 = pg.Frame(ext, frames[i & 1])
            bv, fv = voc.transform(F2.mDescriptors, 4)
            if i >= 20: ts.append(time.perf_counter_ns() - t0)
        out["extract_grid_bow_transform"] = dict(_pct(ts), words=int(len(bv[0])), note="python wrappers included; BowVector / FeatureVector maps built on the host")
        # the same two per-frame sequences through the C ABI alone, into preallocated buffers -- what a C++ caller (the adaptor of
        # INTEGRATION.md section 1) pays; the difference from the two numbers above is the Python wrappers
        bx = [0.0, float(width), 0.0, float(height)]             # no distortion: the image bounds (Frame.cc:461-466)
        gs, gi = np.zeros(64 * 48 + 1, np.int32), np.zeros(cap, np.int32)
        k1, d1, n1 = kps.copy(), desc.copy(), C.c_int(0)
        ext._check(L.pgorb_extract(h, pf[0], width, height, width, C.c_void_p(k1.ctypes.data), C.c_void_p(d1.ctypes.data), cap, C.byref(n1)))
        prev = np.zeros((cap, 2), np.float32); m12 = np.zeros(cap, np.int32)
        prev0 = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
        P = lambda a: C.c_void_p(a.ctypes.data)
        f4 = [C.c_float(v) for v in bx]
        ts = []
        for i in range(max(calls // 4, 50) + 20):
            prev[:n1.value] = prev0[:n1.value]
            t0 = time.perf_counter_ns()
            rc = L.pgorb_extract(h, pf[1], width, height, width, pk, pd, cap, C.byref(n))
            rc = rc or L.pgorb_frame_grid(h, pk, n.value, *f4, P(gs), P(gi))
            nm = L.pgorb_search_for_initialization(h, P(k1), P(d1), n1.value, pk, pd, n.value, *f4, P(prev), P(m12), 100, 0.9, 1)
            if i >= 20: ts.append(time.perf_counter_ns() - t0)
            if rc or nm < 0: ext._check(rc or nm)
        out["c_abi_extract_grid_search_for_initialization"] = dict(_pct(ts), matches=int(nm))
        word, weight, node = np.zeros(cap, np.uint32), np.zeros(cap, np.float64), np.zeros(cap, np.uint32)
        bid, bval = np.zeros(cap + 1, np.uint32), np.zeros(cap + 1, np.float64)
        fnode, fstart, ffeat = np.zeros(cap + 1, np.uint32), np.zeros(cap + 2, np.int32), np.zeros(cap + 1, np.uint32)
        nb, nfv = C.c_int32(), C.c_int32()
        ts = []
        for i in range(max(calls // 4, 50) + 20):
            t0 = time.perf_counter_ns()
            rc = L.pgorb_extract(h, pf[i & 1], width, height, width, pk, pd, cap, C.byref(n))
            rc = rc or L.pgorb_frame_grid(h, pk, n.value, *f4, P(gs), P(gi))
            rc = rc or L.pgorb_bow_transform(h, pd, n.value, 4, P(word), P(weight), P(node))
            rc = rc or L.pgorb_bow_vectors(n.value, P(word), P(weight), P(node), voc.scoring, voc.weighting, P(bid), P(bval), C.byref(nb),
                                           P(fnode), P(fstart), P(ffeat), C.byref(nfv))
            if i >= 20: ts.append(time.perf_counter_ns() - t0)
            if rc: ext._check(rc)
        out["c_abi_extract_grid_bow_transform"] = dict(_pct(ts), words=int(nb.value))
    ext.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--features", type=int, default=2000); ap.add_argument("--calls", type=int, default=2000)
    ap.add_argument("--no-frontend", action="store_true"); ap.add_argument("--out")
    ap.add_argument("--option", action="append", default=[], help="key=value for pgorb_set_option (repeatable)")
    a = ap.parse_args()
    r = run(a.width, a.height, a.features, a.calls, with_frontend=not a.no_frontend, options=a.option)
    txt = json.dumps(r, indent=1)
    print(txt)
    if a.out: open(a.out, "w").write(txt + "\n")
