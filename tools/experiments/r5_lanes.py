"""Device-resident stream: frames/s by lanes / depth / stagger (PGORB_STREAM_STAGGER is read at stream creation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
W, H, NF, B = 1920, 1080, 2000, 128
ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
frames = torch.from_numpy(synth_ride(0, W, H, B)).cuda()
def run(lanes, depth, stagger, secs=2.0, ahead=0):
    os.environ["PGORB_STREAM_STAGGER"] = str(stagger)
    os.environ["PGORB_STREAM_K1_AHEAD"] = str(ahead)
    st = pg.DeviceFrameStream(ext, W, H, B, depth=depth, lanes=lanes)
    for k in range(2 * depth):
        if k >= depth: st.wait(k % depth, on_host=False)
        st.submit(k % depth, frames)
    for k in range(depth): st.wait(k)
    torch.cuda.synchronize()
    for k in range(depth): st.submit(k, frames)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for k in range(40):
            st.wait(k % depth, on_host=False); st.submit(k % depth, frames)
        st.wait(0); st.submit(0, frames); n += 41
    for k in range(depth): st.wait(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st.close()
    return (n + depth) * B / dt
if len(sys.argv) > 1 and sys.argv[1] == "ahead":
    for lanes, depth, ah in ((2, 2, 0), (2, 2, 1), (2, 4, 1), (3, 3, 1), (2, 2, 0), (2, 2, 1)):
        print("lanes %d depth %d K1-ahead %d: %.0f frames/s" % (lanes, depth, ah, run(lanes, depth, 0, ahead=ah)), flush=True)
    sys.exit(0)
for lanes, depth, stg in ((1, 2, 0), (2, 2, 0), (2, 2, 1), (2, 2, 2), (2, 4, 1), (2, 4, 2), (3, 3, 1), (3, 6, 1), (2, 2, 1), (1, 2, 0)):
    print("lanes %d depth %d stagger %d: %.0f frames/s" % (lanes, depth, stg, run(lanes, depth, stg)), flush=True)
