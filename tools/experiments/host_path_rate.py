"""PCIe-inclusive rate of the host-buffer entry point (pgorb_extract_batch) and the 4K config."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
for (W, H, NF, B) in ((1920, 1080, 2000, 32), (3840, 2160, 4000, 16)):
    ext = pg.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    ride = synth_ride(0, W, H, B)
    import ctypes as C
    from pilotguru_amd import _lib
    L = _lib.lib()
    pin = L.pgorb_host_alloc(ride.nbytes)
    pride = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=ride.shape)
    pride[...] = ride
    for label, src in (("pageable", ride), ("pinned  ", pride)):
        fr_ = [src[i] for i in range(B)]
        ext.extract_batch(fr_)
        t0 = time.perf_counter(); K = 5
        for _ in range(K): out = ext.extract_batch(fr_)
        dt = time.perf_counter() - t0
        print("host-buffer path (%s frames) %dx%d/%d: %.0f frames/s (batch %d)" % (label, W, H, NF, B * K / dt, B))
    frames = [ride[i] for i in range(B)]
    ext.extract_batch(frames)
    t0 = time.perf_counter(); K = 5
    for _ in range(K): out = ext.extract_batch(frames)
    dt = time.perf_counter() - t0
    print("host-buffer path %dx%d/%d: %.0f frames/s (batch %d, incl. H2D upload + D2H results), kp/frame %.0f"
          % (W, H, NF, B * K / dt, B, np.mean([len(o[0]) for o in out])))
    fr = torch.from_numpy(ride).cuda()
    for _ in range(2): ext.extract_batch_device(fr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): r = ext.extract_batch_device(fr)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("resident path    %dx%d/%d: %.0f frames/s (batch %d, extract only)" % (W, H, NF, B * K / dt, B))
    del ext
