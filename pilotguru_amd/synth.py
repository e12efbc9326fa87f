"""Deterministic synthetic rides for tests and bench.

Pure integer arithmetic (numpy uint64/int64), so the same bytes come out on
every host.  A *scene* is 3-octave value noise (lattice cells 64/16/4 px,
weights 4:2:1) + filled grey rectangles + +-4 pixel noise, clamped to u8.
Frame k of a ride is the scene window translated by (2k, k) px, so consecutive
frames overlap almost entirely and their descriptors match.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser, vectorised over uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def _hash2(seed, salt, ix, iy):
    with np.errstate(over="ignore"):
        k = (np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D)) ^ (np.uint64(salt) << np.uint64(56))
        k = k ^ (np.asarray(ix, np.uint64) << np.uint64(28)) ^ np.asarray(iy, np.uint64)
    return _mix(k)


def _value_noise(seed, salt, w, h, cell):
    """Bilinear value noise with integer weights; returns int64 in [0, 255]."""
    xs = np.arange(w, dtype=np.int64)
    ys = np.arange(h, dtype=np.int64)
    gx, fx = xs // cell, xs % cell
    gy, fy = ys // cell, ys % cell
    GX, GY = np.meshgrid(np.arange(gx.max() + 2), np.arange(gy.max() + 2))
    lat = (_hash2(seed, salt, GX, GY) & np.uint64(255)).astype(np.int64)
    v00 = lat[gy][:, gx]
    v01 = lat[gy][:, gx + 1]
    v10 = lat[gy + 1][:, gx]
    v11 = lat[gy + 1][:, gx + 1]
    FX = fx[None, :]
    FY = fy[:, None]
    top = v00 * (cell - FX) + v01 * FX
    bot = v10 * (cell - FX) + v11 * FX
    return (top * (cell - FY) + bot * FY) // (cell * cell)


def synth_scene(seed, w, h, nrect=None):
    """One grey scene of size h x w (uint8)."""
    n = (_value_noise(seed, 1, w, h, 64) * 4 + _value_noise(seed, 2, w, h, 16) * 2 +
         _value_noise(seed, 3, w, h, 4)) // 7
    img = n.astype(np.int64)
    if nrect is None:
        nrect = max(8, (400 * w * h) // (1920 * 1080))
    r = _mix(np.arange(nrect * 5, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32)) + np.uint64(77))
    r = r.reshape(nrect, 5)
    for i in range(nrect):
        rw = 8 + int(r[i, 0] % np.uint64(57))
        rh = 8 + int(r[i, 1] % np.uint64(57))
        x0 = int(r[i, 2] % np.uint64(max(1, w - rw)))
        y0 = int(r[i, 3] % np.uint64(max(1, h - rh)))
        img[y0:y0 + rh, x0:x0 + rw] = int(r[i, 4] & np.uint64(255))
    X, Y = np.meshgrid(np.arange(w, dtype=np.uint64), np.arange(h, dtype=np.uint64))
    noise = (_hash2(seed, 9, X, Y) % np.uint64(9)).astype(np.int64) - 4
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def synth_ride(seed, w, h, nframes, dx=2, dy=1):
    """nframes x h x w uint8; frame k = scene[k*dy : k*dy+h, k*dx : k*dx+w]."""
    scene = synth_scene(seed, w + dx * (nframes - 1), h + dy * (nframes - 1))
    out = np.empty((nframes, h, w), np.uint8)
    for k in range(nframes):
        out[k] = scene[k * dy:k * dy + h, k * dx:k * dx + w]
    return out


def synth_scene_road(seed, w, h):
    """A driving-like scene: the top 42 % is "sky" (smooth vertical gradient, +-2 noise, a few faint
    clouds whose contrast of 14 grey levels lies between minThFAST = 7 and iniThFAST = 20), the bottom
    30 % is "asphalt" (flat grey, +-3 noise, a few bright lane markings), the band in between is the
    textured scene.  More than half of the 30-px cells hold no corner at iniThFAST, so the detector's
    per-cell retry at minThFAST (ORBextractor.cc:812-816) runs on most of the image."""
    img = synth_scene(seed, w, h).astype(np.int64)
    X, Y = np.meshgrid(np.arange(w, dtype=np.uint64), np.arange(h, dtype=np.uint64))
    ys = np.arange(h, dtype=np.int64)[:, None]
    sky_end, road_beg = (42 * h) // 100, (70 * h) // 100
    sky = 205 - (ys * 40) // max(h, 1) + ((_hash2(seed, 21, X, Y) % np.uint64(5)).astype(np.int64) - 2)
    road = 90 + ((_hash2(seed, 22, X, Y) % np.uint64(7)).astype(np.int64) - 3)
    img[:sky_end] = np.broadcast_to(sky, (h, w))[:sky_end]
    img[road_beg:] = np.broadcast_to(road, (h, w))[road_beg:]
    nobj = max(6, (60 * w * h) // (1920 * 1080))
    r = _mix(np.arange(nobj * 4, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32)) + np.uint64(991)).reshape(nobj, 4)
    for i in range(nobj):
        rw, rh = 10 + int(r[i, 0] % np.uint64(50)), 6 + int(r[i, 1] % np.uint64(20))
        x0 = int(r[i, 2] % np.uint64(max(1, w - rw)))
        if i % 2 == 0 and sky_end > rh + 2:                                   # cloud: +14 on the sky
            y0 = int(r[i, 3] % np.uint64(max(1, sky_end - rh - 1)))
            img[y0:y0 + rh, x0:x0 + rw] += 14
        elif h - road_beg > rh + 2:                                           # lane marking: bright on the asphalt
            y0 = road_beg + 1 + int(r[i, 3] % np.uint64(max(1, h - road_beg - rh - 1)))
            img[y0:y0 + rh, x0:x0 + rw] = 215
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_ride_road(seed, w, h, nframes, dx=2, dy=0):
    """nframes x h x w uint8 of the driving-like scene; the camera pans horizontally (dy = 0 keeps the
    horizon where it is)."""
    scene = synth_scene_road(seed, w + dx * (nframes - 1), h + dy * (nframes - 1))
    out = np.empty((nframes, h, w), np.uint8)
    for k in range(nframes):
        out[k] = scene[k * dy:k * dy + h, k * dx:k * dx + w]
    return out
