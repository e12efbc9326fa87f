"""Host-side mirror of what pilotguru does to a finished trajectory before writing the JSON.

Same names and argument meaning as the reference (src/slam/smoothing.cc,
src/slam/horizontal_flatten.cc, src/slam/track_image_sequence.cc:16-29,63-99) so the parity
tests read like tests of the reference functions.  The arithmetic lives in libpgorb.so
(pilotguru_amd/csrc/post.cc); this module only moves numpy arrays across the C ABI.

A trajectory is a pair of arrays: translations [n][3] and rotations [n][4] as (w, x, y, z).
"""
import ctypes as C

import numpy as np

from . import _lib


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc != _lib.PGORB_OK:
        raise _lib.PgorbError(rc, what)


def SmoothHeadingDirections(rotations, sigma):
    """smoothing.cc:11-47.  Returns the smoothed, renormalised rotations; sigma in frames, > 0."""
    q = _d(rotations).reshape(-1, 4).copy()
    _check(_lib.lib().pgorb_smooth_heading_directions(_p(q), len(q), int(sigma)),
           "SmoothHeadingDirections: sigma must be > 0")
    return q


def SmoothTimeSeries(data_values, data_timestamps, target_timestamps, sigma):
    """smoothing.cc:57-97."""
    v, t, g = _d(data_values), _d(data_timestamps), _d(target_timestamps)
    if len(v) != len(t):
        raise ValueError("CHECK_EQ(data_timestamps.size(), data_values.size())")
    out = np.zeros(len(g), np.float64)
    _check(_lib.lib().pgorb_smooth_time_series(_p(v), _p(t), len(v), _p(g), len(g), float(sigma), _p(out)),
           "SmoothTimeSeries: sigma must be > 0 and the series non-empty")
    return out


def TrajectoryToPCA(translations):
    """track_image_sequence.cc:16-29 -> (eigenvectors[3][3] rows, eigenvalues[3], mean[3])."""
    t = _d(translations).reshape(-1, 3)
    vec, val, mean = np.zeros((3, 3)), np.zeros(3), np.zeros(3)
    _check(_lib.lib().pgorb_trajectory_pca(_p(t), len(t), _p(vec), _p(val), _p(mean)),
           "TrajectoryToPCA needs at least 3 poses")
    return vec, val, mean


def ProjectDirections(rotations, projection_plane):
    """horizontal_flatten.cc:7-30 -> [n][2]."""
    q, pl = _d(rotations).reshape(-1, 4), _d(projection_plane).reshape(2, 3)
    out = np.zeros((len(q), 2))
    _check(_lib.lib().pgorb_project_directions(_p(q), len(q), _p(pl), _p(out)), "ProjectDirections")
    return out


def ProjectTranslations(translations, projection_plane):
    """horizontal_flatten.cc:32-43 -> projected copy."""
    t, pl = _d(translations).reshape(-1, 3).copy(), _d(projection_plane).reshape(2, 3)
    _check(_lib.lib().pgorb_project_translations(_p(t), len(t), _p(pl)), "ProjectTranslations")
    return t


def Projected2DDirectionsToTurnAngles(directions):
    """horizontal_flatten.cc:45-63."""
    d = _d(directions).reshape(-1, 2)
    out = np.zeros(len(d))
    _check(_lib.lib().pgorb_turn_angles(_p(d), len(d), _p(out)), "Projected2DDirectionsToTurnAngles")
    return out


def FlattenTrajectory(translations, rotations, rotation_smooth_sigma=-1):
    """The tail of TrackImageSequence (track_image_sequence.cc:68-99) up to the JSON writer.

    Returns None when the reference drops the trajectory (third eigenvalue above 1 % of the
    second, :83-90), else a dict with the (possibly smoothed) rotations, the plane (2x3),
    projected directions and turn angles -- what SetPlane / SetTrajectory are given.
    """
    q = _d(rotations).reshape(-1, 4)
    if rotation_smooth_sigma > 0:
        q = SmoothHeadingDirections(q, rotation_smooth_sigma)
    vec, val, _ = TrajectoryToPCA(translations)
    if val[2] > val[1] * 1e-2:
        return None
    plane = vec[:2].copy()
    dirs = ProjectDirections(q, plane)
    return {"rotations": q, "plane": plane, "eigenvalues": val, "projected_directions": dirs,
            "turn_angles": Projected2DDirectionsToTurnAngles(dirs)}
