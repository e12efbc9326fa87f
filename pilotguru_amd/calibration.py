"""Host-side mirror of fit_motion's velocity calibration (src/fit_motion.cc:151-290,
src/calibration/velocity.cc): numpy arrays across the C ABI, the window fits run on the GPU
(pilotguru_amd/csrc/calib.hip).  Series are passed as (values, time_usec) pairs like the
recorder's JSON files hold them: GPS speed [n], gyroscope rates [n][3], accelerations [n][3]."""
import ctypes as C

import numpy as np

from . import _lib


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _series(gps_v, gps_t, rot, rot_t, acc, acc_t):
    a = [_d(gps_v), np.ascontiguousarray(gps_t, np.int64), _d(rot).reshape(-1, 3), np.ascontiguousarray(rot_t, np.int64),
         _d(acc).reshape(-1, 3), np.ascontiguousarray(acc_t, np.int64)]
    if not (len(a[0]) == len(a[1]) and len(a[2]) == len(a[3]) and len(a[4]) == len(a[5])):
        raise ValueError("values and timestamps differ in length")
    return a, [_p(a[0]), _p(a[1]), len(a[0]), _p(a[2]), _p(a[3]), len(a[2]), _p(a[4]), _p(a[5]), len(a[4])]


def _h(ctx):
    """A pgorb context handle, or anything that owns one (pilotguru_amd.ORBextractor)."""
    return getattr(ctx, "_h", ctx)


def _check(ctx, rc):
    if rc != _lib.PGORB_OK:
        raise _lib.PgorbError(rc, _lib.lib().pgorb_last_error(ctx).decode())


class AccelerometerCalibrator:
    """velocity.hpp:39-83: the loss of one window; `__call__(x)` -> (loss, gradient) like operator()."""

    def __init__(self, ctx, reference_velocities, rotation_velocities, accelerations):
        self._ctx = _h(ctx)
        self._keep, self._args = _series(*reference_velocities, *rotation_velocities, *accelerations)

    def __call__(self, x):
        x = _d(x).reshape(-1, 9)
        fx, g = np.zeros(len(x)), np.zeros((len(x), 9))
        _check(self._ctx, _lib.lib().pgorb_calibrator_eval(self._ctx, *self._args, _p(x), len(x), _p(fx), _p(g)))
        return (fx[0], g[0]) if len(x) == 1 else (fx, g)


def FitVelocityWindows(ctx, gps, rotations, accelerations, locations_batch_size=40, locations_shift_step=5,
                       optimization_iters=500):
    """The sliding-window fits of fit_motion.cc:173-190 -> (x[nw][9], residual[nw], iterations[nw])."""
    ctx = _h(ctx)
    keep, args = _series(*gps, *rotations, *accelerations)
    nw = _lib.lib().pgorb_fit_num_windows(len(keep[0]), locations_shift_step)
    x, res, it = np.zeros((nw, 9)), np.zeros(nw), np.zeros(nw, np.int32)
    _check(ctx, _lib.lib().pgorb_fit_velocity_windows(ctx, *args, locations_batch_size, locations_shift_step, optimization_iters,
                                                      _p(x), _p(res), _p(it)))
    return x, res, it


def ComputeForwardVelocitiesFromImu(ctx, gps, rotations, accelerations, vertical_axis, locations_batch_size=40,
                                    locations_shift_step=5, optimization_iters=500, post_smoothing_sigma_sec=0.003,
                                    forward_axis_inference_min_velocity_m_s=5.0, forward_axis_inference_min_rotation_rad=0.2):
    """fit_motion.cc:151-290 up to the JSON writers -> (time_usec[n], speed_m_s[n], forward_axis[3])."""
    ctx = _h(ctx)
    keep, args = _series(*gps, *rotations, *accelerations)
    cap = len(keep[2]) + len(keep[4])
    t, v, fwd, va, n = np.zeros(cap, np.int64), np.zeros(cap), np.zeros(3), _d(vertical_axis), C.c_int32(0)
    _check(ctx, _lib.lib().pgorb_fit_motion_velocities(ctx, *args, _p(va), locations_batch_size, locations_shift_step, optimization_iters,
                                                       post_smoothing_sigma_sec, forward_axis_inference_min_velocity_m_s,
                                                       forward_axis_inference_min_rotation_rad, _p(t), _p(v), C.byref(n), _p(fwd)))
    return t[:n.value].copy(), v[:n.value].copy(), fwd


def GetPrincipalRotationAxes(raw_rotations, integration_interval_usec=500000):
    """rotation.cc:16-57 -> eigenvectors[3][3] (rows); row 0 is taken as the vertical axis (fit_motion.cc:322-329)."""
    rot, t = _d(raw_rotations[0]).reshape(-1, 3), np.ascontiguousarray(raw_rotations[1], np.int64)
    vec = np.zeros((3, 3))
    rc = _lib.lib().pgorb_principal_rotation_axes(_p(rot), _p(t), len(rot), int(integration_interval_usec), _p(vec))
    if rc != _lib.PGORB_OK:
        raise _lib.PgorbError(rc, "GetPrincipalRotationAxes: interval must be > 0 and give at least 3 integrated rotations")
    return vec


def GetAngularVelocitiesAroundAxisDirect(raw_rotations, axis):
    """rotation.cc:111-129: the steering series of fit_motion."""
    rot, a = _d(raw_rotations[0]).reshape(-1, 3), _d(axis)
    out = np.zeros(len(rot))
    rc = _lib.lib().pgorb_angular_velocities_around_axis(_p(rot), len(rot), _p(a), _p(out))
    if rc != _lib.PGORB_OK:
        raise _lib.PgorbError(rc, "GetAngularVelocitiesAroundAxisDirect: the axis must be normalised")
    return out


def KahanSum(values):
    """include/math/math.hpp:8-25: compensated sum of the rows of `values` ([n][dim]), as fit_motion accumulates the
    local-frame velocities."""
    v = _d(values)
    v = v.reshape(len(v), v.shape[-1] if v.ndim > 1 else 1)
    out = np.zeros(v.shape[1])
    rc = _lib.lib().pgorb_kahan_sum(_p(v), v.shape[0], v.shape[1], _p(out))
    if rc != _lib.PGORB_OK:
        raise _lib.PgorbError(rc, "KahanSum")
    return out
