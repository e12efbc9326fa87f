"""pilotguru_amd -- MI355X-native ORB front end for pilotguru's optical_trajectories.

Only what the hot path needs: csrc/ (HIP kernels + C ABI, built into libpgorb.so),
the host-side mirror of the reference's ORBextractor / ORBmatcher interface (orb.py),
and the synthetic ride generator used by tests and bench (synth.py).
"""
from .orb import KEYPOINT_DTYPE, DeviceFrameStream, Frame, FrameStream, MapPoints, ORBextractor, ORBmatcher  # noqa: F401
