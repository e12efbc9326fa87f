"""Edge cases of K9 against the oracle: a single GPS fix, windows of one fix, IMU far sparser than GPS (most intervals
empty) and far denser (many chunks per interval)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_calibration import imu_ride, _bits
import pilotguru_amd as pg
from pilotguru_amd.calibration import FitVelocityWindows, ComputeForwardVelocitiesFromImu
from oracle import orb_oracle as orc
ctx = pg.ORBextractor(500,1.2,4,20,7,max_width=320,max_height=240,max_batch=1)
gps, rot, acc = imu_ride(3, n_gps=6)
for n in (1, 2):
    g = (gps[0][:n], gps[1][:n])
    for batch, shift in ((1,1),(40,5)):
        x,res,it = FitVelocityWindows(ctx, g, rot, acc, batch, shift, 10)
        ox,ores,oit = orc.fit_windows(*g,*rot,*acc,batch,shift,10)
        print(n, batch, shift, np.array_equal(it,oit), np.array_equal(_bits(x),_bits(ox)), np.array_equal(_bits(res),_bits(ores)), it)
# IMU much sparser than GPS (most intervals empty), and very dense IMU (many chunks per interval)
r = np.random.default_rng(0)
tg = np.arange(1,40)*1_000_000 + 10**15
for hz in (0.3, 2000.0):
    n = int(45*hz)+3
    tr = (np.cumsum(r.uniform(0.5,1.5,n))/hz*1e6).astype(np.int64) + 10**15; ta = tr + 137
    tr, ta = np.unique(tr), np.unique(ta)
    R = r.normal(0,0.2,(len(tr),3)); A = r.normal(0,1,(len(ta),3))+[0,0,9.8]; g=(np.full(len(tg),10.0),tg)
    x,res,it = FitVelocityWindows(ctx, g, (R,tr), (A,ta), 40, 5, 30)
    ox,ores,oit = orc.fit_windows(*g,R,tr,A,ta,40,5,30)
    print("hz",hz, np.array_equal(it,oit), np.array_equal(_bits(x),_bits(ox)), np.array_equal(_bits(res),_bits(ores)))
