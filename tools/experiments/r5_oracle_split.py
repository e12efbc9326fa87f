"""Where the CPU oracle spends a 1080p / 2000 frame: the three primitives OpenCV 2.4.9 vectorises (resize, FAST, GaussianBlur) timed alone
against a whole extract (the SIMD haircut)."""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import orb_oracle as oo
from pilotguru_amd.synth import synth_ride
W,H=1920,1080
fr = synth_ride(0, W, H, 3)
o = oo.OrbOracle(2000,1.2,8,20,7)
for f in fr: o.extract(f)
t=time.perf_counter(); 
for f in fr: o.extract(f)
tot=(time.perf_counter()-t)/3
lv=[o.level_image(l) for l in range(8)]
def T(fn,n=3):
    fn(); t=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t)/n
tres=T(lambda:[oo.resize_linear(lv[l-1], lv[l].shape[1], lv[l].shape[0]) for l in range(1,8)])
tblur=T(lambda:[oo.gaussian_blur7(x) for x in lv])
tfast=T(lambda:[oo.fast9_nms(x,20) for x in lv])
print("extract %.1f ms; resize chain %.1f (%.0f%%), blur all levels %.1f (%.0f%%), FAST+NMS whole levels at 20 %.1f (%.0f%%)"%(tot*1e3,tres*1e3,100*tres/tot,tblur*1e3,100*tblur/tot,tfast*1e3,100*tfast/tot))
