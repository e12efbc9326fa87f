import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
os.environ["PGORB_FAST_DBGSKIP"] = "9"
B=16
ext = pg.ORBextractor(2000,1.2,8,20,7,max_width=1920,max_height=1080,max_batch=B)
fr = torch.from_numpy(synth_ride(0,1920,1080,B)).cuda()
L = ext._L
st = (C.c_ulonglong*8)()
L.pgorb_debug_fast_stats(st, 1)
ext.extract_batch_device(fr); torch.cuda.synchronize()
L.pgorb_debug_fast_stats(st, 0)
print(list(st), "cand/block", st[1]/max(st[0],1))
