"""Which cells of the bench scene take the minThFAST retry, per level (oracle score maps; ORBextractor.cc:765-829 cell geometry).
Result: profiles/r05_cell_stats.txt"""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import orb_oracle as oo
from pilotguru_amd.synth import synth_ride
W,H=1920,1080
fr = synth_ride(0, W, H, 2)[1]
o = oo.OrbOracle(2000,1.2,8,20,7)
o.extract(fr)
tot=0; totretry=0
for l in range(8):
    img = o.level_image(l)
    h,w = img.shape
    # cell geometry like ORBextractor.cc:765-829
    E=19
    minBX=E-3; minBY=minBX; maxBX=w-E+3; maxBY=h-E+3
    width=maxBX-minBX; height=maxBY-minBY
    nCols=width//30; nRows=height//30
    wCell=int(np.ceil(width/nCols)); hCell=int(np.ceil(height/nRows))
    s20 = oo.fast9_score_map(img, 20)   # full-image score map at th 20 (scores independent of threshold)
    s7 = oo.fast9_score_map(img, 7)
    ncell=0; nretry=0; cand20=0; cand7r=0
    for i in range(nRows):
        iniY=minBY+i*hCell; maxY=iniY+hCell+6
        if iniY>=maxBY-3: continue
        if maxY>maxBY: maxY=maxBY
        for j in range(nCols):
            iniX=minBX+j*wCell; maxX=iniX+wCell+6
            if iniX>=maxBX-6: continue
            if maxX>maxBX: maxX=maxBX
            ncell+=1
            c20 = (s20[iniY+3:maxY-3, iniX+3:maxX-3]>0).sum()
            cand20+=c20
            if c20==0:
                nretry+=1   # approx (ignores NMS-empty corner case)
                cand7r += (s7[iniY+3:maxY-3, iniX+3:maxX-3]>0).sum()
    print(l, w,h, 'cells',ncell,'retry',nretry, 'frac %.3f'%(nretry/ncell), 'corners20/cell %.1f'%(cand20/ncell), 'corners7 per retry cell %.1f'%(cand7r/max(nretry,1)))
    tot+=ncell; totretry+=nretry
print(tot, totretry, totretry/tot)
