import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pilotguru_amd as pg
from pilotguru_amd.synth import synth_ride
W,H,NF = 1920,1080,2000
for S, B in ((1,128),(2,64),(4,32),(4,64)):
    exts = [pg.ORBextractor(NF,1.2,8,20,7,max_width=W,max_height=H,max_batch=B) for _ in range(S)]
    ride = synth_ride(0,W,H,B)
    frames = torch.from_numpy(ride).cuda()
    streams = [torch.cuda.Stream() for _ in range(S)]
    cap = exts[0].max_keypoints(W,H)
    bufs = [(torch.empty((B,cap,7),dtype=torch.float32,device='cuda'), torch.empty((B,cap,32),dtype=torch.uint8,device='cuda'), torch.empty((B,),dtype=torch.int32,device='cuda')) for _ in range(S)]
    pq = torch.arange(1,B,dtype=torch.int32,device='cuda'); pt = torch.arange(0,B-1,dtype=torch.int32,device='cuda')
    mouts = [(torch.empty((B-1,cap),dtype=torch.int32,device='cuda'), torch.empty((B-1,cap),dtype=torch.int16,device='cuda'), torch.empty((B-1,cap),dtype=torch.int16,device='cuda')) for _ in range(S)]
    def step():
        for i in range(S):
            s = streams[i].cuda_stream
            exts[i].extract_batch_device(frames, *bufs[i], stream=s)
            exts[i].match_batch_device(bufs[i][1], bufs[i][2], pq, pt, mouts[i], stream=s)
    for _ in range(3): step()
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = time.perf_counter()-t0
    print("streams", S, "batch", B, "fps", round(S*B*K/dt))
    del exts
