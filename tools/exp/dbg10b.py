import sys, numpy as np
sys.path.insert(0, '.')
import gpu_video_codec_amd._lib as l
if len(sys.argv) > 1: l.LIB_PATH = sys.argv[1]
from gpu_video_codec_amd import deblock, synth, _lib
from oracle import oracle, h265
ctx = deblock.Context(0)
w,h,n,bd=3840,2160,2,10
fr = np.stack([synth.blocky_plane(w,h,seed=9,frame=i,bit_depth=bd) for i in range(n)])
b = deblock.DeviceBatch(ctx,w,h,n,bit_depth=bd,per_frame_bs=False); b.upload_all(fr)
prm = h265.random_sao_params(w,h,6,seed=19,bit_depth=bd)
d = ctx.alloc(prm.nbytes); d.upload(prm.view(np.uint8).ravel())
want = [h265.sao_plane(oracle.filter_plane(fr[f],32,bit_depth=bd,threads=8),prm,6,bit_depth=bd) for f in range(n)]
tot=0
for it in range(6):
    b.dst.upload(np.zeros(b.frame_bytes*n,np.uint8))
    ctx.deblock_sao_device(b.planes(),32,d.ptr,prm.shape[1],6,fused=_lib.FUSED_ON); ctx.synchronize()
    bad=sum(int((b.download_frame(f)!=want[f]).sum()) for f in range(n)); tot+=bad
    print(sys.argv[1:] , it, bad)
print('TOTAL', tot)
