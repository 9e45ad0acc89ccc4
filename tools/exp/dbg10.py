import sys, numpy as np
sys.path.insert(0, '.')
from gpu_video_codec_amd import deblock, synth, _lib
from oracle import oracle, h265
ctx = deblock.Context(0)
w,h,n,bd=3840,2160,1,10
fr = np.stack([synth.blocky_plane(w,h,seed=9,frame=i,bit_depth=bd) for i in range(n)])
b = deblock.DeviceBatch(ctx,w,h,n,bit_depth=bd,per_frame_bs=False); b.upload_all(fr)
for types in ("mix","off","band","edge"):
    prm = h265.random_sao_params(w,h,6,seed=19,bit_depth=bd)
    if types!="mix": prm["type"]={"off":0,"band":1,"edge":2}[types]
    d = ctx.alloc(prm.nbytes); d.upload(prm.view(np.uint8).ravel())
    dbk = oracle.filter_plane(fr[0],32,bit_depth=bd,threads=8)
    want = h265.sao_plane(dbk,prm,6,bit_depth=bd)
    b.dst.upload(np.zeros(b.frame_bytes*n,np.uint8))
    ctx.deblock_sao_device(b.planes(),32,d.ptr,prm.shape[1],6,fused=_lib.FUSED_ON); ctx.synchronize()
    g=b.download_frame(0); bad=np.argwhere(g!=want)
    print(types,len(bad))
    seen=set()
    for (y,x) in bad[:400]:
        key=(y//8,x//8)
        if key in seen: continue
        seen.add(key)
        c=prm[y//64,x//64]
        if len(seen)<=12: print('  blk',key,'tile',(y//128,x//128),'in-tile',(y%128,x%128),'ctb type',int(c['type']),'cls',int(c['cls']),'off',c['offset'].tolist(),'got',int(g[y,x]),'want',int(want[y,x]),'dbk',int(dbk[y,x]),'src',int(fr[0][y,x]))
    d.free()
