#!/usr/bin/env python3
"""Randomised soak of the HIP path against the oracles (run by hand on a GPU box; not collected by pytest):
   python tests/soak_gpu.py [--cases 400] [--seed 1]
Every case draws a geometry, bit depth, plane kind, QP (scalar or map), bS arrays, frame count, in-place or not, and runs
both kernels of the reference-exact mode and of the spec-exact mode, 8-bit scalar-QP cases also through deblocking + SAO in one kernel and as two launches, and every third case a whole 4:2:0 batch through
hevc_deblocking_filter_device_planes and the host-frame operator; any mismatch prints the case and exits non-zero."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_video_codec_amd import deblock, synth, _lib
from oracle import oracle, h265


def host_large(ctx, rng, cases):
    """Round 4: the host-frame operators on LARGE frames (the crew / BAR / ring pipeline: above 2 MiB), drawn at random -- geometry,
    bit depth, luma only or 4:2:0, row padding, caller bS, a per-CTU QP map, 1..6 copying threads, some planes page-locked or
    registered, single calls and sequences, and the spec-exact host entry; every plane against the oracles."""
    bad = 0
    for case in range(cases):
        w = int(16 * rng.randint(80, 257))          # 1280 .. 4096
        h = int(16 * rng.randint(68, 137))          # 1088 .. 2176
        bd = int(rng.choice([8, 8, 10, 12]))
        chroma = bool(rng.randint(0, 2))
        pad = int(rng.choice([0, 0, 16, 40]))
        qp = int(rng.randint(20, 52))
        use_map = rng.randint(0, 4) == 0
        user_bs = bool(rng.randint(0, 2))
        threads = int(rng.choice([1, 2, 3, 4, 6]))
        kind = rng.choice(["frame", "frame", "sequence", "h265"])
        if kind != "frame":
            use_map = False
        dt = np.uint8 if bd == 8 else np.uint16
        ctx.set_host_threads(threads)
        tag = dict(case=case, w=w, h=h, bd=bd, chroma=chroma, pad=pad, qp=qp, use_map=use_map, user_bs=user_bs, threads=threads, kind=str(kind))
        nfr = int(rng.randint(2, 6)) if kind == "sequence" else 1
        qmap = rng.randint(max(qp - 8, 0), min(qp + 8, 51) + 1, ((h + 63) // 64, (w + 63) // 64)).astype(np.uint8) if use_map else None
        vb, hb = oracle.lcg_bs(w, h, int(rng.randint(1, 1000))) if user_bs else (None, None)
        frames, bufs, wants, locked = [], [], [], []
        for f in range(nfr):
            pl = synth.blocky_yuv420(w, h, seed=int(rng.randint(1, 1 << 30)), bit_depth=bd) if chroma else \
                (synth.blocky_plane(w, h, seed=int(rng.randint(1, 1 << 30)), bit_depth=bd),)
            if kind == "h265":
                vb4 = (rng.randint(0, 3, h265.num_vert_bs(w, h)) | (rng.randint(0, 10, h265.num_vert_bs(w, h)) == 0) * 4).astype(np.uint8)
                hb4 = (rng.randint(0, 3, h265.num_hor_bs(w, h)) | (rng.randint(0, 10, h265.num_hor_bs(w, h)) == 0) * 8).astype(np.uint8)
                wants.append([h265.filter_plane(pl[0], min(qp, 51), vb4, hb4, bit_depth=bd)])
                pl = (pl[0],)
            else:
                wants.append([oracle.filter_plane(p, 0 if use_map else qp, bit_depth=bd, is_chroma=k > 0, vert_bs=vb if k == 0 else None,
                                                  hor_bs=hb if k == 0 else None, qp_map=qmap, threads=8) for k, p in enumerate(pl)])
            bb = []
            for p in pl:
                shape = (p.shape[0], p.shape[1] + pad)
                how = int(rng.randint(0, 6)) if kind == "frame" else 0     # 0-3 pageable, 4 page-locked, 5 registered
                b = ctx.pinned_array(shape, dt) if how == 4 else np.empty(shape, dt)
                if how == 5:
                    ctx.host_register(b)
                locked.append((how, b))
                b[:] = 0x3C
                b[:, :p.shape[1]] = p
                bb.append(b)
            bufs.append(bb)
            frames.append(tuple(b[:, :b.shape[1] - pad] if pad else b for b in bb))
        try:
            if kind == "frame":
                ctx.filter_frame(*frames[0], qp=0 if use_map else qp, bit_depth=bd, vert_bs=vb, hor_bs=hb, qp_map=qmap)
            elif kind == "sequence":
                ctx.filter_sequence(frames, qp=qp, bit_depth=bd, vert_bs=vb, hor_bs=hb)
            else:
                ctx.filter_frame_h265(frames[0][0], qp=min(qp, 51), bit_depth=bd, vert_bs4=vb4, hor_bs4=hb4)
            for f in range(nfr):
                for k in range(len(frames[f])):
                    if not np.array_equal(frames[f][k], wants[f][k]):
                        print("MISMATCH host-large", f, k, tag)
                        bad += 1
                    if pad and not (bufs[f][k][:, -pad:] == 0x3C).all():
                        print("PADDING TOUCHED host-large", f, k, tag)
                        bad += 1
        finally:
            for how, b in locked:
                if how == 4:
                    ctx.free_pinned(b)
                elif how == 5:
                    ctx.host_unregister(b)
        if case % 10 == 9:
            print("soak host-large: %d cases, %d mismatches" % (case + 1, bad), flush=True)
    ctx.set_host_threads(0)
    print("soak host-large done: %d cases, %d mismatches" % (cases, bad))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--host-large", type=int, default=0, metavar="N", help="N random LARGE host frames / sequences through the host operators instead")
    a = ap.parse_args()
    rng = np.random.RandomState(a.seed)
    ctx = deblock.Context(0)
    if a.host_large:
        sys.exit(1 if host_large(ctx, rng, a.host_large) else 0)
    bad = 0
    for case in range(a.cases):
        w = int(8 * rng.choice([rng.randint(1, 20), rng.randint(60, 70), rng.randint(125, 135), rng.randint(1, 600)]))
        h = int(8 * rng.randint(1, 24))
        bd = int(rng.choice([8, 8, 8, 10, 11, 12]))
        chroma = bool(rng.randint(0, 2))
        n = int(rng.randint(1, 4))
        in_place = bool(rng.randint(0, 2))
        qp = int(rng.randint(16, 56))
        use_map = rng.randint(0, 4) == 0
        sc = 2 if chroma else 1
        frames = np.stack([synth.blocky_plane(w, h, seed=int(rng.randint(1, 1 << 30)), bit_depth=bd) for _ in range(n)])
        if rng.randint(0, 3) == 0:  # rough content: more clipping, more 'off' segments
            frames = np.clip(frames.astype(np.int64) + rng.randint(-40, 41, frames.shape) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(frames.dtype)
        qmap = None
        if use_map:
            qmap = rng.randint(max(qp - 10, 0), min(qp + 10, 51) + 1, ((h * sc + 63) // 64, (w * sc + 63) // 64)).astype(np.uint8)
        rvb, rhb = oracle.lcg_bs(w, h, int(rng.randint(1, 1000))) if rng.randint(0, 2) else oracle.default_bs(w, h)
        tag = dict(case=case, w=w, h=h, bd=bd, chroma=chroma, n=n, in_place=in_place, qp=qp, use_map=use_map)
        want_ref = [oracle.filter_plane(frames[f], 0 if use_map else min(qp, 60), is_chroma=chroma, bit_depth=bd, vert_bs=rvb, hor_bs=rhb,
                                        qp_map=qmap) for f in range(n)]
        vb = (rng.randint(0, 3, h265.num_vert_bs(w, h)) | (rng.randint(0, 10, h265.num_vert_bs(w, h)) == 0) * 4).astype(np.uint8)
        hb = (rng.randint(0, 3, h265.num_hor_bs(w, h)) | (rng.randint(0, 10, h265.num_hor_bs(w, h)) == 0) * 8).astype(np.uint8)
        offs = dict(tc_offset_div2=int(rng.randint(-6, 7)), beta_offset_div2=int(rng.randint(-6, 7)))
        cq = int(rng.randint(-12, 13))
        qp_s = min(qp, 51)
        smap = None if qmap is None else np.repeat(np.repeat(qmap, 8, 0), 8, 1)[: (h * sc + 7) // 8, : (w * sc + 7) // 8].copy()
        want_spec = [h265.filter_plane(frames[f], qp_s, vb, hb, c_idx=1 if chroma else 0, bit_depth=bd, qp_map=smap, unit_log2=3,
                                       c_qp_offset=cq if chroma else 0, **offs) for f in range(n)]
        for variant in (_lib.KERNEL_GENERIC, _lib.KERNEL_PACKED):
            b = deblock.DeviceBatch(ctx, w, h, n, bit_depth=bd, is_chroma=chroma, in_place=in_place, per_frame_bs=False)
            try:
                b.upload_all(frames)
                b.set_bs(0, rvb, rhb)
                if qmap is not None:
                    b.set_qp_map(qmap, 6)
                try:
                    ctx.filter_device(b.planes(), 0 if use_map else qp, variant=variant)
                    ctx.synchronize()
                    for f in range(n):
                        if not np.array_equal(b.download_frame(f), want_ref[f]):
                            print("MISMATCH ref", variant, f, tag)
                            bad += 1
                except deblock.DeblockError as e:
                    if e.code != _lib.ERR_UNSUPPORTED:
                        raise
                b.upload_all(frames)
                dv, dh = ctx.alloc(vb.size), ctx.alloc(hb.size)
                dv.upload(vb)
                dh.upload(hb)
                p = b.planes()
                p.vert_bs, p.hor_bs, p.vert_bs_stride, p.hor_bs_stride = dv.ptr, dh.ptr, 0, 0
                dm = None
                if smap is not None:
                    dm = ctx.alloc(smap.size)
                    dm.upload(smap)
                    p.qp_map, p.qp_map_stride, p.ctu_log2, p.qp_map_frame_stride = dm.ptr, smap.shape[1], 3, 0
                else:
                    p.qp_map = None
                try:
                    ctx.filter_device_h265(p, qp_s, c_idx=1 if chroma else 0, cb_qp_offset=cq, variant=variant, **offs)
                    ctx.synchronize()
                    for f in range(n):
                        if not np.array_equal(b.download_frame(f), want_spec[f]):
                            print("MISMATCH spec", variant, f, tag, offs, cq)
                            bad += 1
                except deblock.DeblockError as e:
                    if e.code != _lib.ERR_UNSUPPORTED:
                        raise
                dv.free()
                dh.free()
                if dm:
                    dm.free()
            finally:
                if b.qp_map is not None:
                    b.qp_map.free()
                b.free()
        # cases up to 12 bit (one QP or the QP map) also through deblocking + SAO in one kernel (and as two launches), both filter modes
        if bd <= 12 and not in_place:
            ctb_log2 = int(rng.choice([3, 4, 5, 6]))
            prm = np.stack([h265.random_sao_params(w, h, ctb_log2, seed=int(rng.randint(1, 1 << 30)), bit_depth=bd) for _ in range(n)])
            keep = (rng.randint(0, 5, (n, h // 8, w // 8)) == 0).astype(np.uint8)
            dp, dk = ctx.alloc(prm.nbytes), ctx.alloc(keep.nbytes)
            dp.upload(prm.view(np.uint8).ravel())
            dk.upload(keep.ravel())
            kw = dict(params_frame_stride=prm.shape[1] * prm.shape[2], keep_ptr=dk.ptr, keep_stride=w // 8, keep_frame_stride=(h // 8) * (w // 8))
            b = deblock.DeviceBatch(ctx, w, h, n, bit_depth=bd, is_chroma=chroma, per_frame_bs=False)
            b.upload_all(frames)
            b.set_bs(0, rvb, rhb)
            if qmap is not None:
                b.set_qp_map(qmap, 6)
            want = [h265.sao_plane(want_ref[f], prm[f], ctb_log2, bit_depth=bd, keep=keep[f]) for f in range(n)]
            for fused in (_lib.FUSED_ON, _lib.FUSED_OFF):
                ctx.deblock_sao_device(b.planes(), 0 if use_map else qp, dp.ptr, prm.shape[2], ctb_log2, fused=fused, **kw)
                ctx.synchronize()
                for f in range(n):
                    if not np.array_equal(b.download_frame(f), want[f]):
                        print("MISMATCH deblock+sao ref", fused, f, tag, ctb_log2)
                        bad += 1
            dv, dh = ctx.alloc(vb.size), ctx.alloc(hb.size)
            dv.upload(vb)
            dh.upload(hb)
            p = b.planes()
            p.vert_bs, p.hor_bs, p.vert_bs_stride, p.hor_bs_stride = dv.ptr, dh.ptr, 0, 0
            dm2 = None
            if smap is not None:
                dm2 = ctx.alloc(smap.nbytes)
                dm2.upload(smap)
                p.qp_map, p.qp_map_stride, p.ctu_log2, p.qp_map_frame_stride = dm2.ptr, smap.shape[1], 3, 0
            want = [h265.sao_plane(want_spec[f], prm[f], ctb_log2, bit_depth=bd, keep=keep[f]) for f in range(n)]
            for fused in (_lib.FUSED_ON, _lib.FUSED_OFF):
                ctx.deblock_sao_h265_device(p, qp_s, dp.ptr, prm.shape[2], ctb_log2, c_idx=1 if chroma else 0, cb_qp_offset=cq, fused=fused, **offs, **kw)
                ctx.synchronize()
                for f in range(n):
                    if not np.array_equal(b.download_frame(f), want[f]):
                        print("MISMATCH deblock+sao spec", fused, f, tag, ctb_log2, offs, cq)
                        bad += 1
            for x in (dp, dk, dv, dh) + ((dm2,) if dm2 is not None else ()) + ((b.qp_map,) if b.qp_map is not None else ()):
                x.free()
            b.free()
        # every third case also as a whole 4:2:0 frame batch: Y, U, V in one call (the fused launch where it applies, 8-bit and
        # 16-bit containers) and through the host-frame operator (small frames: no DMA; large: strips)
        if case % 3 == 0 and w % 16 == 0 and h % 16 == 0:
            n2 = int(rng.randint(1, 3))
            yuv = [synth.blocky_yuv420(w, h, seed=int(rng.randint(1, 1 << 30)), bit_depth=bd) for _ in range(n2)]
            want = [[oracle.filter_plane(fr[0], min(qp, 60), bit_depth=bd), oracle.filter_plane(fr[1], min(qp, 60), bit_depth=bd, is_chroma=True),
                     oracle.filter_plane(fr[2], min(qp, 60), bit_depth=bd, is_chroma=True)] for fr in yuv]
            bat = []
            for i, ch in ((0, False), (1, True), (2, True)):
                pl = np.stack([fr[i] for fr in yuv])
                bb = deblock.DeviceBatch(ctx, pl.shape[2], pl.shape[1], n2, bit_depth=bd, is_chroma=ch, in_place=in_place)
                bb.upload_all(pl)
                bat.append(bb)
            ctx.filter_device_planes([bb.planes() for bb in bat], qp)
            ctx.synchronize()
            for i, bb in enumerate(bat):
                for f in range(n2):
                    if not np.array_equal(bb.download_frame(f), want[f][i]):
                        print("MISMATCH planes", i, f, tag)
                        bad += 1
                bb.free()
            got = [p.copy() for p in yuv[0]]
            ctx.filter_frame(*got, qp=min(qp, 60), bit_depth=bd)
            for i in range(3):
                if not np.array_equal(got[i], want[0][i]):
                    print("MISMATCH host frame", i, tag)
                    bad += 1
        if case % 50 == 49:
            print("soak: %d cases, %d mismatches" % (case + 1, bad), flush=True)
    print("soak done: %d cases, %d mismatches" % (a.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
