"""Python binding of the C ABI (include/hevc_deblock.h) -- plumbing for tests and bench.py.

The compute path is libhevcdbk.so (hand-written HIP, gfx950).  Nothing here computes pixels.

`ReadYuvFrame` mirrors the reference's class of the same name
(hevc_deblocking_filter_cpu.h:33-132, 995-1018: ctor(file, w, h, Qp) / SetBoundaryStrenght /
DeblockingFilter / Save) so that tests read like a test of the reference would; its
DeblockingFilter() runs on the GPU through hevc_deblocking_filter().
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (DeblockError, KERNEL_AUTO, KERNEL_GENERIC, KERNEL_PACKED)  # noqa: F401


def _chk(rc, ctx=None):
    if rc != 0:
        detail = ""
        if ctx is not None and rc == _lib.ERR_HIP:
            detail = _lib.lib().hevcdbk_last_error(ctx).decode()
        raise DeblockError(rc, detail)


def num_vert_bs(w, h):
    return int(_lib.lib().hevcdbk_num_vert_bs(w, h))


def num_hor_bs(w, h):
    return int(_lib.lib().hevcdbk_num_hor_bs(w, h))


def default_bs(w, h):
    v = np.empty(num_vert_bs(w, h), np.uint8)
    hh = np.empty(num_hor_bs(w, h), np.uint8)
    _chk(_lib.lib().hevcdbk_default_bs(w, h, v.ctypes.data, hh.ctypes.data))
    return v, hh


def default_tables():
    L = _lib.lib()
    return (np.array(L.hevcdbk_default_tc_table().contents, np.uint32),
            np.array(L.hevcdbk_default_beta_table().contents, np.uint32))


def device_count():
    return int(_lib.lib().hevcdbk_device_count())


def _tables(tc, beta):
    if tc is None and beta is None:
        return None, []
    t = _lib.Tables()
    keep = []
    for name, arr in (("tc", tc), ("beta", beta)):
        if arr is not None:
            a = np.ascontiguousarray(arr, np.uint32)
            keep.append(a)
            setattr(t, name, a.ctypes.data)
    return t, keep


def filter_yuv_file_multi(devices, in_name, out_name, width, height, qp, *, vert_bs=None, hor_bs=None):
    """hevcdbk_filter_yuv_file_multi: the file operator sharded frame-parallel over `devices` (one worker thread and one
    context per entry; no collective).  Returns (n_frames, wall seconds)."""
    bs, keep = None, []
    if vert_bs is not None:
        bs = _lib.Bs()
        for nm, arr in (("vert", vert_bs), ("hor", hor_bs)):
            a = np.ascontiguousarray(arr, np.uint8)
            keep.append(a)
            setattr(bs, nm, a.ctypes.data)
            setattr(bs, "n_" + nm, a.size)
    dev = (C.c_int * len(devices))(*devices)
    n, tm = C.c_uint(0), _lib.Timing()
    rc = _lib.lib().hevcdbk_filter_yuv_file_multi(dev, len(devices), os.fsencode(in_name), os.fsencode(out_name), width, height,
                                                  int(qp), None if bs is None else C.byref(bs), None, C.byref(n), C.byref(tm))
    _chk(rc)
    return n.value, tm.pipelined_s


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        _chk(_lib.lib().hevcdbk_device_malloc(ctx.handle, self.nbytes, C.byref(p)), ctx.handle)
        self.ptr = p.value

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        _chk(_lib.lib().hevcdbk_memcpy_h2d(self.ctx.handle, self.ptr + offset, a.ctypes.data, a.nbytes), self.ctx.handle)

    def download(self, nbytes=None, offset=0, dtype=np.uint8):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes, np.uint8)
        _chk(_lib.lib().hevcdbk_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + offset, nbytes), self.ctx.handle)
        return out.view(dtype)

    def free(self):
        if self.ptr:
            _lib.lib().hevcdbk_device_free(self.ctx.handle, self.ptr)
            self.ptr = None

    @classmethod
    def adopt(cls, ctx, ptr, nbytes):
        """A DeviceBuffer around memory the library allocated (hevcdbk_device_malloc_probed); free() releases it."""
        b = cls.__new__(cls)
        b.ctx, b.nbytes, b.ptr = ctx, int(nbytes), ptr
        return b


class Context:
    """hevcdbk_context: one per HIP device; owns the compute stream and the two copy streams."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _chk(_lib.lib().hevcdbk_create(device, C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            _lib.lib().hevcdbk_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_info(self):
        i = _lib.DeviceInfo()
        _chk(_lib.lib().hevcdbk_get_device_info(self.handle, C.byref(i)))
        return {"name": i.name.decode(), "gcn_arch": i.gcn_arch.decode(), "compute_units": i.compute_units,
                "wavefront_size": i.wavefront_size, "total_global_mem": i.total_global_mem}

    def synchronize(self):
        _chk(_lib.lib().hevcdbk_synchronize(self.handle), self.handle)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # -- hevc_deblocking_filter(frame, bS, QP, tables): host planes, in place --------------------
    def filter_frame(self, y, u=None, v=None, *, qp, bit_depth=8, vert_bs=None, hor_bs=None,
                     chroma_vert_bs=None, chroma_hor_bs=None, qp_map=None, ctu_log2=6,
                     tc_table=None, beta_table=None, check_sizes=True, want_timing=True):
        """Filters the given 2-D numpy planes IN PLACE (they must be writable, C-contiguous rows).
        Returns the reference's timing triple as a dict."""
        sample_bytes = y.dtype.itemsize
        fr = _lib.Frame()
        fr.height, fr.width = y.shape
        fr.bit_depth, fr.sample_bytes = bit_depth, sample_bytes
        planes = [y] + ([u, v] if u is not None else [])
        for i, p in enumerate(planes):
            assert p.flags.writeable and p.strides[1] == p.itemsize
            fr.plane[i] = p.ctypes.data
            fr.pitch[i] = p.strides[0]
        bs = None
        keep = []
        if vert_bs is not None or chroma_vert_bs is not None:
            bs = _lib.Bs()
            for nm, arr in (("vert", vert_bs), ("hor", hor_bs), ("chroma_vert", chroma_vert_bs), ("chroma_hor", chroma_hor_bs)):
                if arr is not None:
                    a = np.ascontiguousarray(arr, np.uint8)
                    keep.append(a)
                    setattr(bs, nm, a.ctypes.data)
                    setattr(bs, "n_" + nm, a.size)
        q = _lib.Qp()
        q.qp, q.ctu_log2 = int(qp), ctu_log2
        if qp_map is not None:
            m = np.ascontiguousarray(qp_map, np.uint8)
            keep.append(m)
            q.map, q.map_stride = m.ctypes.data, m.shape[1]
        t, k2 = _tables(tc_table, beta_table)
        tm = _lib.Timing()
        rc = _lib.lib().hevc_deblocking_filter(self.handle, C.byref(fr), None if bs is None else C.byref(bs),
                                               C.byref(q), None if t is None else C.byref(t),
                                               C.byref(tm) if want_timing else None)
        _chk(rc, self.handle)
        return {"exec_s": tm.exec_s, "total_s": tm.total_s, "copy_s": tm.copy_s, "pipelined_s": tm.pipelined_s}

    def alloc_probed(self, planes, qp, candidates=6, *, tc_table=None, beta_table=None):
        """hevcdbk_device_malloc_probed: a destination pool for launches like `planes` (a DevicePlanes; its dst is ignored), the
        fastest of `candidates` allocations by the filter's own time on each.  Returns (DeviceBuffer, best_ms, worst_ms)."""
        t, _k = _tables(tc_table, beta_table)
        p, best, worst = C.c_void_p(), C.c_float(), C.c_float()
        _chk(_lib.lib().hevcdbk_device_malloc_probed(self.handle, C.byref(planes), int(qp), None if t is None else C.byref(t),
                                                     int(candidates), C.byref(p), C.byref(best), C.byref(worst)), self.handle)
        nbytes = planes.frame_stride * (planes.n_frames - 1) + planes.pitch * planes.plane_h
        return DeviceBuffer.adopt(self, p.value, nbytes), best.value, worst.value

    # -- host side of filter_frame on large pageable frames --------------------------------------
    def set_host_threads(self, n):
        """Threads that copy between the caller's pageable planes and the page-locked ring during a large-frame
        filter_frame call, the calling thread included (0 = the default, 4)."""
        _chk(_lib.lib().hevcdbk_set_host_threads(self.handle, int(n)), self.handle)

    def host_threads(self):
        return int(_lib.lib().hevcdbk_get_host_threads(self.handle))

    def host_register(self, arr):
        """Page-locks the numpy array's memory in place (hevcdbk_host_register): planes inside it are DMA'd where they
        lie by the host-frame operators.  Unregister before the array is freed."""
        _chk(_lib.lib().hevcdbk_host_register(self.handle, arr.ctypes.data, arr.nbytes), self.handle)

    def host_unregister(self, arr):
        _chk(_lib.lib().hevcdbk_host_unregister(self.handle, arr.ctypes.data), self.handle)

    def last_frame_trace(self):
        """Per-strip record of the last large-frame filter_frame call (hevcdbk_last_frame_trace), a list of dicts."""
        n = C.c_uint(0)
        _chk(_lib.lib().hevcdbk_last_frame_trace(self.handle, None, 0, C.byref(n)), self.handle)
        if n.value == 0:
            return []
        buf = (_lib.StripTrace * n.value)()
        _chk(_lib.lib().hevcdbk_last_frame_trace(self.handle, buf, n.value, C.byref(n)), self.handle)
        return [{f: getattr(t, f) for f, _ in _lib.StripTrace._fields_} for t in buf]

    # -- streaming operator on a sequence of host frames -----------------------------------------
    def pinned_array(self, shape, dtype=np.uint8):
        """numpy array in page-locked host memory (hevcdbk_host_malloc_pinned): planes allocated this way are
        DMA'd in place by filter_sequence / filter_frame, with no staging copy.  Free with free_pinned()."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _chk(_lib.lib().hevcdbk_host_malloc_pinned(self.handle, n, C.byref(p)), self.handle)
        buf = (C.c_uint8 * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def free_pinned(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            _chk(_lib.lib().hevcdbk_host_free_pinned(self.handle, p), self.handle)

    def filter_sequence(self, frames, *, qp, bit_depth=8, vert_bs=None, hor_bs=None):
        """frames: list of (y,) or (y, u, v) tuples of writable 2-D numpy planes of one geometry; filtered in
        place through the 3-deep H2D || kernel || D2H pipeline.  Returns the wall time of the sequence (s)."""
        arr = (_lib.Frame * len(frames))()
        for i, pl in enumerate(frames):
            y = pl[0]
            arr[i].height, arr[i].width = y.shape
            arr[i].bit_depth, arr[i].sample_bytes = bit_depth, y.dtype.itemsize
            for k, p in enumerate(pl):
                assert p.flags.writeable and p.strides[1] == p.itemsize
                arr[i].plane[k] = p.ctypes.data
                arr[i].pitch[k] = p.strides[0]
        bs = None
        keep = []
        if vert_bs is not None:
            bs = _lib.Bs()
            for nm, a in (("vert", vert_bs), ("hor", hor_bs)):
                a = np.ascontiguousarray(a, np.uint8)
                keep.append(a)
                setattr(bs, nm, a.ctypes.data)
                setattr(bs, "n_" + nm, a.size)
        q = _lib.Qp()
        q.qp, q.ctu_log2 = int(qp), 6
        tm = _lib.Timing()
        rc = _lib.lib().hevc_deblocking_filter_sequence(self.handle, arr, len(frames), None if bs is None else C.byref(bs),
                                                        C.byref(q), None, C.byref(tm))
        _chk(rc, self.handle)
        return tm.pipelined_s

    def filter_yuv_file(self, in_name, out_name, width, height, qp, *, vert_bs=None, hor_bs=None,
                        tc_table=None, beta_table=None):
        """Multi-frame planar 8-bit 4:2:0 file -> file (hevcdbk_filter_yuv_file): every frame gets what the
        reference's ReadYuvFrame -> [SetBoundaryStrenght] -> DeblockingFilter -> Save gives a one-frame file
        (cpu.h:35-132, 995-1018).  Returns (n_frames, wall seconds including file I/O)."""
        bs = None
        keep = []
        if vert_bs is not None:
            bs = _lib.Bs()
            for nm, arr in (("vert", vert_bs), ("hor", hor_bs)):
                a = np.ascontiguousarray(arr, np.uint8)
                keep.append(a)
                setattr(bs, nm, a.ctypes.data)
                setattr(bs, "n_" + nm, a.size)
        t, k2 = _tables(tc_table, beta_table)
        tm = _lib.Timing()
        n = C.c_uint(0)
        rc = _lib.lib().hevcdbk_filter_yuv_file(self.handle, os.fsencode(in_name), os.fsencode(out_name), width, height,
                                                int(qp), None if bs is None else C.byref(bs),
                                                None if t is None else C.byref(t), C.byref(n), C.byref(tm))
        _chk(rc, self.handle)
        return n.value, tm.pipelined_s

    # -- device-resident operator ---------------------------------------------------------------
    def filter_device(self, planes, qp, *, tc_table=None, beta_table=None, variant=KERNEL_AUTO):
        t, _k = _tables(tc_table, beta_table)
        rc = _lib.lib().hevc_deblocking_filter_device(self.handle, C.byref(planes), int(qp),
                                                      None if t is None else C.byref(t), variant, None)
        _chk(rc, self.handle)

    # -- spec-exact mode (H.265 clause 8.7.2) ------------------------------------------------------
    def filter_frame_h265(self, y, u=None, v=None, *, qp, bit_depth=8, units=None, vert_bs4=None, hor_bs4=None,
                          qp_map=None, unit_log2=3, tc_offset_div2=0, beta_offset_div2=0, cb_qp_offset=0, cr_qp_offset=0):
        """hevc_deblocking_filter_h265 on host planes, in place.  Either `units` = (flags, mv0, mv1, ref0, ref1) per 4x4
        luma unit (bS derived on the GPU, 8.7.2.4) or the 4-sample-granular luma bS arrays."""
        fr = _lib.Frame()
        fr.height, fr.width = y.shape
        fr.bit_depth, fr.sample_bytes = bit_depth, y.dtype.itemsize
        for i, p in enumerate([y] + ([u, v] if u is not None else [])):
            assert p.flags.writeable and p.strides[1] == p.itemsize
            fr.plane[i] = p.ctypes.data
            fr.pitch[i] = p.strides[0]
        keep, un, bs = [], None, None
        if units is not None:
            un = _lib.H265Units()
            for nm, arr, dt in zip(("flags", "mv0", "mv1", "ref0", "ref1"), units, (np.uint16, np.int16, np.int16, np.int32, np.int32)):
                a = np.ascontiguousarray(arr, dt)
                keep.append(a)
                setattr(un, nm, a.ctypes.data)
        if vert_bs4 is not None:
            bs = _lib.Bs()
            for nm, arr in (("vert", vert_bs4), ("hor", hor_bs4)):
                a = np.ascontiguousarray(arr, np.uint8)
                keep.append(a)
                setattr(bs, nm, a.ctypes.data)
                setattr(bs, "n_" + nm, a.size)
        q = _lib.Qp()
        q.qp, q.ctu_log2 = int(qp), unit_log2
        if qp_map is not None:
            m = np.ascontiguousarray(qp_map, np.uint8)
            keep.append(m)
            q.map, q.map_stride = m.ctypes.data, m.shape[1]
        prm = _lib.H265Params(tc_offset_div2, beta_offset_div2, cb_qp_offset, cr_qp_offset)
        tm = _lib.Timing()
        rc = _lib.lib().hevc_deblocking_filter_h265(self.handle, C.byref(fr), None if un is None else C.byref(un),
                                                    None if bs is None else C.byref(bs), C.byref(q), C.byref(prm), C.byref(tm))
        _chk(rc, self.handle)
        return {"exec_s": tm.exec_s, "total_s": tm.total_s, "copy_s": tm.copy_s, "pipelined_s": tm.pipelined_s}

    def filter_device_h265(self, planes, qp, *, c_idx=0, tc_offset_div2=0, beta_offset_div2=0, cb_qp_offset=0, cr_qp_offset=0,
                           variant=KERNEL_AUTO):
        prm = _lib.H265Params(tc_offset_div2, beta_offset_div2, cb_qp_offset, cr_qp_offset)
        _chk(_lib.lib().hevc_deblocking_filter_h265_device(self.handle, C.byref(planes), c_idx, int(qp), C.byref(prm),
                                                           variant, None), self.handle)

    def derive_bs_h265(self, units, w, h, *, chroma=True):
        """8.7.2.4 on the GPU from host arrays; returns (vert, hor[, chroma_vert, chroma_hor]) as host arrays."""
        arrs = [np.ascontiguousarray(a, dt) for a, dt in zip(units, (np.uint16, np.int16, np.int16, np.int32, np.int32))]
        bufs = [self.alloc(a.nbytes) for a in arrs]
        for b, a in zip(bufs, arrs):
            b.upload(a)
        L = _lib.lib()
        sizes = [L.hevcdbk_h265_num_vert_bs(w, h), L.hevcdbk_h265_num_hor_bs(w, h)]
        if chroma:
            sizes += [L.hevcdbk_h265_num_vert_bs(w // 2, h // 2), L.hevcdbk_h265_num_hor_bs(w // 2, h // 2)]
        outs = [self.alloc(max(n, 1)) for n in sizes]
        un = _lib.H265Units(*[b.ptr for b in bufs])
        rc = L.hevcdbk_h265_derive_bs_device(self.handle, C.byref(un), w, h, outs[0].ptr, outs[1].ptr,
                                             outs[2].ptr if chroma else None, outs[3].ptr if chroma else None, None)
        _chk(rc, self.handle)
        self.synchronize()
        res = [o.download(n) for o, n in zip(outs, sizes)]
        for b in bufs + outs:
            b.free()
        return res

    def sao_device(self, planes, params_ptr, params_stride, ctb_log2, *, params_frame_stride=0, keep_ptr=None, keep_stride=0,
                   keep_frame_stride=0):
        """hevc_sao_filter_device: H.265 8.7.3 on planes in HBM, src -> dst."""
        _chk(_lib.lib().hevc_sao_filter_device(self.handle, C.byref(planes), params_ptr, params_stride, params_frame_stride,
                                               ctb_log2, keep_ptr, keep_stride, keep_frame_stride, None), self.handle)

    def deblock_sao_device(self, planes, qp, params_ptr, params_stride, ctb_log2, *, params_frame_stride=0, keep_ptr=None,
                           keep_stride=0, keep_frame_stride=0, tc_table=None, beta_table=None, fused=_lib.FUSED_AUTO, stream=None):
        """hevc_deblock_sao_device: reference-exact deblocking followed by SAO, src -> dst, one kernel where it applies.
        stream: a hipStream_t handle of the caller (int), None = the context's compute stream."""
        t, _k = _tables(tc_table, beta_table)
        _chk(_lib.lib().hevc_deblock_sao_device(self.handle, C.byref(planes), int(qp), None if t is None else C.byref(t), params_ptr,
                                                params_stride, params_frame_stride, ctb_log2, keep_ptr, keep_stride, keep_frame_stride,
                                                fused, stream), self.handle)

    def deblock_sao_h265_device(self, planes, qp, params_ptr, params_stride, ctb_log2, *, c_idx=0, tc_offset_div2=0,
                                beta_offset_div2=0, cb_qp_offset=0, cr_qp_offset=0, params_frame_stride=0, keep_ptr=None,
                                keep_stride=0, keep_frame_stride=0, fused=_lib.FUSED_AUTO):
        """hevc_deblock_sao_h265_device: spec-exact deblocking (8.7.2) followed by SAO (8.7.3), src -> dst."""
        prm = _lib.H265Params(tc_offset_div2, beta_offset_div2, cb_qp_offset, cr_qp_offset)
        _chk(_lib.lib().hevc_deblock_sao_h265_device(self.handle, C.byref(planes), c_idx, int(qp), C.byref(prm), params_ptr, params_stride,
                                                     params_frame_stride, ctb_log2, keep_ptr, keep_stride, keep_frame_stride, fused,
                                                     None), self.handle)

    def deblock_sao_device_planes(self, planes_list, qp, sao_list, *, h265=None, fused=_lib.FUSED_AUTO, tc_table=None, beta_table=None):
        """hevc_deblock_sao_device_planes / hevc_deblock_sao_h265_device_planes: deblocking + SAO of Y, U, V of a batch in one
        call (one launch where the fused kernel takes every plane).  sao_list[i] = (params_ptr, params_stride, ctb_log2) or a
        dict with the optional params_frame_stride / keep / keep_stride / keep_frame_stride; h265 = None (reference-exact
        deblocking) or a dict of tc_offset_div2, beta_offset_div2, cb_qp_offset, cr_qp_offset (spec-exact)."""
        arr = (_lib.DevicePlanes * len(planes_list))(*planes_list)
        sp = (_lib.SaoPlane * len(sao_list))()
        for i, so in enumerate(sao_list):
            d = so if isinstance(so, dict) else {"params": so[0], "params_stride": so[1], "ctb_log2": so[2]}
            sp[i].params, sp[i].params_stride, sp[i].ctb_log2 = d["params"], d["params_stride"], d["ctb_log2"]
            sp[i].params_frame_stride = d.get("params_frame_stride", 0)
            sp[i].keep, sp[i].keep_stride, sp[i].keep_frame_stride = d.get("keep"), d.get("keep_stride", 0), d.get("keep_frame_stride", 0)
        if h265 is None:
            t, _k = _tables(tc_table, beta_table)
            rc = _lib.lib().hevc_deblock_sao_device_planes(self.handle, arr, len(planes_list), int(qp), None if t is None else C.byref(t),
                                                           sp, fused, None)
        else:
            prm = _lib.H265Params(h265.get("tc_offset_div2", 0), h265.get("beta_offset_div2", 0), h265.get("cb_qp_offset", 0),
                                  h265.get("cr_qp_offset", 0))
            rc = _lib.lib().hevc_deblock_sao_h265_device_planes(self.handle, arr, len(planes_list), int(qp), C.byref(prm), sp, fused, None)
        _chk(rc, self.handle)

    def filter_device_planes(self, planes_list, qp, *, tc_table=None, beta_table=None, variant=KERNEL_AUTO):
        """hevc_deblocking_filter_device_planes: Y, U, V of a batch in one call (one fused launch where that applies)."""
        arr = (_lib.DevicePlanes * len(planes_list))(*planes_list)
        t, _k = _tables(tc_table, beta_table)
        _chk(_lib.lib().hevc_deblocking_filter_device_planes(self.handle, arr, len(planes_list), int(qp),
                                                             None if t is None else C.byref(t), variant, None), self.handle)

    def run_timed(self, planes_list, qp, steps, *, variant=KERNEL_AUTO, tc_table=None, beta_table=None):
        """`steps` back-to-back launches of every plane in planes_list; per-step kernel ms (HIP events)."""
        arr = (_lib.DevicePlanes * len(planes_list))(*planes_list)
        ms = (C.c_float * steps)()
        t, _k = _tables(tc_table, beta_table)
        rc = _lib.lib().hevcdbk_device_run_timed(self.handle, arr, len(planes_list), int(qp),
                                                 None if t is None else C.byref(t), variant, steps, ms)
        _chk(rc, self.handle)
        return np.array(ms, np.float64)

    def replay(self, planes_list, qp, steps, *, warmup=0, settle_min_ms=150.0, settle_max_ms=2000.0, settle_tolerance=0.005,
               settle_window=32, variant=KERNEL_AUTO, tc_table=None, beta_table=None):
        """hevcdbk_device_replay: ONE uninterrupted stream of [settle by time][warmup][steps timed] launches, one
        synchronisation at the end.  Returns (per-launch kernel ms of the timed launches, dict of the replay's outputs)."""
        arr = (_lib.DevicePlanes * len(planes_list))(*planes_list)
        ms = (C.c_float * max(steps, 1))()
        t, _k = _tables(tc_table, beta_table)
        r = _lib.Replay(settle_min_ms=settle_min_ms, settle_max_ms=settle_max_ms, settle_tolerance=settle_tolerance,
                        settle_window=settle_window, warmup=warmup, steps=steps)
        rc = _lib.lib().hevcdbk_device_replay(self.handle, arr, len(planes_list), int(qp),
                                              None if t is None else C.byref(t), variant, C.byref(r), ms)
        _chk(rc, self.handle)
        info = {k: getattr(r, k) for k in ("settle_launches", "settled", "settle_ms", "settle_tail_mean_ms", "t_begin", "t_end",
                                           "wall_ms", "span_ms")}
        return np.array(ms[:steps], np.float64), info

    def pci_bus_id(self):
        """'0000:0a:00.0' of this context's device (hevcdbk_device_pci_bus_id), lower case"""
        buf = C.create_string_buffer(64)
        _chk(_lib.lib().hevcdbk_device_pci_bus_id(self.handle, buf, 64), self.handle)
        return buf.value.decode().lower()


class DeviceBatch:
    """n_frames planes of identical geometry resident in HBM (src and dst), plus their bS arrays.
    This is the layout bench.py times: frame f at base + f*frame_stride, tight pitch."""

    def __init__(self, ctx, plane_w, plane_h, n_frames, *, bit_depth=8, sample_bytes=None, is_chroma=False,
                 in_place=False, per_frame_bs=True, pitch=None, storage=None):
        """storage = (src_buffer, dst_buffer) of another batch: lay this batch out in that device memory instead of
        allocating (a decoder's frame pool serving another geometry); free() then leaves those buffers alone."""
        self.ctx = ctx
        self.w, self.h, self.n = plane_w, plane_h, n_frames
        self.bit_depth = bit_depth
        self.sb = sample_bytes or (1 if bit_depth == 8 else 2)
        self.dtype = np.uint8 if self.sb == 1 else np.uint16
        self.is_chroma = is_chroma
        self.pitch = plane_w * self.sb if pitch is None else int(pitch)  # bytes; > width*sb leaves row padding
        assert self.pitch >= plane_w * self.sb and self.pitch % self.sb == 0
        self.frame_bytes = self.pitch * plane_h
        self._borrowed = storage is not None
        if storage is not None:
            self.src, self.dst = storage
            assert self.src.nbytes >= self.frame_bytes * n_frames and self.dst.nbytes >= self.frame_bytes * n_frames
        else:
            self.src = ctx.alloc(self.frame_bytes * n_frames)
            self.dst = self.src if in_place else ctx.alloc(self.frame_bytes * n_frames)
        self.nv, self.nh = num_vert_bs(plane_w, plane_h), num_hor_bs(plane_w, plane_h)
        self.per_frame_bs = per_frame_bs
        nb = n_frames if per_frame_bs else 1
        self.vert = ctx.alloc(self.nv * nb)
        self.hor = ctx.alloc(self.nh * nb)
        dv, dh = default_bs(plane_w, plane_h)
        self.vert.upload(np.tile(dv, nb))
        self.hor.upload(np.tile(dh, nb))
        self.qp_map = None
        self.map_stride = 0
        self.ctu_log2 = 6

    def _pitched(self, a, fill=0):
        """(.., h, w) samples -> (.., h, pitch/sb) with `fill` in the row padding"""
        ps = self.pitch // self.sb
        if ps == self.w:
            return np.ascontiguousarray(a, self.dtype)
        out = np.full(a.shape[:-1] + (ps,), fill, self.dtype)
        out[..., : self.w] = a
        return out

    def upload_frame(self, f, plane, fill=0):
        a = np.asarray(plane, self.dtype)
        assert a.shape == (self.h, self.w)
        self.src.upload(self._pitched(a, fill), f * self.frame_bytes)

    def upload_all(self, frames, fill=0):
        a = np.asarray(frames, self.dtype)
        assert a.shape == (self.n, self.h, self.w)
        self.src.upload(self._pitched(a, fill))

    def set_bs(self, f, vert, hor):
        assert self.per_frame_bs or f == 0
        self.vert.upload(np.ascontiguousarray(vert, np.uint8), f * self.nv)
        self.hor.upload(np.ascontiguousarray(hor, np.uint8), f * self.nh)

    def set_qp_map(self, qmap, ctu_log2=6):
        m = np.ascontiguousarray(qmap, np.uint8)
        self.qp_map = self.ctx.alloc(m.nbytes)
        self.qp_map.upload(m)
        self.map_stride, self.ctu_log2 = m.shape[1], ctu_log2

    def planes(self):
        p = _lib.DevicePlanes()
        p.src, p.dst = self.src.ptr, self.dst.ptr
        p.pitch, p.frame_stride, p.n_frames = self.pitch, self.frame_bytes, self.n
        p.plane_w, p.plane_h = self.w, self.h
        p.bit_depth, p.sample_bytes, p.is_chroma = self.bit_depth, self.sb, int(self.is_chroma)
        p.vert_bs, p.hor_bs = self.vert.ptr, self.hor.ptr
        p.vert_bs_stride = self.nv if self.per_frame_bs else 0
        p.hor_bs_stride = self.nh if self.per_frame_bs else 0
        if self.qp_map is not None:
            p.qp_map, p.qp_map_stride, p.ctu_log2 = self.qp_map.ptr, self.map_stride, self.ctu_log2
        return p

    def download_frame(self, f, which="dst", with_padding=False):
        buf = self.dst if which == "dst" else self.src
        a = buf.download(self.frame_bytes, f * self.frame_bytes, self.dtype).reshape(self.h, self.pitch // self.sb)
        return a if with_padding else a[:, : self.w]

    def free(self):
        own = (self.vert, self.hor, self.qp_map) if self._borrowed else (self.src, self.dst, self.vert, self.hor, self.qp_map)
        for b in own:
            if b is not None and b.ptr:
                b.free()


class ReadYuvFrame:
    """Same call surface as the reference class (cpu.h:33), executed by the HIP library.

    ctor(file_name, width, height, Qp=20)            cpu.h:35
    SetBoundaryStrenght(vert_bs, hor_bs)              cpu.h:120  (luma only)
    DeblockingFilter()                                cpu.h:134  -> hevc_deblocking_filter on the GPU
    Save(output_file_name)                            cpu.h:995
    Errors are raised as DeblockError carrying the reference's message text.
    """

    def __init__(self, file_name, width, height, Qp=20, ctx=None):
        with open(file_name, "rb") as fh:
            buf = fh.read()
        if len(buf) != 3 * width * height // 2:            # cpu.h:43-45
            raise DeblockError(_lib.ERR_FILE_SIZE)
        if width % 8 != 0 or height % 8 != 0:               # cpu.h:46-48
            raise DeblockError(_lib.ERR_DIMENSIONS)
        a = np.frombuffer(buf, np.uint8)
        ysz, csz = width * height, width * height // 4
        self.y = a[:ysz].reshape(height, width).copy()
        self.u = a[ysz:ysz + csz].reshape(height // 2, width // 2).copy()
        self.v = a[ysz + csz:].reshape(height // 2, width // 2).copy()
        self.width, self.height, self.Qp = width, height, Qp
        self._vert = self._hor = None
        self._own_ctx = ctx is None
        self._ctx = ctx or Context(0)
        self.timing = None

    def SetBoundaryStrenght(self, vert_bs, hor_bs):
        vert_bs = np.ascontiguousarray(vert_bs, np.uint8)
        hor_bs = np.ascontiguousarray(hor_bs, np.uint8)
        if vert_bs.size != num_vert_bs(self.width, self.height) or hor_bs.size != num_hor_bs(self.width, self.height):
            raise DeblockError(_lib.ERR_BS_SIZE)              # cpu.h:122-123
        self._vert, self._hor = vert_bs.copy(), hor_bs.copy()

    def DeblockingFilter(self, num_threads=1):
        # num_threads is the reference's OpenMP knob (cpu.h:134-135); meaningless on the GPU, accepted and ignored
        self.timing = self._ctx.filter_frame(self.y, self.u, self.v, qp=self.Qp, vert_bs=self._vert, hor_bs=self._hor)

    def Save(self, output_file_name):
        with open(output_file_name, "wb") as fh:
            fh.write(self.tobytes())

    def tobytes(self):
        return self.y.tobytes() + self.u.tobytes() + self.v.tobytes()

    def close(self):
        if self._own_ctx and self._ctx:
            self._ctx.close()
            self._ctx = None
