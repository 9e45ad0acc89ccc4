"""ctypes bindings for the CPU checker (oracle/liboracle.so) and, when present, for the
reference's own CPU implementation (oracle/_ref/libref_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package gpu_video_codec_amd.

Reference citations: "cpu.h" = /root/reference/hevc_deblocking_filter/hevc_deblocking_filter_cpu.h.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libref_oracle.so")
REF_HEADER = "/root/reference/hevc_deblocking_filter/hevc_deblocking_filter_cpu.h"

PLANE_Y, PLANE_U, PLANE_V = 1, 2, 4
ERR_FILE_SIZE, ERR_DIMENSIONS, ERR_BS_SIZE = -1, -2, -3


class _Qp(C.Structure):
    _fields_ = [("qp", C.c_uint), ("map", C.c_void_p), ("map_stride", C.c_uint), ("ctu_log2", C.c_uint)]


class _Tables(C.Structure):
    _fields_ = [("tc", C.c_void_p), ("beta", C.c_void_p)]


def build(ref=True):
    """(Re)build liboracle.so and, when the reference is mounted, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.exists(REF_HEADER):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        _lib = C.CDLL(LIB_PATH)
        _lib.dbko_num_vert_bs.restype = C.c_size_t
        _lib.dbko_num_hor_bs.restype = C.c_size_t
        _lib.dbko_default_bs.restype = None
    return _lib


def have_ref():
    return os.path.exists(REF_LIB_PATH)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_LIB_PATH)
        _ref.ref_frame_destroy.restype = None
    return _ref


def num_vert_bs(w, h):
    return int(lib().dbko_num_vert_bs(C.c_uint(w), C.c_uint(h)))


def num_hor_bs(w, h):
    return int(lib().dbko_num_hor_bs(C.c_uint(w), C.c_uint(h)))


def default_bs(w, h):
    """cpu.h:92-99: the default 'all intra' pattern for a w x h plane."""
    v = np.empty(num_vert_bs(w, h), np.uint8)
    hh = np.empty(num_hor_bs(w, h), np.uint8)
    lib().dbko_default_bs(C.c_uint(w), C.c_uint(h), v.ctypes.data_as(C.c_void_p), hh.ctypes.data_as(C.c_void_p))
    return v, hh


def tables():
    tc = np.ctypeslib.as_array((C.c_uint * 52).in_dll(lib(), "dbko_tc_table")).copy()
    beta = np.ctypeslib.as_array((C.c_uint * 52).in_dll(lib(), "dbko_beta_table")).copy()
    return tc, beta


def lcg_bs(w, h, seed):
    """Seeded bS in {0,1,2}: s = s*1664525 + 1013904223 (mod 2^32), value (s>>16)%3, the vert
    array first then hor (identical to oracle/ref_harness.cpp main())."""
    nv, nh = num_vert_bs(w, h), num_hor_bs(w, h)
    out = np.empty(nv + nh, np.uint8)
    s = seed & 0xFFFFFFFF
    for i in range(nv + nh):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = (s >> 16) % 3
    return out[:nv].copy(), out[nv:].copy()


def _qp_struct(qp, qp_map, ctu_log2):
    q = _Qp()
    keep = None
    if qp_map is not None:
        keep = np.ascontiguousarray(qp_map, np.uint8)
        q.qp, q.map, q.map_stride, q.ctu_log2 = 0, keep.ctypes.data, keep.shape[1], ctu_log2
    else:
        q.qp, q.map, q.map_stride, q.ctu_log2 = int(qp), None, 0, ctu_log2
    return q, keep


def _tables_struct(tc, beta):
    t = _Tables()
    keep = []
    for name, arr in (("tc", tc), ("beta", beta)):
        if arr is not None:
            a = np.ascontiguousarray(arr, np.uint32)
            assert a.size == 52
            keep.append(a)
            setattr(t, name, a.ctypes.data)
    return t, keep


def _dtype(bit_depth, sample_bytes):
    if sample_bytes is None:
        sample_bytes = 1 if bit_depth == 8 else 2
    return (np.uint8 if sample_bytes == 1 else np.uint16), sample_bytes


def filter_plane(plane, qp, *, is_chroma=False, bit_depth=8, sample_bytes=None, vert_bs=None, hor_bs=None,
                 qp_map=None, ctu_log2=6, tc_table=None, beta_table=None, threads=1):
    """Filter one un-padded plane (2-D numpy array, uint8 or uint16); returns a new array."""
    dt, sample_bytes = _dtype(bit_depth, sample_bytes)
    out = np.ascontiguousarray(plane, dt).copy()
    h, w = out.shape
    q, _k1 = _qp_struct(qp, qp_map, ctu_log2)
    t, _k2 = _tables_struct(tc_table, beta_table)
    vb = None if vert_bs is None else np.ascontiguousarray(vert_bs, np.uint8)
    hb = None if hor_bs is None else np.ascontiguousarray(hor_bs, np.uint8)
    if vb is not None and vb.size != num_vert_bs(w, h) or hb is not None and hb.size != num_hor_bs(w, h):
        raise ValueError("Incorrect size of input boundary strenght array")
    rc = lib().dbko_filter_plane(
        out.ctypes.data_as(C.c_void_p), C.c_uint(w), C.c_uint(h), C.c_size_t(out.strides[0]),
        C.c_uint(bit_depth), C.c_uint(sample_bytes), C.c_int(1 if is_chroma else 0),
        None if vb is None else vb.ctypes.data_as(C.c_void_p),
        None if hb is None else hb.ctypes.data_as(C.c_void_p),
        C.byref(q), C.byref(t), C.c_uint(threads))
    if rc:
        raise RuntimeError("dbko_filter_plane failed: %d" % rc)
    return out


class Frame:
    """Mirror of class ReadYuvFrame (cpu.h:33-132, 995-1018) on top of the restatement."""

    def __init__(self, y, u=None, v=None, bit_depth=8, sample_bytes=None):
        dt, sample_bytes = _dtype(bit_depth, sample_bytes)
        self._y = np.ascontiguousarray(y, dt)
        self._u = None if u is None else np.ascontiguousarray(u, dt)
        self._v = None if v is None else np.ascontiguousarray(v, dt)
        self.height, self.width = self._y.shape
        self.bit_depth = bit_depth
        self._dt = dt
        self._h = C.c_void_p()
        rc = lib().dbko_frame_create(
            C.byref(self._h), C.c_uint(self.width), C.c_uint(self.height), C.c_uint(bit_depth),
            C.c_uint(sample_bytes), self._y.ctypes.data_as(C.c_void_p), C.c_size_t(self._y.strides[0]),
            None if self._u is None else self._u.ctypes.data_as(C.c_void_p),
            C.c_size_t(0 if self._u is None else self._u.strides[0]),
            None if self._v is None else self._v.ctypes.data_as(C.c_void_p),
            C.c_size_t(0 if self._v is None else self._v.strides[0]))
        if rc:
            raise RuntimeError("dbko_frame_create failed: %d" % rc)

    def set_boundary_strength(self, vert_bs, hor_bs):
        vb = np.ascontiguousarray(vert_bs, np.uint8)
        hb = np.ascontiguousarray(hor_bs, np.uint8)
        return lib().dbko_frame_set_boundary_strength(
            self._h, vb.ctypes.data_as(C.c_void_p), C.c_size_t(vb.size),
            hb.ctypes.data_as(C.c_void_p), C.c_size_t(hb.size))

    def filter(self, qp, planes=PLANE_Y | PLANE_U | PLANE_V, threads=1, qp_map=None, ctu_log2=6,
               tc_table=None, beta_table=None):
        q, _k1 = _qp_struct(qp, qp_map, ctu_log2)
        t, _k2 = _tables_struct(tc_table, beta_table)
        rc = lib().dbko_frame_filter(self._h, C.byref(q), C.byref(t), C.c_uint(planes), C.c_uint(threads))
        if rc:
            raise RuntimeError("dbko_frame_filter failed: %d" % rc)

    def save(self):
        y = np.empty((self.height, self.width), self._dt)
        u = v = None
        if self._u is not None:
            u = np.empty((self.height // 2, self.width // 2), self._dt)
            v = np.empty_like(u)
        lib().dbko_frame_save(
            self._h, y.ctypes.data_as(C.c_void_p), C.c_size_t(y.strides[0]),
            None if u is None else u.ctypes.data_as(C.c_void_p), C.c_size_t(0 if u is None else u.strides[0]),
            None if v is None else v.ctypes.data_as(C.c_void_p), C.c_size_t(0 if v is None else v.strides[0]))
        return y, u, v

    def close(self):
        if self._h:
            lib().dbko_frame_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def split_yuv420(buf, w, h):
    """Planar 4:2:0 file bytes -> (Y, U, V) arrays (cpu.h:66-82 read order)."""
    a = np.frombuffer(buf, np.uint8)
    if a.size != 3 * w * h // 2:
        raise ValueError("Incorrect file size")
    ysz, csz = w * h, w * h // 4
    return (a[:ysz].reshape(h, w), a[ysz:ysz + csz].reshape(h // 2, w // 2),
            a[ysz + csz:].reshape(h // 2, w // 2))


def join_yuv420(y, u, v):
    return np.concatenate([y.ravel(), u.ravel(), v.ravel()]).tobytes()


def filter_yuv420(buf, w, h, qp, vert_bs=None, hor_bs=None, threads=1):
    """ExecuteCpu-equivalent on bytes: ctor -> [SetBoundaryStrenght] -> DeblockingFilter -> Save."""
    y, u, v = split_yuv420(buf, w, h)
    f = Frame(y, u, v)
    if vert_bs is not None:
        rc = f.set_boundary_strength(vert_bs, hor_bs)
        if rc:
            raise RuntimeError("set_boundary_strength: %d" % rc)
    f.filter(qp, threads=threads)
    out = join_yuv420(*f.save())
    f.close()
    return out


# ------------------------------------------------------------------------------------------
# the reference itself (only when oracle/_ref was built in the container that has /root/reference)

class RefFrame:
    """The reference's ReadYuvFrame through oracle/_ref/libref_oracle.so (file based, 8-bit 4:2:0)."""

    def __init__(self, yuv_bytes, w, h, qp):
        self._tmp = tempfile.NamedTemporaryFile(suffix=".yuv", delete=False)
        self._tmp.write(yuv_bytes)
        self._tmp.close()
        self.w, self.h = w, h
        self._h = C.c_void_p()
        rc = ref().ref_frame_create(C.byref(self._h), self._tmp.name.encode(), C.c_uint(w), C.c_uint(h), C.c_uint(qp))
        if rc:
            os.unlink(self._tmp.name)
            raise RuntimeError("ref_frame_create failed: %d" % rc)

    def set_bs(self, vert_bs, hor_bs):
        vb = np.ascontiguousarray(vert_bs, np.uint8)
        hb = np.ascontiguousarray(hor_bs, np.uint8)
        return ref().ref_frame_set_bs(self._h, vb.ctypes.data_as(C.c_void_p), C.c_uint(vb.size),
                                      hb.ctypes.data_as(C.c_void_p), C.c_uint(hb.size))

    def filter(self, threads=1):
        return ref().ref_frame_filter(self._h, C.c_uint(threads))

    def save(self):
        out = self._tmp.name + ".out"
        ref().ref_frame_save(self._h, out.encode())
        with open(out, "rb") as fh:
            data = fh.read()
        os.unlink(out)
        return data

    def close(self):
        if self._h:
            ref().ref_frame_destroy(self._h)
            self._h = C.c_void_p()
        if os.path.exists(self._tmp.name):
            os.unlink(self._tmp.name)


def ref_filter_yuv420(buf, w, h, qp, vert_bs=None, hor_bs=None, threads=1):
    f = RefFrame(buf, w, h, qp)
    try:
        if vert_bs is not None:
            rc = f.set_bs(vert_bs, hor_bs)
            if rc:
                raise RuntimeError("ref set_bs: %d" % rc)
        f.filter(threads)
        return f.save()
    finally:
        f.close()
