"""ctypes bindings for oracle/h265_oracle.c -- the CPU restatement of H.265 clause 8.7.2 (spec-exact mode).

TEST INFRASTRUCTURE ONLY, and PARITY UNPINNED (see oracle/h265_oracle.h): imported by tests/ only.
"""
import ctypes as C

import numpy as np

from . import oracle as _o

BS_MASK, KEEP_P, KEEP_Q = 3, 4, 8
U_INTRA, U_CBF, U_TU_LEFT, U_TU_TOP, U_PU_LEFT, U_PU_TOP = 1, 2, 4, 8, 16, 32
U_KEEP, U_DBK_OFF, U_PRED_L0, U_PRED_L1, U_NOX_LEFT, U_NOX_TOP = 64, 128, 256, 512, 1024, 2048


class _Params(C.Structure):
    _fields_ = [("tc_offset_div2", C.c_int), ("beta_offset_div2", C.c_int), ("c_qp_offset", C.c_int)]


class _Units(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("mv0", C.c_void_p), ("mv1", C.c_void_p), ("ref0", C.c_void_p), ("ref1", C.c_void_p)]


def _lib():
    L = _o.lib()
    L.dbko_h265_num_vert_bs.restype = C.c_size_t
    L.dbko_h265_num_hor_bs.restype = C.c_size_t
    L.dbko_h265_chroma_bs.restype = None
    return L


def num_vert_bs(w, h):
    return (w // 8 + 1) * (h // 4)


def num_hor_bs(w, h):
    return (h // 8 + 1) * (w // 4)


def tables():
    L = _lib()
    beta = np.ctypeslib.as_array((C.c_uint8 * 52).in_dll(L, "dbko_h265_beta_table")).copy()
    tc = np.ctypeslib.as_array((C.c_uint8 * 54).in_dll(L, "dbko_h265_tc_table")).copy()
    return tc, beta


def filter_plane(plane, qp, vert_bs4, hor_bs4, *, c_idx=0, bit_depth=8, qp_map=None, unit_log2=3,
                 tc_offset_div2=0, beta_offset_div2=0, c_qp_offset=0):
    """Deblock one un-padded plane per H.265 8.7.2 (vertical edges of the picture, then horizontal); returns a copy."""
    out = np.ascontiguousarray(plane).copy()
    h, w = out.shape
    vb = np.ascontiguousarray(vert_bs4, np.uint8)
    hb = np.ascontiguousarray(hor_bs4, np.uint8)
    assert vb.size == num_vert_bs(w, h) and hb.size == num_hor_bs(w, h)
    m = None if qp_map is None else np.ascontiguousarray(qp_map, np.uint8)
    prm = _Params(tc_offset_div2, beta_offset_div2, c_qp_offset)
    rc = _lib().dbko_h265_filter_plane(
        out.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(out.strides[0]), bit_depth, out.itemsize, c_idx,
        vb.ctypes.data_as(C.c_void_p), hb.ctypes.data_as(C.c_void_p), int(qp),
        None if m is None else m.ctypes.data_as(C.c_void_p), 0 if m is None else m.shape[1], unit_log2, C.byref(prm))
    if rc:
        raise RuntimeError("dbko_h265_filter_plane: %d" % rc)
    return out


def derive_bs(flags, mv0, mv1, ref0, ref1, w, h):
    """8.7.2.4 on per-4x4-unit arrays (flags uint16 (H/4, W/4); mv int16 (H/4, W/4, 2); ref int32 (H/4, W/4))."""
    keep = [np.ascontiguousarray(flags, np.uint16), np.ascontiguousarray(mv0, np.int16), np.ascontiguousarray(mv1, np.int16),
            np.ascontiguousarray(ref0, np.int32), np.ascontiguousarray(ref1, np.int32)]
    u = _Units(*[a.ctypes.data for a in keep])
    vb = np.zeros(num_vert_bs(w, h), np.uint8)
    hb = np.zeros(num_hor_bs(w, h), np.uint8)
    rc = _lib().dbko_h265_derive_bs(C.byref(u), w, h, vb.ctypes.data_as(C.c_void_p), hb.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("dbko_h265_derive_bs: %d" % rc)
    return vb, hb


def chroma_bs(vert_bs4, hor_bs4, w, h):
    vb = np.ascontiguousarray(vert_bs4, np.uint8)
    hb = np.ascontiguousarray(hor_bs4, np.uint8)
    cvb = np.zeros(num_vert_bs(w // 2, h // 2), np.uint8)
    chb = np.zeros(num_hor_bs(w // 2, h // 2), np.uint8)
    _lib().dbko_h265_chroma_bs(vb.ctypes.data_as(C.c_void_p), hb.ctypes.data_as(C.c_void_p), w, h,
                               cvb.ctypes.data_as(C.c_void_p), chb.ctypes.data_as(C.c_void_p))
    return cvb, chb


def random_units(w, h, seed, *, p_intra=0.3, grid=8):
    """Seeded synthetic coding structure: a random quadtree-ish partition into 8..32 blocks with per-block
    prediction data, so that edges, cbf, motion and the special flags all occur.  Integer-only generator."""
    rng = np.random.RandomState(seed)
    uw, uh = w // 4, h // 4
    flags = np.zeros((uh, uw), np.uint16)
    mv0 = np.zeros((uh, uw, 2), np.int16)
    mv1 = np.zeros((uh, uw, 2), np.int16)
    ref0 = np.zeros((uh, uw), np.int32)
    ref1 = np.zeros((uh, uw), np.int32)
    for by in range(0, uh, 8):
        for bx in range(0, uw, 8):
            size = int(rng.choice([2, 4, 8]))  # block size in 4x4 units: 8, 16, 32 luma samples
            for y0 in range(by, min(by + 8, uh), size):
                for x0 in range(bx, min(bx + 8, uw), size):
                    y1, x1 = min(y0 + size, uh), min(x0 + size, uw)
                    f = 0
                    r = rng.randint(0, 100)
                    if r < int(p_intra * 100):
                        f |= U_INTRA
                    else:
                        kind = rng.randint(0, 3)
                        if kind in (0, 2):
                            f |= U_PRED_L0
                        if kind in (1, 2):
                            f |= U_PRED_L1
                    if rng.randint(0, 3) == 0:
                        f |= U_CBF
                    if rng.randint(0, 12) == 0:
                        f |= U_KEEP
                    if rng.randint(0, 25) == 0:
                        f |= U_DBK_OFF
                    flags[y0:y1, x0:x1] = f
                    mv0[y0:y1, x0:x1] = rng.randint(-6, 7, 2)
                    mv1[y0:y1, x0:x1] = rng.randint(-6, 7, 2)
                    ref0[y0:y1, x0:x1] = rng.randint(0, 3)
                    ref1[y0:y1, x0:x1] = rng.randint(0, 3)
                    # the block's left / top borders are both transform and prediction edges; now and then only one
                    e = rng.randint(0, 8)
                    left = U_TU_LEFT | U_PU_LEFT if e > 1 else (U_TU_LEFT if e == 0 else U_PU_LEFT)
                    top = U_TU_TOP | U_PU_TOP if e > 1 else (U_TU_TOP if e == 0 else U_PU_TOP)
                    flags[y0:y1, x0] |= left
                    flags[y0, x0:x1] |= top
                    if rng.randint(0, 30) == 0:
                        flags[y0:y1, x0] |= U_NOX_LEFT
                    if rng.randint(0, 30) == 0:
                        flags[y0, x0:x1] |= U_NOX_TOP
                    # an inner 4-sample transform split (off the 8x8 grid: must never produce an edge)
                    if size >= 2 and rng.randint(0, 2) == 0:
                        flags[y0:y1, x0 + 1] |= U_TU_LEFT
    return flags, mv0, mv1, ref0, ref1


# ---- sample adaptive offset, clause 8.7.3 ---------------------------------------------------------------------------

SAO_CTB_DTYPE = np.dtype([("type", "u1"), ("cls", "u1"), ("offset", "i1", (4,))])


def sao_plane(plane, params, ctb_log2, *, bit_depth=8, keep=None):
    """SAO of one un-padded plane; params = structured array (CTB rows, CTB cols) of SAO_CTB_DTYPE; returns a new array."""
    src = np.ascontiguousarray(plane)
    dst = np.empty_like(src)
    h, w = src.shape
    prm = np.ascontiguousarray(params, SAO_CTB_DTYPE)
    assert prm.shape[0] >= (h + (1 << ctb_log2) - 1) >> ctb_log2 and prm.shape[1] >= (w + (1 << ctb_log2) - 1) >> ctb_log2
    k = None if keep is None else np.ascontiguousarray(keep, np.uint8)
    rc = _lib().dbko_h265_sao_plane(
        src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(src.strides[0]), bit_depth, src.itemsize,
        prm.ctypes.data_as(C.c_void_p), prm.shape[1], ctb_log2, None if k is None else k.ctypes.data_as(C.c_void_p),
        0 if k is None else k.shape[1])
    if rc:
        raise RuntimeError("dbko_h265_sao_plane: %d" % rc)
    return dst


def merge_sao_params(p, seed, p_left=0.5, p_up=0.25):
    """What sao_merge_left_flag / sao_merge_up_flag do to a picture's parameters: in raster order a CTB takes its left
    neighbour's entry with probability p_left, else the entry above with probability p_up, else keeps its own."""
    rng = np.random.RandomState(seed)
    out = p.copy()
    rows, cols = out.shape
    r = rng.random_sample((rows, cols, 2))
    for y in range(rows):
        for x in range(cols):
            if x > 0 and r[y, x, 0] < p_left:
                out[y, x] = out[y, x - 1]
            elif y > 0 and r[y, x, 1] < p_up:
                out[y, x] = out[y - 1, x]
    return out


def random_sao_params(w, h, ctb_log2, seed, bit_depth=8):
    rng = np.random.RandomState(seed)
    rows, cols = (h + (1 << ctb_log2) - 1) >> ctb_log2, (w + (1 << ctb_log2) - 1) >> ctb_log2
    p = np.zeros((rows, cols), SAO_CTB_DTYPE)
    p["type"] = rng.randint(0, 3, (rows, cols))
    band = p["type"] == 1
    p["cls"] = np.where(band, rng.randint(0, 32, (rows, cols)), rng.randint(0, 4, (rows, cols)))
    lim = (1 << (min(bit_depth, 10) - 5)) - 1
    off = rng.randint(-lim, lim + 1, (rows, cols, 4))
    # edge offsets: the first two are non-negative, the last two non-positive (7.4.9.3.2); band offsets keep their signs
    eo = ~band
    off[eo, 0:2] = np.abs(off[eo, 0:2])
    off[eo, 2:4] = -np.abs(off[eo, 2:4])
    p["offset"] = off
    return p
