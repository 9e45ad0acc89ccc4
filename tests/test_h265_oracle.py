"""Spec-exact mode (H.265 clause 8.7.2), CPU side.

The oracle for this mode (oracle/h265_oracle.c) is PARITY UNPINNED: the reference does not implement the standard's
filter and no HEVC decoder is available here.  What these tests establish without a GPU:
  * the oracle agrees with the PINNED reference-mode oracle wherever the two modes must coincide,
  * its vertical and horizontal passes are each other's transpose,
  * hand-derived known answers for the decisions, the chroma filter, the tables and the bS rules,
  * the kernels' block-local arithmetic (csrc/deblock_h265.h, run on the CPU by tests/host_sim) equals the oracle's
    picture-order arithmetic bit for bit -- luma, chroma, 8 and 10 bit, QP maps, offsets, keep flags, derived bS.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")


@pytest.fixture(scope="module")
def h265():
    from oracle import h265 as h
    return h


@pytest.fixture(scope="module")
def sim():
    subprocess.check_call(["make", "-s", "-C", SIM_DIR])
    L = C.CDLL(os.path.join(SIM_DIR, "libdbk_hostsim.so"))
    L.host_sim_h265_filter_plane.restype = None
    L.host_sim_h265_derive_bs.restype = None
    return L


def full_bs(h265, w, h, value, rng=None):
    """bS arrays with `value` on every interior edge segment (or random 0..2 + keep bits when rng is given)."""
    vb = np.zeros((h // 4, w // 8 + 1), np.uint8)
    hb = np.zeros((h // 8 + 1, w // 4), np.uint8)
    if rng is None:
        vb[:, 1:w // 8] = value
        hb[1:h // 8, :] = value
    else:
        vb[:] = rng.randint(0, 3, vb.shape) | (rng.randint(0, 8, vb.shape) == 0) * 4 | (rng.randint(0, 8, vb.shape) == 0) * 8
        hb[:] = rng.randint(0, 3, hb.shape) | (rng.randint(0, 8, hb.shape) == 0) * 4 | (rng.randint(0, 8, hb.shape) == 0) * 8
    return vb.ravel(), hb.ravel()


def sim_filter(sim, plane, qp, vb, hb, *, c_idx=0, bit_depth=8, qp_map=None, unit_log2=3, tc_off=0, beta_off=0, c_qp_off=0,
               packed=0):
    out = np.ascontiguousarray(plane).copy()
    h, w = out.shape
    vb = np.ascontiguousarray(vb, np.uint8)
    hb = np.ascontiguousarray(hb, np.uint8)
    m = None if qp_map is None else np.ascontiguousarray(qp_map, np.uint8)
    sim.host_sim_h265_filter_plane(
        out.ctypes.data_as(C.c_void_p), w, h, C.c_long(out.strides[0]), out.itemsize, bit_depth, c_idx,
        vb.ctypes.data_as(C.c_void_p), hb.ctypes.data_as(C.c_void_p), int(qp),
        None if m is None else m.ctypes.data_as(C.c_void_p), 0 if m is None else m.shape[1], unit_log2,
        tc_off, beta_off, c_qp_off, packed)
    return out


def test_tables_extend_the_reference_tables(h265, oracle):
    tc, beta = h265.tables()
    rtc, rbeta = oracle.tables()
    assert np.array_equal(tc[:52], rtc) and np.array_equal(beta, rbeta)
    assert tc[52] == 22 and tc[53] == 24


def test_agrees_with_pinned_reference_mode_where_the_modes_coincide(h265, oracle):
    """Vertical edges only, bS 1 (tc index = QP in both modes), piecewise-constant 8x8 blocks whose steps are small
    enough that the normal delta never exceeds tc (|step| <= 10 at tc 4: (6*10+8)>>4 = 4) and an even tc so 5tc/2 == (5tc+1)>>1:
    every decision and every filtered value must be the same in the standard's filter and in the reference's."""
    rng = np.random.RandomState(5)
    for bit_depth, qp in ((8, 36), (8, 37), (10, 36)):
        w, h = 96, 64
        sc = 1 << (bit_depth - 8)
        levels = 120 + np.cumsum(rng.randint(-10, 11, (h // 8, w // 8)), axis=1)
        plane = (np.kron(levels, np.ones((8, 8), np.int64)) * sc).astype(np.uint8 if bit_depth == 8 else np.uint16)
        vb4, hb4 = full_bs(h265, w, h, 1)
        hb4[:] = 0
        got = h265.filter_plane(plane, qp, vb4, hb4, bit_depth=bit_depth)
        rvb = np.zeros((h // 8, w // 8 + 1), np.uint8)
        rvb[:, 1:w // 8] = 1  # reference granularity: one bS per 8 rows; x = 0 and x = W off
        rhb = np.zeros(oracle.num_hor_bs(w, h), np.uint8)
        want = oracle.filter_plane(plane, qp, bit_depth=bit_depth, vert_bs=rvb.ravel(), hor_bs=rhb)
        assert np.array_equal(got, want), (bit_depth, qp)
        assert not np.array_equal(got, plane)


def test_vertical_and_horizontal_passes_are_transposes(h265):
    from gpu_video_codec_amd import synth
    rng = np.random.RandomState(9)
    w, h = 64, 96
    y = synth.blocky_plane(w, h, seed=3)
    vb = np.zeros((h // 4, w // 8 + 1), np.uint8)
    vb[:, 1:w // 8] = rng.randint(0, 3, (h // 4, w // 8 - 1))
    hb0 = np.zeros(h265.num_hor_bs(w, h), np.uint8)
    a = h265.filter_plane(y, 33, vb.ravel(), hb0)
    # transposed problem: plane^T is h x w, its horizontal edges at y' = 8*bx, segments x4' = y4
    hbT = np.ascontiguousarray(vb.T)                      # (w/8+1, h/4) == hor layout of the transposed plane
    vbT = np.zeros(h265.num_vert_bs(h, w), np.uint8)
    b = h265.filter_plane(np.ascontiguousarray(y.T), 33, vbT, hbT.ravel())
    assert np.array_equal(a, b.T)
    assert not np.array_equal(a, y)


def test_known_answers_luma_and_chroma(h265):
    """Single segments worked by hand from 8.7.2.5.3 / .6 / .7 / .8."""
    w, h = 16, 8
    vb = np.zeros((h // 4, w // 8 + 1), np.uint8)
    hb = np.zeros(h265.num_hor_bs(w, h), np.uint8)
    # flat 100 | 110 step, QP 37, bS 2 -> tc index 39 -> tc 5, beta 36: d = 0 < beta; dSam: 0 < 9, 0 < 4, 10 < 13 -> strong
    plane = np.full((h, w), 100, np.uint8)
    plane[:, 8:] = 110
    vb[:, 1] = 2
    out = h265.filter_plane(plane, 37, vb.ravel(), hb)
    # p2' = (2*100+3*100+100+100+110+4)>>3 = 101; p1' = (100+100+100+110+2)>>2 = 103; p0' = (100+200+200+220+110+4)>>3 = 104
    # q0' = (100+200+220+220+110+4)>>3 = 106; q1' = (100+110+110+110+2)>>2 = 108; q2' = (100+110+110+330+220+4)>>3 = 109
    assert list(out[0, 4:12]) == [100, 101, 103, 104, 106, 108, 109, 110]
    # same step at bS 1 -> tc index 37 -> tc 4: |p0-q0| = 10 < (20+1)>>1 = 10 is false -> normal filter:
    # delta = (9*10 - 0 + 8)>>4 = 6 -> clipped to tc = 4; p0' = 104, q0' = 106;
    # dEp: 0 < (36+18)>>3 = 6 -> p1' = 100 + clip(+-2, (((100+100+1)>>1) - 100 + 4)>>1 = 2) = 102; q1' = 110 + clip(((110+110+1)>>1) - 110 - 4)>>1 = -2 -> 108
    vb[:, 1] = 1
    out = h265.filter_plane(plane, 37, vb.ravel(), hb)
    assert list(out[0, 4:12]) == [100, 100, 102, 104, 106, 108, 110, 110]
    # keep-P: only the Q side moves
    vb[:, 1] = 1 | h265.KEEP_P
    out = h265.filter_plane(plane, 37, vb.ravel(), hb)
    assert list(out[0, 4:12]) == [100, 100, 100, 100, 106, 108, 110, 110]
    # chroma: bS 1 is ignored; bS 2 at QpI = 37 + 2 = 39 -> QpC 35 -> tc index 37 -> tc 4;
    # delta = clip(+-4, ((10<<2) + 100 - 110 + 4)>>3 = 4) = 4 -> p0' = 104, q0' = 106, nothing else moves
    vb[:, 1] = 1
    assert np.array_equal(h265.filter_plane(plane, 37, vb.ravel(), hb, c_idx=1, c_qp_offset=2), plane)
    vb[:, 1] = 2
    out = h265.filter_plane(plane, 37, vb.ravel(), hb, c_idx=1, c_qp_offset=2)
    assert list(out[3, 4:12]) == [100, 100, 100, 104, 106, 110, 110, 110]
    # picture-boundary columns of the array are ignored
    vb[:] = 0
    vb[:, 0] = 2
    vb[:, 2] = 2
    assert np.array_equal(h265.filter_plane(plane, 37, vb.ravel(), hb), plane)


def test_bs_rules(h265):
    """8.7.2.4 case by case on a 16x16 picture: one vertical edge at x = 8 with four 4-row segments, one horizontal."""
    w = h = 16
    H = h265

    def units():
        return (np.zeros((4, 4), np.uint16), np.zeros((4, 4, 2), np.int16), np.zeros((4, 4, 2), np.int16),
                np.zeros((4, 4), np.int32), np.zeros((4, 4), np.int32))

    def vseg(f, mv0, mv1, r0, r1):
        vb, hb = H.derive_bs(f, mv0, mv1, r0, r1, w, h)
        return vb.reshape(4, 3)[:, 1], hb.reshape(3, 4)[1]

    f, mv0, mv1, r0, r1 = units()
    f[:] = H.U_PRED_L0
    f[:, 2] |= H.U_TU_LEFT | H.U_PU_LEFT          # units (.,2) are the Q side of the edge x = 8
    f[0, 1] |= H.U_INTRA                           # row 0: P intra -> 2
    f[1, 2] |= H.U_CBF                             # row 1: coefficients on a transform edge -> 1
    mv0[2, 2] = (3, 0)                             # row 2: |dmv| = 3 < 4, same picture -> 0
    mv0[3, 2] = (0, -4)                            # row 3: |dmv| = 4 -> 1
    v, hseg = vseg(f, mv0, mv1, r0, r1)
    assert list(v) == [2, 1, 0, 1] and not hseg.any()
    # prediction-only edge: cbf does not count; different reference picture does; KEEP and the switches
    f, mv0, mv1, r0, r1 = units()
    f[:] = H.U_PRED_L0
    f[:, 2] |= H.U_PU_LEFT
    f[0, 2] |= H.U_CBF                             # no transform edge -> still 0
    r0[1, 2] = 7                                   # other picture -> 1
    f[2, 1] |= H.U_INTRA | H.U_KEEP                # 2 + keep P
    f[3, 2] |= H.U_INTRA | H.U_DBK_OFF             # deblocking disabled in Q's slice -> 0
    v, _ = vseg(f, mv0, mv1, r0, r1)
    assert list(v) == [0, 1, 2 | H.KEEP_P, 0]
    # two motion vectors
    f, mv0, mv1, r0, r1 = units()
    f[:] = H.U_PRED_L0 | H.U_PRED_L1
    f[:, 2] |= H.U_PU_LEFT
    r0[:], r1[:] = 1, 2
    # row 0: Q swaps the lists (ref0 = 2, ref1 = 1) with matching vectors -> 0
    r0[0, 2], r1[0, 2] = 2, 1
    mv0[0, 1], mv1[0, 1] = (5, 5), (-3, 2)
    mv0[0, 2], mv1[0, 2] = (-3, 2), (5, 5)
    # row 1: same pictures, list-1 vector differs by 4 -> 1
    mv1[1, 2] = (4, 0)
    # row 2: number of vectors differs -> 1
    f[2, 2] &= ~np.uint16(H.U_PRED_L1)
    # row 3: both vectors to the same picture on both sides: needs BOTH pairings to be far
    r0[3, 1] = r1[3, 1] = r0[3, 2] = r1[3, 2] = 4
    mv0[3, 1], mv1[3, 1] = (0, 0), (8, 0)
    mv0[3, 2], mv1[3, 2] = (8, 0), (0, 0)          # straight pairing far, crossed pairing equal -> 0
    v, _ = vseg(f, mv0, mv1, r0, r1)
    assert list(v) == [0, 1, 1, 0]
    mv1[3, 2] = (0, 4)                             # now the crossed pairing is far as well -> 1
    v, _ = vseg(f, mv0, mv1, r0, r1)
    assert v[3] == 1
    # horizontal edge, NOX switch, off-grid transform edge, picture boundary
    f, mv0, mv1, r0, r1 = units()
    f[:] = H.U_INTRA
    f[2, :] |= H.U_TU_TOP
    f[2, 3] |= H.U_NOX_TOP
    f[1, :] |= H.U_TU_TOP                          # y = 4 is not on the 8x8 grid
    f[0, :] |= H.U_TU_TOP                          # y = 0 is the picture boundary
    f[:, 0] |= H.U_TU_LEFT
    vb, hb = H.derive_bs(f, mv0, mv1, r0, r1, w, h)
    assert list(hb.reshape(3, 4)[1]) == [2, 2, 2, 0] and not hb.reshape(3, 4)[[0, 2]].any() and not vb.any()


def test_chroma_bs_is_the_luma_bs_at_twice_the_position(h265):
    w, h = 64, 48
    rng = np.random.RandomState(2)
    vb, hb = full_bs(h265, w, h, 0, rng)
    cvb, chb = h265.chroma_bs(vb, hb, w, h)
    V, Hh = vb.reshape(h // 4, w // 8 + 1), hb.reshape(h // 8 + 1, w // 4)
    assert np.array_equal(cvb.reshape(h // 8, w // 16 + 1), V[0::2, 0::2])
    assert np.array_equal(chb.reshape(h // 16 + 1, w // 8), Hh[0::2, 0::2])


def test_kernel_block_arithmetic_equals_picture_order_oracle(h265, sim):
    """csrc/deblock_h265.h (offset blocks, ver1 -> ver2 -> hor1 -> hor2 per block) against the oracle (all vertical
    edges of the picture, then all horizontal edges): luma and chroma, 8 and 10 bit, random bS with keep bits, scalar QP
    and per-8x8 QP maps, tc / beta / chroma QP offsets."""
    from gpu_video_codec_amd import synth
    rng = np.random.RandomState(11)
    cases = 0
    for (w, h, bd) in [(64, 48, 8), (16, 16, 8), (8, 8, 8), (136, 72, 8), (64, 64, 10), (72, 40, 12)]:
        plane = synth.blocky_plane(w, h, seed=w + h + bd, bit_depth=bd)
        for qp in (22, 27, 32, 37, 44, 51):
            for c_idx in (0, 1):
                vb, hb = full_bs(h265, w, h, 2) if qp == 32 else full_bs(h265, w, h, 0, rng)
                offs = dict(tc_off=int(rng.randint(-6, 7)), beta_off=int(rng.randint(-6, 7)), c_qp_off=int(rng.randint(-12, 13)))
                if qp == 32:
                    offs = dict(tc_off=0, beta_off=0, c_qp_off=0)
                sc = 2 if c_idx else 1
                for use_map in (False, True):
                    qmap = None
                    if use_map:
                        qmap = rng.randint(max(0, qp - 8), min(51, qp + 8) + 1, ((h * sc + 7) // 8, (w * sc + 7) // 8)).astype(np.uint8)
                    want = h265.filter_plane(plane, qp, vb, hb, c_idx=c_idx, bit_depth=bd, qp_map=qmap, unit_log2=3,
                                             tc_offset_div2=offs["tc_off"], beta_offset_div2=offs["beta_off"],
                                             c_qp_offset=offs["c_qp_off"])
                    # 1 = the packed-int16 form the fast kernels run (up to 12 bit)
                    for packed in (0, 1):
                        got = sim_filter(sim, plane, qp, vb, hb, c_idx=c_idx, bit_depth=bd, qp_map=qmap, unit_log2=3,
                                         packed=packed, **offs)
                        assert np.array_equal(got, want), (w, h, bd, qp, c_idx, use_map, offs, packed)
                    cases += (got != plane).any()
    assert cases > 80  # the filter really ran in most cases
    # round 4: the one-QP kernels' form for a wave that holds bS 1 next to bS 2 (LumaKSel: every tc-dependent operand one of two
    # scalars, picked per lane) -- on the CPU a "wave" is one block, so the simulator sends EVERY block through that form
    sim.host_sim_h265_force_mixed(1)
    try:
        for (w, h, bd) in [(136, 72, 8), (64, 64, 10), (72, 40, 12)]:
            plane = synth.blocky_plane(w, h, seed=3 * w + bd, bit_depth=bd)
            for qp in (18, 27, 37, 46, 51):
                vb, hb = full_bs(h265, w, h, 0, rng)
                offs = dict(tc_off=int(rng.randint(-6, 7)), beta_off=int(rng.randint(-6, 7)), c_qp_off=0)
                want = h265.filter_plane(plane, qp, vb, hb, bit_depth=bd, tc_offset_div2=offs["tc_off"], beta_offset_div2=offs["beta_off"])
                got = sim_filter(sim, plane, qp, vb, hb, bit_depth=bd, packed=1, **offs)
                assert np.array_equal(got, want), ("mixed form", w, h, bd, qp, offs)
    finally:
        sim.host_sim_h265_force_mixed(0)
    # samples hugging 0 and 255 with large tc: the conditional Clip1 of the packed form must fire
    w, h = 64, 32
    base = np.where(rng.randint(0, 2, (h // 8, w // 8)) == 0, 2, 252)
    plane = np.clip(np.kron(base, np.ones((8, 8), np.int64)) + rng.randint(-2, 4, (h, w)), 0, 255).astype(np.uint8)
    plane[:, ::16] = np.clip(plane[:, ::16].astype(int) + 9, 0, 255)
    for qp in (45, 51):
        for c_idx in (0, 1):
            vb, hb = full_bs(h265, w, h, 2)
            want = h265.filter_plane(plane, qp, vb, hb, c_idx=c_idx, tc_offset_div2=6, beta_offset_div2=6)
            for packed in (0, 1):
                got = sim_filter(sim, plane, qp, vb, hb, c_idx=c_idx, tc_off=6, beta_off=6, packed=packed)
                assert np.array_equal(got, want), (qp, c_idx, packed)
            assert (want != plane).any()


def test_kernel_bs_derivation_equals_oracle(h265, sim):
    for (w, h, seed) in [(64, 64, 1), (136, 72, 2), (16, 16, 3), (8, 8, 4), (352, 288, 5)]:
        flags, mv0, mv1, r0, r1 = h265.random_units(w, h, seed)
        vb, hb = h265.derive_bs(flags, mv0, mv1, r0, r1, w, h)
        gv = np.zeros_like(vb)
        gh = np.zeros_like(hb)
        sim.host_sim_h265_derive_bs(flags.ctypes.data_as(C.c_void_p), mv0.ctypes.data_as(C.c_void_p),
                                    mv1.ctypes.data_as(C.c_void_p), r0.ctypes.data_as(C.c_void_p),
                                    r1.ctypes.data_as(C.c_void_p), w, h, gv.ctypes.data_as(C.c_void_p),
                                    gh.ctypes.data_as(C.c_void_p))
        assert np.array_equal(gv, vb) and np.array_equal(gh, hb), (w, h)
        if w >= 64:
            vals = set(np.unique(np.concatenate([vb, hb])))
            assert {0, 1, 2} <= vals and any(v & 4 for v in vals) and any(v & 8 for v in vals)


def test_sao_known_answers(h265):
    """8.7.3 by hand: band table wrap-around, each edge class and category, picture borders, clipping, keep map."""
    P = h265.SAO_CTB_DTYPE
    plane = np.array([[10, 10, 10, 10, 10, 10, 10, 10],
                      [10, 50, 10, 10, 10, 10, 10, 10],
                      [10, 10, 10, 3, 10, 10, 10, 10],
                      [10, 10, 10, 10, 10, 20, 20, 20],
                      [10, 10, 10, 10, 10, 20, 20, 20],
                      [10, 10, 10, 10, 10, 20, 20, 20],
                      [250, 250, 250, 250, 10, 10, 10, 10],
                      [250, 250, 250, 250, 10, 10, 10, 10]], np.uint8)
    prm = np.zeros((1, 1), P)
    # off: identity
    assert np.array_equal(h265.sao_plane(plane, prm, 4), plane)
    # band offset, bands of 8 values: position 31 covers bands 31, 0, 1, 2 -> values 248..255 get offset[0], 0..7 offset[1],
    # 8..15 offset[2], 16..23 offset[3]
    prm[0, 0] = (1, 31, (7, -2, 1, -3))
    out = h265.sao_plane(plane, prm, 4)
    assert out[6, 0] == 255 and out[2, 3] == 1 and out[0, 0] == 11 and out[3, 5] == 17 and out[1, 1] == 50
    # edge class 0 (horizontal): the peak 50 is category 4 (both neighbours lower), the valley 3 category 1
    prm[0, 0] = (2, 0, (4, 2, -2, -5))
    out = h265.sao_plane(plane, prm, 4)
    assert out[1, 1] == 45 and out[2, 3] == 7
    assert out[1, 0] == 10 and out[1, 2] == 12  # (1,0): picture border -> untouched; (1,2): 50 left, 10 right -> category 2
    assert out[3, 4] == 12 and out[3, 5] == 18   # step 10|20: low side category 2 (+2), high side category 3 (-2)
    # class 1 (vertical): rows 0 and 7 untouched; (2,1) below the peak: upper 50, lower 10 -> category 2
    prm[0, 0] = (2, 1, (4, 2, -2, -5))
    out = h265.sao_plane(plane, prm, 4)
    assert np.array_equal(out[0], plane[0]) and np.array_equal(out[7], plane[7]) and out[1, 1] == 45 and out[2, 1] == 12
    # classes 2 and 3 use the diagonals: the peak's diagonal neighbours see it on one side only
    prm[0, 0] = (2, 2, (4, 2, -2, -5))
    out = h265.sao_plane(plane, prm, 4)
    assert out[2, 2] == 12 and out[2, 0] == 10   # (2,2): a = (1,1) = 50, b = (3,3) = 10 -> category 2; (2,0) at the border
    prm[0, 0] = (2, 3, (4, 2, -2, -5))
    out = h265.sao_plane(plane, prm, 4)
    assert out[2, 0] == 10 and out[1, 1] == 45 and out[2, 2] == 10
    # keep map: the 8x8 block stays as it is
    prm[0, 0] = (1, 31, (7, -2, 1, -3))
    assert np.array_equal(h265.sao_plane(plane, prm, 4, keep=np.ones((1, 1), np.uint8)), plane)
    # 10 bit: bands of 32 values, offsets up to 31, clip to 1023
    p10 = (plane.astype(np.uint16) * 4)
    prm[0, 0] = (1, 31, (31, -2, 1, -3))
    out = h265.sao_plane(p10, prm, 4, bit_depth=10)
    assert out[6, 0] == 1023 and out[0, 0] == 41 and out[2, 3] == 10
