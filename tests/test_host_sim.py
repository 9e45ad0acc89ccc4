"""The kernels' per-block arithmetic (csrc/deblock_core.h, csrc/deblock_packed.h), compiled for the
CPU by tests/host_sim and compared with the oracle bit-for-bit.  Catches arithmetic / segment-order
/ guard bugs in the kernel source before it ever reaches a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")


@pytest.fixture(scope="module")
def sim():
    subprocess.check_call(["make", "-s", "-C", SIM_DIR])
    L = C.CDLL(os.path.join(SIM_DIR, "libdbk_hostsim.so"))
    L.host_sim_filter_plane.restype = None
    return L


def run_sim(sim, oracle, plane, qp, *, is_chroma=False, bit_depth=8, vbs=None, hbs=None, qp_map=None, packed=0,
            tc_table=None, beta_table=None):
    out = np.ascontiguousarray(plane).copy()
    h, w = out.shape
    tc_t, beta_t = oracle.tables()
    if tc_table is not None:
        tc_t = np.asarray(tc_table)
    if beta_table is not None:
        beta_t = np.asarray(beta_table)
    dv, dh = oracle.default_bs(w, h)
    vbs = dv if vbs is None else np.ascontiguousarray(vbs, np.uint8)
    hbs = dh if hbs is None else np.ascontiguousarray(hbs, np.uint8)
    q = min(qp, 51)
    shift = bit_depth - 8
    tct = tc_t.astype(np.uint8)
    bt = beta_t.astype(np.uint8)
    m = None if qp_map is None else np.ascontiguousarray(qp_map, np.uint8)
    sim.host_sim_filter_plane(
        out.ctypes.data_as(C.c_void_p), w, h, C.c_long(out.strides[0]), out.itemsize, int(is_chroma),
        vbs.ctypes.data_as(C.c_void_p), hbs.ctypes.data_as(C.c_void_p),
        int(tc_t[q]) << shift, int(beta_t[q]) << shift, (1 << bit_depth) - 1,
        None if m is None else m.ctypes.data_as(C.c_void_p), 0 if m is None else m.shape[1], 6,
        tct.ctypes.data_as(C.c_void_p), bt.ctypes.data_as(C.c_void_p), shift, packed)
    return out


def variants(sim):
    return [0, 1] if sim.host_sim_have_packed() else [0]


def test_bundled_images_all_planes(sim, oracle, golden_inputs, manifest):
    for packed in variants(sim):
        for name, ent in manifest["images"].items():
            w, h = ent["width"], ent["height"]
            y, u, v = oracle.split_yuv420(golden_inputs[name], w, h)
            for qp in (22, 30, 35, 37, 45, 51):
                assert np.array_equal(run_sim(sim, oracle, y, qp, packed=packed), oracle.filter_plane(y, qp)), (name, qp, packed)
                for c in (u, v):
                    assert np.array_equal(run_sim(sim, oracle, c, qp, is_chroma=True, packed=packed),
                                          oracle.filter_plane(c, qp, is_chroma=True)), (name, qp, packed)


def test_random_bs_and_hard_content(sim, oracle):
    from gpu_video_codec_amd import synth
    rng = np.random.default_rng(5)
    for packed in variants(sim):
        for (w, h) in [(8, 8), (16, 8), (8, 24), (64, 48), (200, 120), (352, 288)]:
            for qp in (18, 27, 33, 40, 51):
                y = synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30))).copy()
                y[: max(h // 4, 1), : max(w // 4, 1)] = rng.integers(0, 256, (max(h // 4, 1), max(w // 4, 1)), dtype=np.uint8)
                y[h // 2:, w // 2:] = 255
                y[h // 2:, : w // 8] = 0
                vb, hb = oracle.lcg_bs(w, h, int(rng.integers(1, 1000)))
                assert np.array_equal(run_sim(sim, oracle, y, qp, vbs=vb, hbs=hb, packed=packed),
                                      oracle.filter_plane(y, qp, vert_bs=vb, hor_bs=hb)), (w, h, qp, packed)
                # chroma with overridden bS (extension): exercises bS 1 (skipped) and the shifted read of Q9(ii)
                assert np.array_equal(run_sim(sim, oracle, y, qp, is_chroma=True, vbs=vb, hbs=hb, packed=packed),
                                      oracle.filter_plane(y, qp, is_chroma=True, vert_bs=vb, hor_bs=hb)), (w, h, qp, packed)


def test_pure_noise_and_extremes(sim, oracle):
    rng = np.random.default_rng(7)
    for packed in variants(sim):
        for trial in range(6):
            y = rng.integers(0, 256, (64, 96), dtype=np.uint8)
            if trial % 2:
                y = (y // 64 * 85).astype(np.uint8)  # coarse steps: many strong/normal hits incl. clipping at 0/255
            for qp in (30, 51):
                assert np.array_equal(run_sim(sim, oracle, y, qp, packed=packed), oracle.filter_plane(y, qp))
                assert np.array_equal(run_sim(sim, oracle, y, qp, is_chroma=True, packed=packed),
                                      oracle.filter_plane(y, qp, is_chroma=True))


def test_16bit_and_qp_map_generic(sim, oracle):
    from gpu_video_codec_amd import synth
    y10 = synth.blocky_plane(256, 128, seed=9, bit_depth=10)
    assert np.array_equal(run_sim(sim, oracle, y10, 32, bit_depth=10), oracle.filter_plane(y10, 32, bit_depth=10))
    assert np.array_equal(run_sim(sim, oracle, y10, 40, bit_depth=10, is_chroma=True),
                          oracle.filter_plane(y10, 40, bit_depth=10, is_chroma=True))
    y = synth.blocky_plane(768, 576, seed=2)
    qmap = synth.ctu_qp_map(768, 576, seed=4)
    assert np.array_equal(run_sim(sim, oracle, y, 0, qp_map=qmap), oracle.filter_plane(y, 0, qp_map=qmap))
    c = synth.blocky_plane(384, 288, seed=3)
    assert np.array_equal(run_sim(sim, oracle, c, 0, qp_map=qmap, is_chroma=True),
                          oracle.filter_plane(c, 0, qp_map=qmap, is_chroma=True))


def test_packed_16bit_luma(sim, oracle, golden_inputs):
    """The packed core on 16-bit containers: 10/11-bit synthetic data vs the oracle, and 8-bit data in
    16-bit containers vs the reference-pinned 8-bit result (the pinned degenerate case)."""
    if not sim.host_sim_have_packed():
        pytest.skip("packed core not built")
    from gpu_video_codec_amd import synth
    rng = np.random.default_rng(21)
    for bd in (10, 11, 12):  # up to 11 bit every intermediate fits int16; 12 bit takes the WIDE variant of the core
        for (w, h) in [(8, 8), (64, 48), (520, 72)]:
            for qp in (22, 32, 45, 51):
                y = synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30)), bit_depth=bd).copy()
                y[: max(h // 4, 1), : max(w // 4, 1)] = rng.integers(0, 1 << bd, (max(h // 4, 1), max(w // 4, 1)), dtype=np.uint16)
                y[h // 2:, w // 2:] = (1 << bd) - 1
                vb, hb = oracle.lcg_bs(w, h, int(rng.integers(1, 1000)))
                got = run_sim(sim, oracle, y, qp, bit_depth=bd, vbs=vb, hbs=hb, packed=1)
                assert np.array_equal(got, oracle.filter_plane(y, qp, bit_depth=bd, vert_bs=vb, hor_bs=hb)), (bd, w, h, qp)
    y8, _, _ = oracle.split_yuv420(golden_inputs["image2"], 768, 576)
    got = run_sim(sim, oracle, y8.astype(np.uint16), 30, bit_depth=8, packed=1)
    assert np.array_equal(got, oracle.filter_plane(y8, 30).astype(np.uint16))


def test_packed_16bit_chroma(sim, oracle, golden_inputs):
    """Chroma through the packed core on 16-bit containers (csrc/deblock_packed16.h): 10 and 12 bit vs the oracle, the
    guard quirk at bx == CW/8 included, and 8-bit data in 16-bit containers vs the reference-pinned 8-bit result."""
    if not sim.host_sim_have_packed():
        pytest.skip("packed core not built")
    from gpu_video_codec_amd import synth
    rng = np.random.default_rng(22)
    for bd in (10, 12):  # 5*max_v + 4 fits int16 up to 12 bit
        for (w, h) in [(8, 8), (64, 48), (264, 72)]:
            for qp in (22, 32, 45, 51):
                c = synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30)), bit_depth=bd, dc_range=4).copy()
                c[: max(h // 4, 1), : max(w // 4, 1)] = rng.integers(0, 1 << bd, (max(h // 4, 1), max(w // 4, 1)), dtype=np.uint16)
                c[h // 2:, w // 2:] = (1 << bd) - 1
                vb, hb = oracle.lcg_bs(w, h, int(rng.integers(1, 1000)))
                got = run_sim(sim, oracle, c, qp, is_chroma=True, bit_depth=bd, vbs=vb, hbs=hb, packed=1)
                want = oracle.filter_plane(c, qp, is_chroma=True, bit_depth=bd, vert_bs=vb, hor_bs=hb)
                assert np.array_equal(got, want), (bd, w, h, qp)
                got = run_sim(sim, oracle, c, qp, is_chroma=True, bit_depth=bd, packed=1)  # default bS: Q9 shifted read
                assert np.array_equal(got, oracle.filter_plane(c, qp, is_chroma=True, bit_depth=bd)), (bd, w, h, qp)
    _, u8, v8 = oracle.split_yuv420(golden_inputs["image2"], 768, 576)
    for c8 in (u8, v8):
        got = run_sim(sim, oracle, c8.astype(np.uint16), 30, is_chroma=True, bit_depth=8, packed=1)
        assert np.array_equal(got, oracle.filter_plane(c8, 30, is_chroma=True).astype(np.uint16))


def test_packed_with_ctu_qp_map(sim, oracle):
    """Per-segment tc/beta (the per-CTU QP map extension) through the packed core: 8-bit luma and chroma,
    10-bit luma; QPs spanning tc == 0 (QP < 18) up to QP 51 so every threshold edge case occurs."""
    if not sim.host_sim_have_packed():
        pytest.skip("packed core not built")
    from gpu_video_codec_amd import synth
    y = synth.blocky_plane(768, 576, seed=2)
    for lo, hi, seed in ((22, 42, 4), (10, 51, 5), (0, 20, 6)):
        qmap = synth.ctu_qp_map(768, 576, seed=seed, lo=lo, hi=hi)
        vb, hb = oracle.lcg_bs(768, 576, seed)
        assert np.array_equal(run_sim(sim, oracle, y, 0, qp_map=qmap, vbs=vb, hbs=hb, packed=1),
                              oracle.filter_plane(y, 0, qp_map=qmap, vert_bs=vb, hor_bs=hb)), (lo, hi)
        c = synth.blocky_plane(384, 288, seed=3)
        assert np.array_equal(run_sim(sim, oracle, c, 0, qp_map=qmap, is_chroma=True, packed=1),
                              oracle.filter_plane(c, 0, qp_map=qmap, is_chroma=True)), (lo, hi)
        y10 = synth.blocky_plane(768, 576, seed=7, bit_depth=10)
        assert np.array_equal(run_sim(sim, oracle, y10, 0, qp_map=qmap, bit_depth=10, packed=1),
                              oracle.filter_plane(y10, 0, qp_map=qmap, bit_depth=10)), (lo, hi)
        # round 4: the operands come out of the per-QP table (deblock_packed.h ktab_build / LumaKEager::load) -- 12 bit takes the
        # WIDE sums through the same rows, and caller tables with large entries fill rows the default tables never reach
        y12 = synth.blocky_plane(768, 576, seed=9, bit_depth=12)
        assert np.array_equal(run_sim(sim, oracle, y12, 0, qp_map=qmap, bit_depth=12, packed=1),
                              oracle.filter_plane(y12, 0, qp_map=qmap, bit_depth=12)), (lo, hi)
        tct = (np.arange(52) * 3 % 97 + (np.arange(52) > 40) * 30).astype(np.uint8)
        bt = (np.arange(52) * 5 % 200).astype(np.uint8)
        assert np.array_equal(run_sim(sim, oracle, y, 0, qp_map=qmap, vbs=vb, hbs=hb, packed=1, tc_table=tct, beta_table=bt),
                              oracle.filter_plane(y, 0, qp_map=qmap, vert_bs=vb, hor_bs=hb, tc_table=tct, beta_table=bt)), (lo, hi)


def test_packed_core_operand_range_with_maximal_custom_tables(sim, oracle):
    """Caller-supplied tc / beta tables (hevcdbk_tables) may hold entries up to 255, scaled by 1 << (bit_depth - 8): the
    launcher hands a plane to the packed luma core only while packed_luma_tc_fits() holds.  Inside that range -- including
    exactly at its edge -- the packed arithmetic must equal the oracle; the first value beyond it must be refused (and is
    known to overflow the 16-bit fields)."""
    if not sim.host_sim_have_packed():
        pytest.skip("packed core not compiled into the host sim")
    rng = np.random.default_rng(2024)
    for bd, edge in ((8, 255), (10, 255), (11, 128), (12, 128)):
        max_v, shift = (1 << bd) - 1, bd - 8
        assert sim.host_sim_packed_luma_tc_fits(max_v, edge << shift) == 1, bd
        if edge < 255:
            assert sim.host_sim_packed_luma_tc_fits(max_v, (edge + 1) << shift) == 0, bd
        h, w = 64, 256
        # flat areas with small steps (strong filter), ramps (normal filter) and full-range noise (clipping at 0 / max_v)
        y = np.full((h, w), max_v // 2, np.int64)
        y[:, : w // 4] += rng.integers(-3, 4, (h, w // 4)) << shift
        y[:, w // 4: w // 2] = rng.integers(0, max_v + 1, (h, w // 4))
        y[:, w // 2: 3 * w // 4] = (np.arange(w // 4)[None, :] * (max_v // 80) + rng.integers(0, 2 << shift, (h, w // 4))) % (max_v + 1)
        y[:, 3 * w // 4:] = np.where(rng.integers(0, 2, (h, w // 4)) > 0, max_v - rng.integers(0, 5, (h, w // 4)), rng.integers(0, 5, (h, w // 4)))
        y = y.clip(0, max_v).astype(np.uint8 if bd == 8 else np.uint16)
        tct = np.full(52, edge, np.uint8)
        bt = np.full(52, 255, np.uint8)
        for qp in (20, 51):
            want = oracle.filter_plane(y, qp, bit_depth=bd, tc_table=tct, beta_table=bt)
            for packed in (0, 1):
                got = run_sim(sim, oracle, y, qp, bit_depth=bd, packed=packed, tc_table=tct, beta_table=bt)
                assert np.array_equal(got, want), (bd, qp, packed)
