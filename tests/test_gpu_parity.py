"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libhevcdbk.so), against the oracle on the same seeded inputs and against the golden vectors the
reference produced.  Bit-exact is the only accepted tolerance (8/16-bit integer samples)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, sha256

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gpu_video_codec_amd import deblock
    c = deblock.Context(0)   # raises (loudly) when the library or the device is missing
    yield c
    c.close()


def variants(ctx):
    """Kernel variants that accept a plain 8-bit scalar-QP luma plane on this build."""
    from gpu_video_codec_amd import _lib, deblock
    out = [_lib.KERNEL_GENERIC]
    b = deblock.DeviceBatch(ctx, 16, 16, 1)
    try:
        ctx.filter_device(b.planes(), 30, variant=_lib.KERNEL_PACKED)
        ctx.synchronize()
        out.append(_lib.KERNEL_PACKED)
    except deblock.DeblockError as e:
        assert e.code == _lib.ERR_UNSUPPORTED
    b.free()
    return out


def run_batch(ctx, planes_np, qp, *, variant, is_chroma=False, bit_depth=8, bs=None, qp_map=None, in_place=False,
              tc_table=None, beta_table=None):
    from gpu_video_codec_amd import deblock
    a = np.ascontiguousarray(planes_np)
    n, h, w = a.shape
    b = deblock.DeviceBatch(ctx, w, h, n, bit_depth=bit_depth, sample_bytes=a.itemsize, is_chroma=is_chroma, in_place=in_place)
    b.upload_all(a)
    if bs is not None:
        for f in range(n):
            b.set_bs(f, bs[f][0], bs[f][1])
    if qp_map is not None:
        b.set_qp_map(qp_map)
    ctx.filter_device(b.planes(), qp, variant=variant, tc_table=tc_table, beta_table=beta_table)
    ctx.synchronize()
    out = np.stack([b.download_frame(f) for f in range(n)])
    if not in_place:  # src must be untouched
        assert np.array_equal(np.stack([b.download_frame(f, "src") for f in range(n)]), a)
    b.free()
    return out


# ---- golden vectors from the reference ------------------------------------------------------------

def test_host_frame_operator_golden_manifest(ctx, oracle, manifest, golden_inputs):
    """hevc_deblocking_filter(frame, bS, QP, tables) on the three bundled inputs: every manifest case."""
    for name, ent in manifest["images"].items():
        w, h = ent["width"], ent["height"]
        for c in ent["cases"]:
            y, u, v = (p.copy() for p in oracle.split_yuv420(golden_inputs[name], w, h))
            vb = hb = None
            if c["bs_seed"] is not None:
                vb, hb = oracle.lcg_bs(w, h, c["bs_seed"])
            ctx.filter_frame(y, u, v, qp=c["qp"], vert_bs=vb, hor_bs=hb)
            out = oracle.join_yuv420(y, u, v)
            assert sha256(out[: w * h]) == c["luma_sha256"], (name, c)
            assert sha256(out) == c["sha256"], (name, c)


def test_config2_mother_daughter_full_file(ctx, oracle, golden_inputs):
    """BASELINE config 2: Y+U+V, pinned async copies, byte-for-byte against the reference output."""
    with open(os.path.join(GOLDEN, "mother-daughter_qp35.ref.yuv"), "rb") as fh:
        want = fh.read()
    y, u, v = (p.copy() for p in oracle.split_yuv420(golden_inputs["mother-daughter"], 352, 288))
    t = ctx.filter_frame(y, u, v, qp=35)
    assert oracle.join_yuv420(y, u, v) == want
    # small frames: the kernel reads and writes the page-locked staging buffer itself, so there is no copy phase
    assert t["exec_s"] > 0 and t["copy_s"] >= 0 and abs(t["total_s"] - t["exec_s"] - t["copy_s"]) < 1e-9


def test_golden_synth_4k_frame(ctx, oracle, manifest):
    """The seeded synthetic 4K 4:2:0 frame whose hash the reference itself produced."""
    from gpu_video_codec_amd import synth
    for c in manifest["synth"]:
        w, h = c["width"], c["height"]
        y, u, v = synth.blocky_yuv420(w, h, seed=c["seed"], frame=c["frame"])
        vb = hb = None
        if c["bs_seed"] is not None:
            vb, hb = oracle.lcg_bs(w, h, c["bs_seed"])
        y, u, v = y.copy(), u.copy(), v.copy()
        ctx.filter_frame(y, u, v, qp=c["qp"], vert_bs=vb, hor_bs=hb)
        assert sha256(oracle.join_yuv420(y, u, v)) == c["sha256"], c


# ---- device-resident operator vs oracle ---------------------------------------------------------------

@pytest.mark.parametrize("size", [(8, 8), (16, 8), (8, 16), (24, 40), (504, 8), (512, 16), (520, 24), (1032, 64), (352, 288)])
def test_device_luma_sizes_ragged_and_tiny(ctx, oracle, size):
    from gpu_video_codec_amd import synth
    w, h = size
    rng = np.random.default_rng(w * 131 + h)
    for variant in variants(ctx):
        frames, bss = [], []
        for f in range(3):
            y = synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30))).copy()
            if f == 1:
                y[:] = rng.integers(0, 256, (h, w), dtype=np.uint8)
            if f == 2:
                y[h // 2:, :] = 255
                y[:, : max(w // 4, 1)] = 0
            frames.append(y)
            bss.append(oracle.lcg_bs(w, h, 10 + f) if f else oracle.default_bs(w, h))
        for qp in (27, 37, 51):
            got = run_batch(ctx, np.stack(frames), qp, variant=variant, bs=bss)
            for f in range(3):
                want = oracle.filter_plane(frames[f], qp, vert_bs=bss[f][0], hor_bs=bss[f][1])
                assert np.array_equal(got[f], want), (size, qp, f, variant)


def test_device_chroma_planes(ctx, oracle, golden_inputs):
    from gpu_video_codec_amd import _lib
    _, u, v = oracle.split_yuv420(golden_inputs["image2"], 768, 576)
    for variant in variants(ctx):
        for qp in (30, 45):
            got = run_batch(ctx, np.stack([u, v]), qp, variant=variant, is_chroma=True)
            assert np.array_equal(got[0], oracle.filter_plane(u, qp, is_chroma=True))
            assert np.array_equal(got[1], oracle.filter_plane(v, qp, is_chroma=True))
        # chroma bS override (extension): bS 1 must be skipped, shifted read of SURVEY Q9(ii) reproduced
        vb, hb = oracle.lcg_bs(384, 288, 3)
        got = run_batch(ctx, np.stack([u]), 37, variant=variant, is_chroma=True, bs=[(vb, hb)])
        assert np.array_equal(got[0], oracle.filter_plane(u, 37, is_chroma=True, vert_bs=vb, hor_bs=hb))


def test_device_in_place_equals_out_of_place(ctx, oracle, golden_inputs):
    y, _, _ = oracle.split_yuv420(golden_inputs["image2"], 768, 576)
    for variant in variants(ctx):
        a = run_batch(ctx, np.stack([y, y[::-1]]), 33, variant=variant, in_place=False)
        b = run_batch(ctx, np.stack([y, y[::-1]]), 33, variant=variant, in_place=True)
        assert np.array_equal(a, b)
        assert np.array_equal(a[0], oracle.filter_plane(y, 33))


def test_device_10bit_luma_and_chroma(ctx, oracle):
    """BASELINE config 5 arithmetic (16-bit containers).  Parity vs the oracle's uint16 instantiation,
    itself pinned at 8 bit only ("parity unpinned" beyond that: the reference is 8-bit)."""
    from gpu_video_codec_amd import synth, _lib
    y = synth.blocky_plane(1928, 264, seed=4, bit_depth=10).copy()
    y[:64, :256] = np.random.default_rng(2).integers(0, 1024, (64, 256), dtype=np.uint16)
    y[200:, 1500:] = 1023
    vb, hb = oracle.lcg_bs(1928, 264, 77)
    for variant in (_lib.KERNEL_GENERIC, _lib.KERNEL_PACKED, _lib.KERNEL_AUTO):  # packed = dbk_packed16_kernel
        for qp in (22, 32, 51):
            got = run_batch(ctx, np.stack([y, y[::-1]]), qp, variant=variant, bit_depth=10, bs=[(vb, hb), oracle.default_bs(1928, 264)])
            assert np.array_equal(got[0], oracle.filter_plane(y, qp, bit_depth=10, vert_bs=vb, hor_bs=hb)), (variant, qp)
            assert np.array_equal(got[1], oracle.filter_plane(y[::-1], qp, bit_depth=10)), (variant, qp)
    for size in [(8, 8), (16, 24), (504, 16), (1032, 40)]:  # tiny / ragged rows through the 16-bit packed kernel
        t = synth.blocky_plane(size[0], size[1], seed=9, bit_depth=11)
        got = run_batch(ctx, t[None], 40, variant=_lib.KERNEL_PACKED, bit_depth=11)
        assert np.array_equal(got[0], oracle.filter_plane(t, 40, bit_depth=11)), size
    # 12 bit: the WIDE variant of the packed core (sums beyond int16); uniform noise, blocky content, saturated samples
    t = np.random.default_rng(5).integers(0, 4096, (64, 256), dtype=np.uint16)
    t2 = synth.blocky_plane(1032, 72, seed=12, bit_depth=12).copy()
    t2[40:, 512:] = 4095
    t2[:16, :64] = 0
    for plane in (t, t2):
        for qp in (30, 42, 51):
            for variant in (_lib.KERNEL_AUTO, _lib.KERNEL_PACKED, _lib.KERNEL_GENERIC):
                got = run_batch(ctx, plane[None], qp, variant=variant, bit_depth=12)
                assert np.array_equal(got[0], oracle.filter_plane(plane, qp, bit_depth=12)), (plane.shape, qp, variant)
    # 13 bit and up: AUTO takes the 32-bit kernel, PACKED refuses
    t = np.random.default_rng(6).integers(0, 8192, (64, 256), dtype=np.uint16)
    got = run_batch(ctx, t[None], 51, variant=_lib.KERNEL_AUTO, bit_depth=13)
    assert np.array_equal(got[0], oracle.filter_plane(t, 51, bit_depth=13))
    from gpu_video_codec_amd import deblock
    with pytest.raises(deblock.DeblockError) as e:
        run_batch(ctx, t[None], 51, variant=_lib.KERNEL_PACKED, bit_depth=13)
    assert e.value.code == _lib.ERR_UNSUPPORTED
    got = run_batch(ctx, y[None], 40, variant=_lib.KERNEL_AUTO, bit_depth=10, is_chroma=True)
    assert np.array_equal(got[0], oracle.filter_plane(y, 40, bit_depth=10, is_chroma=True))
    # 8-bit data in 16-bit containers == the reference-pinned 8-bit result
    y8 = synth.blocky_plane(640, 128, seed=5)
    got = run_batch(ctx, y8.astype(np.uint16)[None], 35, variant=_lib.KERNEL_AUTO, bit_depth=8)
    assert np.array_equal(got[0], oracle.filter_plane(y8, 35).astype(np.uint16))


def test_config3_image2_random_bs_and_ctu_qp_map(ctx, oracle, golden_inputs):
    """BASELINE config 3: image2 768x576, LCG bS in {0,1,2}; (3a) scalar QP is reference-pinned,
    (3b) per-CTU QP map is pinned by the oracle restatement only ("parity unpinned")."""
    from gpu_video_codec_amd import synth, _lib
    y, u, v = oracle.split_yuv420(golden_inputs["image2"], 768, 576)
    vb, hb = oracle.lcg_bs(768, 576, 2024)
    for qp in (30, 37):
        for variant in variants(ctx):
            got = run_batch(ctx, y[None], qp, variant=variant, bs=[(vb, hb)])
            assert np.array_equal(got[0], oracle.filter_plane(y, qp, vert_bs=vb, hor_bs=hb))
    for lo, hi, seed in ((22, 42, 7), (0, 51, 8)):
        qmap = synth.ctu_qp_map(768, 576, seed=seed, lo=lo, hi=hi)
        for variant in (_lib.KERNEL_GENERIC, _lib.KERNEL_PACKED, _lib.KERNEL_AUTO):  # per-lane tc/beta in both kernels
            got = run_batch(ctx, y[None], 0, variant=variant, bs=[(vb, hb)], qp_map=qmap)
            assert np.array_equal(got[0], oracle.filter_plane(y, 0, vert_bs=vb, hor_bs=hb, qp_map=qmap)), (variant, lo)
            got = run_batch(ctx, u[None], 0, variant=variant, is_chroma=True, qp_map=qmap)
            assert np.array_equal(got[0], oracle.filter_plane(u, 0, is_chroma=True, qp_map=qmap)), (variant, lo)
        y10 = synth.blocky_plane(768, 576, seed=11, bit_depth=10)
        got = run_batch(ctx, y10[None], 0, variant=_lib.KERNEL_PACKED, bit_depth=10, qp_map=qmap)
        assert np.array_equal(got[0], oracle.filter_plane(y10, 0, bit_depth=10, qp_map=qmap)), lo
    qmap = synth.ctu_qp_map(768, 576, seed=7)
    # constant map == scalar QP (the pinned degenerate case)
    got = run_batch(ctx, y[None], 0, variant=_lib.KERNEL_AUTO, qp_map=np.full((9, 12), 37, np.uint8))
    assert np.array_equal(got[0], oracle.filter_plane(y, 37))
    # host-frame operator with a map
    yy, uu, vv = y.copy(), u.copy(), v.copy()
    ctx.filter_frame(yy, uu, vv, qp=0, qp_map=qmap, vert_bs=vb, hor_bs=hb)
    assert np.array_equal(yy, oracle.filter_plane(y, 0, vert_bs=vb, hor_bs=hb, qp_map=qmap))
    assert np.array_equal(uu, oracle.filter_plane(u, 0, is_chroma=True, qp_map=qmap))


def test_qp_map_with_lane_varying_tc_on_flat_content(ctx, oracle):
    """with a per-CTU QP map the strong filter's clip range (2*tc) differs from lane to lane inside one wave,
    and its three-operand add takes -c as a VECTOR operand there (an SGPR constraint on a divergent value could be legalised
    with v_readfirstlane: lane 0's tc for every lane).  Flat content so that nearly every segment is strong-filtered, an
    8x8 / 16x16 CTU map so that neighbouring lanes hold different QPs, 8 and 10 bit, every kernel family."""
    from gpu_video_codec_amd import deblock, synth, _lib
    rng = np.random.RandomState(31)
    for (w, h, bd, ctu_log2) in ((1024, 64, 8, 3), (1032, 72, 8, 4), (520, 136, 10, 3), (4096, 16, 8, 3)):
        xx, yy = np.meshgrid(np.arange(w), np.arange(h))
        flat = (100 + 40 * np.sin(xx / 180.0) + 0.05 * yy) * (1 << (bd - 8))
        steps = np.repeat(np.repeat(rng.randint(-3, 4, (h // 8, w // 8)), 8, 0), 8, 1) * (1 << (bd - 8))   # small block steps: strong filter
        plane = np.clip(flat + steps, 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16)
        qmap = rng.randint(30, 52, ((h + (1 << ctu_log2) - 1) >> ctu_log2, (w + (1 << ctu_log2) - 1) >> ctu_log2)).astype(np.uint8)
        want = oracle.filter_plane(plane, 0, bit_depth=bd, qp_map=qmap, ctu_log2=ctu_log2)
        assert (want != plane).mean() > 0.2   # the content does get filtered
        for variant in (_lib.KERNEL_GENERIC, _lib.KERNEL_PACKED, _lib.KERNEL_AUTO):
            b = deblock.DeviceBatch(ctx, w, h, 1, bit_depth=bd)
            b.upload_all(plane[None])
            b.set_qp_map(qmap, ctu_log2)
            ctx.filter_device(b.planes(), 0, variant=variant)
            ctx.synchronize()
            assert np.array_equal(b.download_frame(0), want), (w, h, bd, ctu_log2, variant)
            b.qp_map.free()
            b.free()


def test_qp_map_regions_equal_scalar_launches_at_4k(ctx):
    """A size-independent property at BASELINE's frame size that ties the QP-map kernels to the scalar-QP kernels (which the
    reference-made hashes pin): with a map that is constant over the four quadrants of a 3840x2160 picture, every sample further
    than 8 samples from a quadrant border must come out exactly as a launch with that quadrant's ONE QP makes it (no decision
    that reaches it sees another QP: tests/test_oracle.py measures 4 samples) -- 64- and 16-sample units, default and seeded bS,
    8 and 10 bit, packed and generic kernels."""
    from gpu_video_codec_amd import deblock, synth, _lib
    w, h = 3840, 2160
    rng = np.random.RandomState(5)
    for (bd, log2, variant, seeded) in ((8, 6, _lib.KERNEL_AUTO, False), (8, 4, _lib.KERNEL_PACKED, True), (10, 6, _lib.KERNEL_AUTO, True),
                                        (8, 6, _lib.KERNEL_GENERIC, True)):
        unit = 1 << log2
        rows, cols = (h + unit - 1) // unit, (w + unit - 1) // unit
        by, bx = (rows // 2) * unit, (cols // 2) * unit
        qps = {(0, 0): 27, (0, 1): 39, (1, 0): 46, (1, 1): 33}
        qmap = np.zeros((rows, cols), np.uint8)
        for (ry, rx), q in qps.items():
            qmap[(0 if ry == 0 else rows // 2):(rows // 2 if ry == 0 else rows), (0 if rx == 0 else cols // 2):(cols // 2 if rx == 0 else cols)] = q
        plane = synth.blocky_plane(w, h, seed=17 + bd, bit_depth=bd)
        b = deblock.DeviceBatch(ctx, w, h, 1, bit_depth=bd)
        b.upload_all(plane[None])
        if seeded:
            b.set_bs(0, rng.randint(0, 3, b.nv).astype(np.uint8), rng.randint(0, 3, b.nh).astype(np.uint8))
        b.set_qp_map(qmap, log2)
        ctx.filter_device(b.planes(), 0, variant=variant)
        ctx.synchronize()
        got = b.download_frame(0)
        b.qp_map.free()
        b.qp_map = None
        p = b.planes()  # the same planes and bS arrays, no map: one QP per launch
        m = 8
        assert (got != plane).mean() > 0.05
        for (ry, rx), q in qps.items():
            ctx.filter_device(p, q, variant=variant)
            ctx.synchronize()
            want = b.download_frame(0)
            y0, y1 = (0, by - m) if ry == 0 else (by + m, h)
            x0, x1 = (0, bx - m) if rx == 0 else (bx + m, w)
            assert np.array_equal(got[y0:y1, x0:x1], want[y0:y1, x0:x1]), (bd, log2, variant, seeded, ry, rx)
            assert not np.array_equal(got, want)
        b.free()


def test_custom_tables(ctx, oracle, golden_inputs):
    y, _, _ = oracle.split_yuv420(golden_inputs["image1"], 352, 288)
    tc, beta = oracle.tables()
    tc2, beta2 = np.minimum(tc + 2, 255), np.minimum(beta + 9, 255)
    for variant in variants(ctx):
        got = run_batch(ctx, y[None], 33, variant=variant, tc_table=tc2, beta_table=beta2)
        assert np.array_equal(got[0], oracle.filter_plane(y, 33, tc_table=tc2, beta_table=beta2))


def test_maximal_custom_tables_at_10_11_12_bit(ctx, oracle):
    """Caller tables with entries up to 255 scale to tc values the packed core's 16-bit fields cannot hold at 11 and 12
    bit (8*max_v + 4 + 16*tc, 10*tc): HEVCDBK_KERNEL_AUTO must then take the 32-bit kernel (same bytes as the oracle) and
    an explicit HEVCDBK_KERNEL_PACKED must be refused; inside the range -- up to its very edge -- the packed kernel runs
    and is bit-exact.  Scalar QP and QP map (the map case is bounded by the largest table entry)."""
    from gpu_video_codec_amd import deblock, _lib
    rng = np.random.default_rng(77)
    for bd, entry, fits in ((8, 255, True), (10, 255, True), (11, 128, True), (11, 255, False), (12, 128, True), (12, 129, False),
                            (12, 255, False)):
        max_v, shift = (1 << bd) - 1, bd - 8
        h, w = 72, 520
        y = np.full((h, w), max_v // 2, np.int64)
        y[:, : w // 4] += rng.integers(-3, 4, (h, w // 4)) << shift
        y[:, w // 4: w // 2] = rng.integers(0, max_v + 1, (h, w // 4))
        y[:, w // 2:] = np.where(rng.integers(0, 2, (h, w - w // 2)) > 0, max_v - rng.integers(0, 5, (h, w - w // 2)),
                                 rng.integers(0, 5, (h, w - w // 2)))
        y = y.clip(0, max_v).astype(np.uint8 if bd == 8 else np.uint16)
        tct, bt = np.full(52, entry, np.uint8), np.full(52, 255, np.uint8)
        want = oracle.filter_plane(y, 40, bit_depth=bd, tc_table=tct, beta_table=bt)
        for variant in (_lib.KERNEL_AUTO, _lib.KERNEL_GENERIC):
            got = run_batch(ctx, y[None], 40, variant=variant, bit_depth=bd, tc_table=tct, beta_table=bt)
            assert np.array_equal(got[0], want), (bd, entry, variant)
        if fits:
            got = run_batch(ctx, y[None], 40, variant=_lib.KERNEL_PACKED, bit_depth=bd, tc_table=tct, beta_table=bt)
            assert np.array_equal(got[0], want), (bd, entry, "packed")
        else:
            with pytest.raises(deblock.DeblockError) as e:
                run_batch(ctx, y[None], 40, variant=_lib.KERNEL_PACKED, bit_depth=bd, tc_table=tct, beta_table=bt)
            assert e.value.code == _lib.ERR_UNSUPPORTED
        # QP map: a table whose entry AT THE MAP'S QPs is small but whose largest entry is out of range still goes 32-bit
        tcm = np.full(52, 4, np.uint8)
        tcm[51] = entry
        qmap = np.full(((h + 63) // 64, (w + 63) // 64), 30, np.uint8)
        wantm = oracle.filter_plane(y, 30, bit_depth=bd, tc_table=tcm, beta_table=bt, qp_map=qmap)
        got = run_batch(ctx, y[None], 30, variant=_lib.KERNEL_AUTO, bit_depth=bd, tc_table=tcm, beta_table=bt, qp_map=qmap)
        assert np.array_equal(got[0], wantm), (bd, entry, "map")


# ---- properties at BASELINE's full sizes ------------------------------------------------------------

def test_4k_batch_properties(ctx, oracle):
    """3840x2160: frames of a batch are independent, QP <= 17 is the identity, QP >= 52 == 51,
    and one frame is compared with the oracle in full."""
    from gpu_video_codec_amd import synth
    base = synth.blocky_plane(3840, 2160, seed=1)
    frames = np.stack([base, np.roll(base, (8, 16), (0, 1)), base[::-1]])
    for variant in variants(ctx):
        got = run_batch(ctx, frames, 32, variant=variant)
        assert np.array_equal(got[0], oracle.filter_plane(base, 32, threads=8))
        single = run_batch(ctx, frames[2:3], 32, variant=variant)
        assert np.array_equal(single[0], got[2])
        assert np.array_equal(run_batch(ctx, frames[:1], 17, variant=variant)[0], base)
        assert np.array_equal(run_batch(ctx, frames[:1], 99, variant=variant), run_batch(ctx, frames[:1], 51, variant=variant))
        # with a position-independent bS (all 2; the default pattern has the Q3 zeros) a circular shift
        # by whole 8x8 blocks commutes with the filter away from the frame border and the wrap seam
        vb2 = np.full(oracle.num_vert_bs(3840, 2160), 2, np.uint8)
        hb2 = np.full(oracle.num_hor_bs(3840, 2160), 2, np.uint8)
        a0 = run_batch(ctx, frames[0:1], 32, variant=variant, bs=[(vb2, hb2)])[0]
        sh = run_batch(ctx, frames[1:2], 32, variant=variant, bs=[(vb2, hb2)])[0]
        assert np.array_equal(sh[16:-16, 24:-24], np.roll(a0, (8, 16), (0, 1))[16:-16, 24:-24])


def test_8k_10bit_frame(ctx, oracle):
    """BASELINE config 5 geometry (7680x4320, 10-bit in 16-bit containers), one frame, full compare."""
    from gpu_video_codec_amd import synth, _lib
    y = synth.blocky_plane(7680, 4320, seed=2, bit_depth=10)
    got = run_batch(ctx, y[None], 32, variant=_lib.KERNEL_AUTO, bit_depth=10)
    assert np.array_equal(got[0], oracle.filter_plane(y, 32, bit_depth=10, threads=8))


# ---- reference-shaped interface and error behaviour ---------------------------------------------------

def test_read_yuv_frame_mirror_and_execute_gpu(ctx, oracle, manifest, tmp_path, capfd):
    from gpu_video_codec_amd import deblock, _lib
    src = os.path.join(GOLDEN, "image1_352x288_yv12.yuv")
    f = deblock.ReadYuvFrame(src, 352, 288, 30, ctx=ctx)
    f.DeblockingFilter()
    out = tmp_path / "image1_filtered.yuv"
    f.Save(str(out))
    want = [c for c in manifest["images"]["image1"]["cases"] if c["qp"] == 30 and c["bs_seed"] is None][0]
    assert sha256(out.read_bytes()) == want["sha256"]
    f2 = deblock.ReadYuvFrame(src, 352, 288, 37, ctx=ctx)
    vb, hb = oracle.lcg_bs(352, 288, 1)
    f2.SetBoundaryStrenght(vb, hb)
    f2.DeblockingFilter()
    want = [c for c in manifest["images"]["image1"]["cases"] if c["qp"] == 37 and c["bs_seed"] == 1][0]
    assert sha256(f2.tobytes()) == want["sha256"]
    # ExecuteGpu-equivalent: file in -> file out + the reference's three console lines (gpu.cu:1292,1302-1303)
    out2 = tmp_path / "eg.yuv"
    rc = _lib.lib().hevcdbk_execute_gpu(src.encode(), str(out2).encode(), 352, 288, 30, 20, 20, 20, 20, 0)
    assert rc == 0
    want = [c for c in manifest["images"]["image1"]["cases"] if c["qp"] == 30 and c["bs_seed"] is None][0]
    assert sha256(out2.read_bytes()) == want["sha256"]
    txt = capfd.readouterr().out
    for line in ("Execution Time without copy on GPU:", "Execution Time with copy on GPU:", "Copy Operation Time with GPU buffers:"):
        assert line in txt


def test_driver_binary(tmp_path):
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "gpu_video_codec_amd", "hevc_deblock_main")
    src = os.path.join(GOLDEN, "mother-daughter_352x288_yv12.yuv")
    out = tmp_path / "md.yuv"
    r = subprocess.run([exe, src, str(out), "352", "288", "35"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Execution Time without copy on GPU:" in r.stdout
    with open(os.path.join(GOLDEN, "mother-daughter_qp35.ref.yuv"), "rb") as fh:
        assert out.read_bytes() == fh.read()


def test_error_codes_on_device(ctx, oracle):
    from gpu_video_codec_amd import deblock, _lib
    y = np.zeros((16, 16), np.uint8)
    with pytest.raises(deblock.DeblockError) as e:
        ctx.filter_frame(y, qp=30, vert_bs=np.zeros(3, np.uint8), hor_bs=np.zeros(4, np.uint8))
    assert e.value.code == _lib.ERR_BS_SIZE
    with pytest.raises(deblock.DeblockError) as e:
        ctx.filter_frame(np.zeros((20, 20), np.uint8), qp=30)
    assert e.value.code == _lib.ERR_DIMENSIONS
    # empty / degenerate batch: zero frames is a no-op, not an error
    b = deblock.DeviceBatch(ctx, 16, 16, 1)
    p = b.planes()
    p.n_frames = 0
    ctx.filter_device(p, 30)
    ctx.synchronize()
    b.free()


def test_replay_settles_warms_up_and_times(ctx, oracle):
    """hevcdbk_device_replay (what bench.py times with): settle by time, warm-up and timed launches as one stream; the
    per-launch times, the wall clock of the timed window and the GPU-clock span agree, and dst holds the filtered frames."""
    from gpu_video_codec_amd import deblock, synth, _lib
    y = np.stack([synth.blocky_plane(352, 288, seed=5, frame=f) for f in range(3)])
    b = deblock.DeviceBatch(ctx, 352, 288, 3)
    b.upload_all(y)
    ms, info = ctx.replay([b.planes()], 30, 7, warmup=2, settle_min_ms=5.0, settle_max_ms=60.0)
    assert ms.shape == (7,) and np.all(ms > 0) and np.all(ms < 50)
    assert info["settle_launches"] >= 64 and 5.0 <= info["settle_ms"] <= 500.0 and info["settle_tail_mean_ms"] > 0
    assert info["t_end"] > info["t_begin"] and abs(info["wall_ms"] - (info["t_end"] - info["t_begin"]) * 1e3) < 1e-6
    assert info["span_ms"] >= ms.sum() * 0.999 and info["span_ms"] < 50        # launches of one stream do not overlap
    # the host's view of the same window: it starts when the host has SEEN the last warm-up launch end (launches this short
    # are partly over by then) and ends with the synchronisation, so it can be shorter than the GPU-clock span but not much longer
    assert 0 < info["wall_ms"] < info["span_ms"] + 5.0
    for f in range(3):
        assert np.array_equal(b.download_frame(f), oracle.filter_plane(y[f], 30))
    # no settling, no warm-up = hevcdbk_device_run_timed
    ms2 = ctx.run_timed([b.planes()], 30, 4)
    ms3, info3 = ctx.replay([b.planes()], 30, 4, warmup=0, settle_min_ms=0.0, settle_max_ms=0.0)
    assert ms2.shape == ms3.shape == (4,) and info3["settle_launches"] == 0 and not info3["settled"]
    # a fixed settling time (what the ranks of a multi-GPU run do after their barrier) and zero timed steps
    _none, info4 = ctx.replay([b.planes()], 30, 0, warmup=0, settle_min_ms=20.0, settle_max_ms=20.0)
    assert 20.0 <= info4["settle_ms"] < 200.0 and info4["settle_launches"] > 0
    # an operand the forced kernel family does not take fails before ANYTHING is enqueued (plane 0 used to be
    # queued -- dst partly written -- when plane 1 was refused): 14-bit chroma is beyond the packed kernels
    u = deblock.DeviceBatch(ctx, 176, 144, 3, is_chroma=True, bit_depth=14)
    u.upload_all(np.stack([synth.blocky_plane(176, 144, seed=9, frame=f, bit_depth=14) for f in range(3)]))
    marker = np.full(b.frame_bytes * 3, 0xA5, np.uint8)
    b.dst.upload(marker)
    with pytest.raises(deblock.DeblockError) as e:
        ctx.filter_device_planes([b.planes(), u.planes()], 30, variant=_lib.KERNEL_PACKED)
    assert e.value.code == _lib.ERR_UNSUPPORTED
    ctx.synchronize()
    assert np.all(b.download_frame(0) == 0xA5) and np.all(b.download_frame(2) == 0xA5)
    ctx.filter_device_planes([b.planes(), u.planes()], 30)    # AUTO: plane by plane, packed luma + 32-bit chroma kernel
    ctx.synchronize()
    assert np.array_equal(b.download_frame(1), oracle.filter_plane(y[1], 30))
    assert ctx.pci_bus_id().count(":") == 2
    b.free()
    u.free()


def test_row_major_mapping_forced(ctx, oracle):
    """The row-major (linear) block mapping is chosen automatically only for rows wider than one
    workgroup (8K); force it (HEVCDBK_MAP_LINEAR of the kernel selector) on small, ragged and multi-frame
    geometries as well, and the row map (HEVCDBK_MAP_ROWS) on the same ones: same bytes."""
    from gpu_video_codec_amd import synth, _lib
    rng = np.random.default_rng(99)
    for mapping in (_lib.MAP_LINEAR, _lib.MAP_ROWS):
        variant = _lib.KERNEL_PACKED | mapping
        for (w, h) in [(352, 288), (520, 136), (1032, 72), (3840, 64), (4104, 40)]:
            frames = np.stack([synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30))) for _ in range(3)])
            frames[1, : h // 2, : w // 3] = rng.integers(0, 256, (h // 2, w // 3), dtype=np.uint8)
            bss = [oracle.lcg_bs(w, h, 5), oracle.default_bs(w, h), oracle.lcg_bs(w, h, 6)]
            got = run_batch(ctx, frames, 37, variant=variant, bs=bss)
            for f in range(3):
                assert np.array_equal(got[f], oracle.filter_plane(frames[f], 37, vert_bs=bss[f][0], hor_bs=bss[f][1])), (w, h, f)
            f10 = np.stack([synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30)), bit_depth=10) for _ in range(2)])
            got = run_batch(ctx, f10, 32, variant=variant, bit_depth=10)
            for f in range(2):
                assert np.array_equal(got[f], oracle.filter_plane(f10[f], 32, bit_depth=10)), (w, h, f)
            c = frames[:2]
            got = run_batch(ctx, c, 40, variant=variant, is_chroma=True)
            for f in range(2):
                assert np.array_equal(got[f], oracle.filter_plane(c[f], 40, is_chroma=True)), (w, h, f)
    # an unknown map selector is refused, not ignored
    from gpu_video_codec_amd import deblock
    b = deblock.DeviceBatch(ctx, 64, 64, 1)
    with pytest.raises(deblock.DeblockError) as e:
        ctx.filter_device(b.planes(), 30, variant=_lib.KERNEL_PACKED | 0x300)
    assert e.value.code == _lib.ERR_ARG
    b.free()


def test_device_planes_one_call_large_frames(ctx, oracle):
    """hevc_deblocking_filter_device_planes: Y, U, V of a batch of LARGE 4:2:0 frames in one call (one fused launch for
    scalar-QP planes of one bit depth, 8-bit or 16-bit containers up to 12 bit, SURVEY 8f rank 1, no frame-size gate), and the
    operands the fused kernels do not take (a QP map) through the same entry, plane by plane: same bytes as the oracle plane by
    plane.  10 / 11 bit = the plain 16-bit core, 12 bit = its WIDE luma variant; rows narrower and wider than a workgroup."""
    from gpu_video_codec_amd import deblock, synth, _lib
    for (w, h, n, bd) in [(3840, 144, 2, 8), (1920, 1088, 2, 8), (7680, 48, 1, 8), (1280, 80, 3, 10), (3840, 144, 2, 10),
                          (16, 16, 2, 10), (8208, 32, 1, 11), (1936, 80, 2, 12)]:
        ys = np.stack([synth.blocky_plane(w, h, seed=10 + f, bit_depth=bd) for f in range(n)])
        us = np.stack([synth.blocky_plane(w // 2, h // 2, seed=20 + f, bit_depth=bd, dc_range=4) for f in range(n)])
        vs = np.stack([synth.blocky_plane(w // 2, h // 2, seed=30 + f, bit_depth=bd, dc_range=4) for f in range(n)])
        for variant in (_lib.KERNEL_AUTO, _lib.KERNEL_GENERIC):
            bat = []
            for a, ch in ((ys, False), (us, True), (vs, True)):
                b = deblock.DeviceBatch(ctx, a.shape[2], a.shape[1], n, bit_depth=bd, is_chroma=ch)
                b.upload_all(a)
                bat.append(b)
            ctx.filter_device_planes([b.planes() for b in bat], 35, variant=variant)
            ctx.synchronize()
            for b, a, ch in zip(bat, (ys, us, vs), (False, True, True)):
                for f in range(n):
                    assert np.array_equal(b.download_frame(f), oracle.filter_plane(a[f], 35, is_chroma=ch, bit_depth=bd)), (w, h, bd, variant, ch, f)
                b.free()
    # a QP map on the luma plane: not fusable, still one call
    w, h = 1920, 144
    y = synth.blocky_plane(w, h, seed=3)
    u = synth.blocky_plane(w // 2, h // 2, seed=4, dc_range=4)
    qmap = np.random.default_rng(5).integers(22, 45, ((h + 63) // 64, (w + 63) // 64)).astype(np.uint8)
    by = deblock.DeviceBatch(ctx, w, h, 1)
    by.upload_all(y[None])
    by.set_qp_map(qmap)
    bu = deblock.DeviceBatch(ctx, w // 2, h // 2, 1, is_chroma=True)
    bu.upload_all(u[None])
    bu.set_qp_map(qmap)
    ctx.filter_device_planes([by.planes(), bu.planes()], 30)
    ctx.synchronize()
    assert np.array_equal(by.download_frame(0), oracle.filter_plane(y, 30, qp_map=qmap))
    assert np.array_equal(bu.download_frame(0), oracle.filter_plane(u, 30, qp_map=qmap, is_chroma=True))
    by.free()
    bu.free()
    # argument errors
    with pytest.raises(deblock.DeblockError) as e:
        ctx.filter_device_planes([], 30)
    assert e.value.code == _lib.ERR_ARG


_DIAG_CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from gpu_video_codec_amd import _lib, deblock, synth
from oracle import oracle
_lib.use_diagnostic_library(%(spec)r)
sys.path.insert(0, %(tests)r)
from test_gpu_parity import run_batch
flat = np.full((96, 4096), 100, np.uint8)
flat[:, 2048:] += 4          # mild steps on every 8x8 edge: strong filter everywhere -> the queue overflows
flat[::16] += 2
frames = {"synth": synth.blocky_plane(3840, 72, seed=8), "flat": flat,
          "narrow": synth.blocky_plane(200, 64, seed=9), "noise": np.random.default_rng(1).integers(0, 256, (64, 1032), dtype=np.uint8)}
with deblock.Context(0) as ctx:
    for name, y in frames.items():
        for qp in (32, 45):
            got = run_batch(ctx, np.stack([y, y[::-1]]), qp, variant=_lib.KERNEL_PACKED)
            assert np.array_equal(got[0], oracle.filter_plane(y, qp)), (name, qp)
            assert np.array_equal(got[1], oracle.filter_plane(y[::-1], qp)), (name, qp)
    # the copy variant exists here (and only here): dst = src
    y = frames["synth"]
    got = run_batch(ctx, np.stack([y]), 32, variant=_lib.DIAG_KERNEL_COPY)
    assert np.array_equal(got[0], y)
    # the stripe map (HEVCDBK_DIAG_MAP_STRIPE): persistent waves, next tile and its bS bytes prefetched into LDS by
    # buffer_load ... lds, frame border by extra workgroups of the same launch.  One and several waves per row group, rows
    # per group 1..8, a last group that is not full, fewer items than persistent workgroups and many more, in place,
    # per-frame random bS (every guard), a QP with tc = 0; operands it does not take fall back to the geometry's own map
    rng = np.random.default_rng(4242)
    for variant in (_lib.KERNEL_PACKED | _lib.DIAG_MAP_STRIPE, _lib.KERNEL_PACKED | _lib.DIAG_MAP_TILES,
                    _lib.KERNEL_PACKED | _lib.DIAG_MAP_PIPE, _lib.KERNEL_PACKED | _lib.DIAG_MAP_GROUP):
        # ... the pipe map (HEVCDBK_DIAG_MAP_PIPE): N block rows per workgroup (knob rows=N, default 4), the next row's tile
        # prefetched in registers; and the group map (HEVCDBK_DIAG_MAP_GROUP): k whole block rows minus column 0 per
        # workgroup, no idle lanes, frame border by extra workgroups of the same launch
        # ... and the tile map (HEVCDBK_DIAG_MAP_TILES): whole block rows staged in LDS, blocks filtered out of LDS, column
        # bx = 0 by the second launch; 1..8 block rows per workgroup, a last workgroup that is not full, one-block-row heights
        for (w, h, n) in [(3840, 72, 2), (3840, 136, 3), (1920, 264, 2), (1280, 72, 5), (512, 40, 3), (128, 8, 2), (128, 24, 2),
                          (256, 136, 40), (4096, 48, 1), (7680, 40, 1), (1024, 1032, 3), (384, 16, 3)]:
            fr = np.stack([synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30))) for _ in range(min(n, 4))])
            fr = np.concatenate([fr] * (n // len(fr) + 1))[:n].copy()
            fr[0, : h // 2, : w // 3] = rng.integers(0, 256, (h // 2, w // 3), dtype=np.uint8)
            if n > 1:
                fr[1] = fr[1][::-1]
            bss = [oracle.lcg_bs(w, h, 5 + f) if f %% 2 == 0 else oracle.default_bs(w, h) for f in range(n)]
            for qp, in_place in ((37, False), (32, True), (17, False)):
                got = run_batch(ctx, fr, qp, variant=variant, bs=bss, in_place=in_place)
                for f in sorted({0, 1 %% n, n // 2, n - 1}):
                    assert np.array_equal(got[f], oracle.filter_plane(fr[f], qp, vert_bs=bss[f][0], hor_bs=bss[f][1])), (w, h, n, qp, f)
        y = synth.blocky_plane(520, 72, seed=3)
        assert np.array_equal(run_batch(ctx, y[None], 37, variant=variant)[0], oracle.filter_plane(y, 37))
        c = synth.blocky_plane(1920, 136, seed=4)
        assert np.array_equal(run_batch(ctx, c[None], 40, variant=variant, is_chroma=True)[0], oracle.filter_plane(c, 40, is_chroma=True))
        t = synth.blocky_plane(1920, 72, seed=5, bit_depth=10)
        assert np.array_equal(run_batch(ctx, t[None], 32, variant=variant, bit_depth=10)[0], oracle.filter_plane(t, 32, bit_depth=10))
print("DIAG-OK")
"""


def _run_diag_child(spec):
    import subprocess
    code = _DIAG_CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "spec": spec}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIAG-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_diagnostic_library_variants_are_bit_exact():
    """libhevcdbk_diag.so (same sources, -DHEVCDBK_DIAG) in a process of its own: the LDS-queue kernel (knob "queue":
    strong segments scheduled through a workgroup queue) gives the same bytes, including queue overflow
    (an all-flat frame makes every segment strong) and planes narrower than a workgroup; so do the instrumented
    instantiation with its knobs at rest, other workgroup widths and the wave-priority experiment."""
    _run_diag_child("queue")
    _run_diag_child("mode3,wg=256,prio=3")


def test_stripe_map_barrier_variant():
    """The stripe map once more with its "workgroup barrier before the stores" experiment switched on (knob dummy=16)."""
    _run_diag_child("dummy=16")


def test_pipe_map_row_counts_and_barrier():
    """The pipe map with other row counts per workgroup (3: a last group that is not full; 16: more rows than small planes
    have) and with its "waves of a block row store together" barrier (knob dummy=64)."""
    _run_diag_child("rows=3,dummy=64")
    _run_diag_child("rows=16")


def test_product_library_has_no_diagnostics(ctx):
    """The shipped library refuses the copy selector and exports no diagnostic entry point."""
    from gpu_video_codec_amd import deblock, _lib
    assert not hasattr(_lib.lib(), "hevcdbk_diag_set")
    b = deblock.DeviceBatch(ctx, 64, 64, 1)
    with pytest.raises(deblock.DeblockError) as e:
        ctx.filter_device(b.planes(), 30, variant=_lib.DIAG_KERNEL_COPY)
    assert e.value.code == _lib.ERR_ARG
    b.free()


def test_pitched_planes_and_untouched_row_padding(ctx, oracle):
    """Planes whose pitch exceeds the width (decoder surfaces): same result, and the bytes between the end of a
    row and the pitch are never written (the last offset block's out-of-image half is not stored)."""
    from gpu_video_codec_amd import deblock, synth, _lib
    for (w, h, pitch, bd) in [(520, 72, 576, 8), (1032, 40, 1032 + 24, 8), (352, 288, 512, 8), (520, 72, 2 * 520 + 48, 10)]:
        y = synth.blocky_plane(w, h, seed=w + h, bit_depth=bd)
        for variant in (_lib.KERNEL_GENERIC, _lib.KERNEL_PACKED):
            for in_place in (False, True):
                b = deblock.DeviceBatch(ctx, w, h, 2, bit_depth=bd, pitch=pitch, in_place=in_place)
                b.upload_all(np.stack([y, y[::-1]]), fill=0xA5 if bd == 8 else 0x3A5)
                if not in_place:  # pre-fill dst with a pattern so untouched bytes are recognisable
                    b.dst.upload(np.full(b.frame_bytes * 2, 0x5A, np.uint8))
                ctx.filter_device(b.planes(), 37, variant=variant)
                ctx.synchronize()
                for f, src in enumerate((y, y[::-1])):
                    full = b.download_frame(f, with_padding=True)
                    assert np.array_equal(full[:, :w], oracle.filter_plane(src, 37, bit_depth=bd)), (w, pitch, variant, in_place, f)
                    pad = full[:, w:]
                    want = (0xA5 if bd == 8 else 0x3A5) if in_place else (0x5A if bd == 8 else 0x5A5A)
                    assert (pad == want).all(), (w, pitch, variant, in_place, f)
                b.free()


def test_streaming_sequence_operator(ctx, oracle, golden_inputs):
    """hevc_deblocking_filter_sequence: more frames than pipeline slots, pageable and pinned planes mixed,
    pitched rows, luma-only and Y+U+V; every frame must equal the single-frame result."""
    from gpu_video_codec_amd import synth
    # 4:2:0 sequence, pageable memory, 7 frames > 3 slots
    frames, want = [], []
    for i in range(7):
        y, u, v = synth.blocky_yuv420(352, 288, seed=20 + i)
        want.append(oracle.filter_yuv420(oracle.join_yuv420(y, u, v), 352, 288, 35))
        frames.append((y.copy(), u.copy(), v.copy()))
    t = ctx.filter_sequence(frames, qp=35)
    assert t > 0
    for i, pl in enumerate(frames):
        assert oracle.join_yuv420(*pl) == want[i], i
    # luma-only, pinned (zero-copy) and pageable alternating, with a random luma bS and a padded pitch
    vb, hb = oracle.lcg_bs(768, 576, 31)
    src = [np.roll(oracle.split_yuv420(golden_inputs["image2"], 768, 576)[0], 8 * i, axis=1) for i in range(5)]
    planes, pinned, fulls = [], [], []
    for i, s in enumerate(src):
        if i % 2 == 0:
            buf = ctx.pinned_array((576, 832), np.uint8)
            pinned.append(buf)
        else:
            buf = np.empty((576, 832), np.uint8)
        buf[:] = 0xEE
        view = buf[:, :768]
        view[:] = s
        fulls.append(buf)
        planes.append((view,))
    ctx.filter_sequence(planes, qp=37, vert_bs=vb, hor_bs=hb)
    for i, s in enumerate(src):
        assert np.array_equal(planes[i][0], oracle.filter_plane(s, 37, vert_bs=vb, hor_bs=hb)), i
        assert (fulls[i][:, 768:] == 0xEE).all(), i  # row padding of the caller's planes untouched
    for b in pinned:
        ctx.free_pinned(b)


def test_streaming_sequence_of_large_pageable_frames(ctx, oracle):
    """hevc_deblocking_filter_sequence on frames above 2 MiB that all lie in ordinary pageable memory: the round-4 pipeline (the
    crew writes frame i into slot i % 3 of HBM through the BAR, the kernels store into slot i % 3 of a page-locked ring, the crew
    copies out) -- 1 .. 8 frames (fewer and more than the three slots), luma only and Y+U+V, 8 and 10 bit, pitched rows whose
    padding must stay untouched, a caller bS shared by the frames, 1 and 4 copying threads; every frame against the oracle.  One
    page-locked plane in the sequence sends it down the DMA path instead: same results."""
    from gpu_video_codec_amd import synth
    try:
        for threads in (4, 1):
            ctx.set_host_threads(threads)
            for (w, h, bd, n, chroma, pad) in ((1920, 1088, 8, 8, False, 0), (1920, 1088, 8, 5, True, 32), (2560, 1440, 10, 4, False, 16),
                                               (3840, 2160, 8, 2, False, 0), (1920, 1088, 8, 1, True, 0)):
                dt = np.uint8 if bd == 8 else np.uint16
                vb, hb = oracle.lcg_bs(w, h, 3 + n)
                frames, bufs, want = [], [], []
                for i in range(n):
                    pl = synth.blocky_yuv420(w, h, seed=100 + 7 * i + w, bit_depth=bd) if chroma else (synth.blocky_plane(w, h, seed=100 + 7 * i + w, bit_depth=bd),)
                    want.append([oracle.filter_plane(p, 36, bit_depth=bd, is_chroma=k > 0, vert_bs=vb if k == 0 else None,
                                                     hor_bs=hb if k == 0 else None, threads=8) for k, p in enumerate(pl)])
                    bb = [np.full((p.shape[0], p.shape[1] + pad), 0x3C, dt) for p in pl]
                    for b, p in zip(bb, pl):
                        b[:, :p.shape[1]] = p
                    bufs.append(bb)
                    frames.append(tuple(b[:, :p.shape[1]] for b, p in zip(bb, pl)))
                t = ctx.filter_sequence(frames, qp=36, bit_depth=bd, vert_bs=vb, hor_bs=hb)
                assert t > 0
                for i in range(n):
                    for k in range(len(frames[i])):
                        assert np.array_equal(frames[i][k], want[i][k]), (threads, w, h, bd, n, chroma, i, k)
                        if pad:
                            assert (bufs[i][k][:, -pad:] == 0x3C).all(), (threads, w, h, i, k)
        # one page-locked frame among pageable ones: the DMA pipeline takes the sequence
        w, h = 1920, 1088
        src = [synth.blocky_plane(w, h, seed=300 + i) for i in range(4)]
        pin = ctx.pinned_array((h, w), np.uint8)
        planes = [(src[0].copy(),), (pin,), (src[2].copy(),), (src[3].copy(),)]
        pin[:] = src[1]
        ctx.filter_sequence(planes, qp=31)
        for i in range(4):
            assert np.array_equal(planes[i][0], oracle.filter_plane(src[i], 31, threads=8)), i
        ctx.free_pinned(pin)
    finally:
        ctx.set_host_threads(0)


def test_fused_launch_and_resident_default_bs(ctx, oracle):
    """Small 4:2:0 frames go out as one DMA in, one fused Y+U+V launch, one DMA out, and the default bS stays on
    the device between calls: alternate geometries, default and caller bS, pitched planes, and the streaming
    operator, so that every way the resident copy can go stale is exercised."""
    from gpu_video_codec_amd import synth

    def run(w, h, seed, qp, user_bs, pitched=False):
        y, u, v = synth.blocky_yuv420(w, h, seed=seed)
        kw = {}
        if user_bs:
            kw["vert_bs"], kw["hor_bs"] = oracle.lcg_bs(w, h, seed)
            kw["chroma_vert_bs"], kw["chroma_hor_bs"] = oracle.lcg_bs(w // 2, h // 2, seed + 1)
        want = oracle.join_yuv420(
            oracle.filter_plane(y, qp, vert_bs=kw.get("vert_bs"), hor_bs=kw.get("hor_bs")),
            oracle.filter_plane(u, qp, is_chroma=True, vert_bs=kw.get("chroma_vert_bs"), hor_bs=kw.get("chroma_hor_bs")),
            oracle.filter_plane(v, qp, is_chroma=True, vert_bs=kw.get("chroma_vert_bs"), hor_bs=kw.get("chroma_hor_bs")))
        if pitched:
            bufs = [np.full((p.shape[0], p.shape[1] + 24), 0xC3, np.uint8) for p in (y, u, v)]
            views = [b[:, :p.shape[1]] for b, p in zip(bufs, (y, u, v))]
            for vw, p in zip(views, (y, u, v)):
                vw[:] = p
            ctx.filter_frame(*views, qp=qp, **kw)
            assert all((b[:, -24:] == 0xC3).all() for b in bufs)
            got = oracle.join_yuv420(*[np.ascontiguousarray(vw) for vw in views])
        else:
            y, u, v = y.copy(), u.copy(), v.copy()
            ctx.filter_frame(y, u, v, qp=qp, **kw)
            got = oracle.join_yuv420(y, u, v)
        assert got == want, (w, h, seed, qp, user_bs, pitched)

    run(352, 288, 1, 35, False)
    run(352, 288, 2, 30, False)          # resident default bS reused
    run(352, 288, 3, 41, True)           # caller bS overwrites it
    run(352, 288, 4, 35, False)          # default must be rebuilt
    run(288, 352, 5, 35, False)          # same byte count, other geometry
    run(16, 16, 6, 45, False)            # smallest 4:2:0 frame
    run(1040, 32, 7, 38, False, True)    # a row wider than one workgroup, pitched planes
    run(32, 1040, 8, 38, True, True)
    run(352, 288, 9, 35, False)
    frames, want = [], []
    for i in range(4):
        y, u, v = synth.blocky_yuv420(352, 288, seed=40 + i)
        want.append(oracle.filter_yuv420(oracle.join_yuv420(y, u, v), 352, 288, 33))
        frames.append((y.copy(), u.copy(), v.copy()))
    ctx.filter_sequence(frames, qp=33)
    for i, pl in enumerate(frames):
        assert oracle.join_yuv420(*pl) == want[i], i
    run(352, 288, 10, 35, False)


def test_host_frame_operator_16bit_containers(ctx, oracle):
    """hevc_deblocking_filter on 10 / 12-bit 4:2:0 frames in 16-bit containers: small frames (one fused Y+U+V launch of the
    16-bit kernels working on page-locked memory, no DMA), pitched caller planes whose rows are not 8-byte aligned (staged into
    the tightly packed pinned buffer), page-locked caller planes (worked on where they lie), and a large frame (strip pipeline).  Parity
    unpinned beyond 8 bit (the reference is 8-bit only): checked against the CPU restatement."""
    from gpu_video_codec_amd import synth

    def want_of(planes, qp, bd, **kw):
        return [oracle.filter_plane(planes[0], qp, bit_depth=bd, vert_bs=kw.get("vert_bs"), hor_bs=kw.get("hor_bs")),
                oracle.filter_plane(planes[1], qp, bit_depth=bd, is_chroma=True),
                oracle.filter_plane(planes[2], qp, bit_depth=bd, is_chroma=True)]

    for (w, h, bd, qp, seed) in [(352, 288, 10, 35, 1), (16, 16, 10, 40, 2), (1040, 32, 12, 38, 3), (1280, 720, 10, 30, 4),
                                 (3840, 2160, 10, 32, 5)]:
        src = synth.blocky_yuv420(w, h, seed=seed, bit_depth=bd)
        kw = {}
        if seed == 1:
            kw["vert_bs"], kw["hor_bs"] = oracle.lcg_bs(w, h, seed)
        want = want_of(src, qp, bd, **kw)
        got = [p.copy() for p in src]
        ctx.filter_frame(*got, qp=qp, bit_depth=bd, **kw)
        for g, x in zip(got, want):
            assert np.array_equal(g, x), (w, h, bd)
        if w <= 1280:
            # pitched views: 3 samples of row padding = 6 bytes, rows not 8-byte aligned
            bufs = [np.full((p.shape[0], p.shape[1] + 3), 0x0155, np.uint16) for p in src]
            views = [b[:, :p.shape[1]] for b, p in zip(bufs, src)]
            for vw, p in zip(views, src):
                vw[:] = p
            ctx.filter_frame(*views, qp=qp, bit_depth=bd, **kw)
            assert all((b[:, -3:] == 0x0155).all() for b in bufs)
            for vw, x in zip(views, want):
                assert np.array_equal(vw, x), (w, h, bd, "pitched")
            # page-locked caller planes: the kernel works on them where they lie
            pinned = [ctx.pinned_array(p.shape, np.uint16) for p in src]
            for pp, p in zip(pinned, src):
                pp[:] = p
            ctx.filter_frame(*pinned, qp=qp, bit_depth=bd, **kw)
            for pp, x in zip(pinned, want):
                assert np.array_equal(pp, x), (w, h, bd, "pinned")
            for pp in pinned:
                ctx.free_pinned(pp)


def test_multi_frame_yuv_file_operator(ctx, oracle, golden_inputs, tmp_path):
    """hevcdbk_filter_yuv_file: a sequence file is filtered frame by frame exactly as the reference filters a
    one-frame file; chunking (more frames than one pinned chunk holds), caller bS, and the error paths."""
    from gpu_video_codec_amd import synth, _lib
    # (1) the bundled one-frame file: identical to the reference's own output
    src = os.path.join(GOLDEN, "mother-daughter_352x288_yv12.yuv")
    out = tmp_path / "one.yuv"
    n, wall = ctx.filter_yuv_file(src, str(out), 352, 288, 35)
    assert n == 1 and wall > 0
    with open(os.path.join(GOLDEN, "mother-daughter_qp35.ref.yuv"), "rb") as fh:
        assert out.read_bytes() == fh.read()
    # (2) 150 CIF frames (> 64 per chunk -> 3 chunks, the last one short), default bS
    frames = [oracle.join_yuv420(*synth.blocky_yuv420(352, 288, seed=100 + i)) for i in range(150)]
    seq = tmp_path / "seq.yuv"
    seq.write_bytes(b"".join(frames))
    out = tmp_path / "seq_out.yuv"
    n, _ = ctx.filter_yuv_file(str(seq), str(out), 352, 288, 37)
    assert n == 150
    got = out.read_bytes()
    fb = 352 * 288 * 3 // 2
    assert len(got) == 150 * fb
    for i in (0, 1, 63, 64, 65, 127, 128, 149):
        assert got[i * fb:(i + 1) * fb] == oracle.filter_yuv420(frames[i], 352, 288, 37), i
    # (3) caller luma bS (SetBoundaryStrenght semantics: chroma keeps the default), 768x576, 5 frames
    vb, hb = oracle.lcg_bs(768, 576, 77)
    base = golden_inputs["image2"]
    fr = [base] + [oracle.join_yuv420(*synth.blocky_yuv420(768, 576, seed=300 + i)) for i in range(4)]
    seq.write_bytes(b"".join(fr))
    n, _ = ctx.filter_yuv_file(str(seq), str(out), 768, 576, 30, vert_bs=vb, hor_bs=hb)
    assert n == 5
    got = out.read_bytes()
    fb = 768 * 576 * 3 // 2
    for i in range(5):
        assert got[i * fb:(i + 1) * fb] == oracle.filter_yuv420(fr[i], 768, 576, 30, vert_bs=vb, hor_bs=hb), i
    # (4) errors: ragged size, bad dimensions, same file, missing file
    rag = tmp_path / "ragged.yuv"
    rag.write_bytes(frames[0] + b"\0" * 17)
    for args, code in (((str(rag), str(out), 352, 288, 30), _lib.ERR_FILE_SIZE),
                       ((str(tmp_path / "missing.yuv"), str(out), 352, 288, 30), _lib.ERR_IO),
                       ((str(seq), str(seq), 768, 576, 30), _lib.ERR_ARG)):
        with pytest.raises(_lib.DeblockError) as e:
            ctx.filter_yuv_file(*args)
        assert e.value.code == code, args
    odd = tmp_path / "odd.yuv"
    odd.write_bytes(b"\0" * (24 * 24 * 3 // 2))  # 24x24: luma fine, chroma 12x12 is not a multiple of 8
    with pytest.raises(_lib.DeblockError) as e:
        ctx.filter_yuv_file(str(odd), str(out), 24, 24, 30)
    assert e.value.code == _lib.ERR_DIMENSIONS


def test_reference_padded_plane_layout(ctx, oracle, golden_inputs):
    """A caller that keeps the reference's (W+8)x(H+8) padded planes (cpu.h:55-82) passes the interior pointer and
    the padded pitch: no pack/unpack step, the padding stays as it was, the interior equals the reference's Save."""
    w, h = 352, 288
    y, u, v = oracle.split_yuv420(golden_inputs["image1"], w, h)
    padded = []
    for p in (y, u, v):
        buf = np.zeros((p.shape[0] + 8, p.shape[1] + 8), np.uint8)
        buf[4:-4, 4:-4] = p
        padded.append(buf)
    views = [b[4:-4, 4:-4] for b in padded]
    ctx.filter_frame(*views, qp=30)
    want = oracle.split_yuv420(oracle.filter_yuv420(golden_inputs["image1"], w, h, 30), w, h)
    for b, vw, wnt in zip(padded, views, want):
        assert np.array_equal(vw, wnt)
        assert not b[:4].any() and not b[-4:].any() and not b[:, :4].any() and not b[:, -4:].any()


def test_two_contexts_driven_from_two_host_threads(oracle):
    """a context is single-threaded, different contexts may run concurrently (the reference's GPU
    path is process-global state, gpu.cu:37-77).  Two threads, each with its own context, filter different frames with
    different QPs at the same time; both must get their own exact results, repeatedly."""
    import threading
    from gpu_video_codec_amd import deblock, synth
    errors = []

    def work(seed, qp, w, h):
        try:
            ctx = deblock.Context(0)
            for it in range(12):
                y, u, v = synth.blocky_yuv420(w, h, seed=seed + it)
                want = oracle.filter_yuv420(oracle.join_yuv420(y, u, v), w, h, qp)
                y, u, v = y.copy(), u.copy(), v.copy()
                ctx.filter_frame(y, u, v, qp=qp)
                if oracle.join_yuv420(y, u, v) != want:
                    errors.append((seed, it))
            ctx.close()
        except Exception as e:  # noqa
            errors.append((seed, repr(e)))

    ts = [threading.Thread(target=work, args=(100, 30, 352, 288)), threading.Thread(target=work, args=(200, 41, 768, 576))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_packed_16bit_chroma_kernel(ctx, oracle):
    """Reference-exact chroma on 16-bit containers through the packed kernel (10 and 12 bit; 13 bit is refused), both
    kernels against the oracle: default bS (guard quirk Q9), caller bS, a QP map, widths with partial and row-major waves."""
    from gpu_video_codec_amd import synth, _lib
    rng = np.random.default_rng(31)
    for (w, h, bd) in [(8, 8, 10), (264, 72, 10), (1032, 40, 12), (4104, 24, 10), (960, 544, 12)]:
        c = np.stack([synth.blocky_plane(w, h, seed=int(rng.integers(1, 1 << 30)), bit_depth=bd, dc_range=4) for _ in range(2)])
        c[0, : h // 2, : w // 2] = rng.integers(0, 1 << bd, (h // 2, w // 2), dtype=np.uint16)
        vb, hb = oracle.lcg_bs(w, h, 5)
        for variant in (_lib.KERNEL_GENERIC, _lib.KERNEL_PACKED):
            for qp in (27, 40, 51):
                got = run_batch(ctx, c, qp, variant=variant, bit_depth=bd, is_chroma=True, bs=[(vb, hb), oracle.default_bs(w, h)])
                assert np.array_equal(got[0], oracle.filter_plane(c[0], qp, bit_depth=bd, is_chroma=True, vert_bs=vb, hor_bs=hb)), (w, h, bd, variant, qp)
                assert np.array_equal(got[1], oracle.filter_plane(c[1], qp, bit_depth=bd, is_chroma=True)), (w, h, bd, variant, qp)
    qmap = synth.ctu_qp_map(2 * 264, 2 * 72, seed=3, lo=25, hi=51)
    c = synth.blocky_plane(264, 72, seed=77, bit_depth=10, dc_range=4)
    got = run_batch(ctx, c[None], 0, variant=_lib.KERNEL_PACKED, bit_depth=10, is_chroma=True, qp_map=qmap)
    assert np.array_equal(got[0], oracle.filter_plane(c, 0, bit_depth=10, is_chroma=True, qp_map=qmap))
    from gpu_video_codec_amd import deblock
    t = synth.blocky_plane(64, 64, seed=1, bit_depth=13)
    with pytest.raises(deblock.DeblockError) as e:
        run_batch(ctx, t[None], 40, variant=_lib.KERNEL_PACKED, bit_depth=13, is_chroma=True)
    assert e.value.code == _lib.ERR_UNSUPPORTED


def test_file_operator_sharded_over_workers(ctx, oracle, tmp_path):
    """hevcdbk_filter_yuv_file_multi (SURVEY 8e inside the C ABI): chunks of the file go round-robin to worker threads, each
    with its own context.  The device lists enumerate hevcdbk_device_count(): on a one-GPU box the workers share device 0
    -- the sharding, offsets and ownership logic are exactly what G devices run -- and on an 8-GPU node every real device
    filters its share.  Output must equal the single-context operator's byte for byte."""
    from gpu_video_codec_amd import deblock, synth
    w, h, n = 352, 288, 150
    frames = [oracle.join_yuv420(*synth.blocky_yuv420(w, h, seed=500 + i)) for i in range(n)]
    src = tmp_path / "in.yuv"
    src.write_bytes(b"".join(frames))
    one = tmp_path / "one.yuv"
    assert ctx.filter_yuv_file(str(src), str(one), w, h, 34)[0] == n
    want = one.read_bytes()
    fb = w * h * 3 // 2
    for i in (0, 63, 64, 149):
        assert want[i * fb:(i + 1) * fb] == oracle.filter_yuv420(frames[i], w, h, 34), i
    ndev = deblock.device_count()
    every = list(range(ndev))              # one worker per REAL device of the box (1 on the test box, 8 on a node)
    twice = every + every[::-1]            # and each device driven by two workers
    for devices in ([0], [0, 0], [0, 0, 0, 0, 0], every, twice):
        out = tmp_path / ("multi%d_%d.yuv" % (len(devices), devices[-1]))
        got_n, wall = deblock.filter_yuv_file_multi(devices, str(src), str(out), w, h, 34)
        assert got_n == n and wall > 0
        assert out.read_bytes() == want, devices
    # caller bS travels to every worker
    vb, hb = oracle.lcg_bs(w, h, 3)
    deblock.filter_yuv_file_multi([0, 0], str(src), str(one), w, h, 40, vert_bs=vb, hor_bs=hb)
    got = one.read_bytes()
    for i in (0, 100):
        assert got[i * fb:(i + 1) * fb] == oracle.filter_yuv420(frames[i], w, h, 40, vert_bs=vb, hor_bs=hb), i
    # a device that does not exist fails the call
    with pytest.raises(deblock.DeblockError):
        deblock.filter_yuv_file_multi([0, 99], str(src), str(one), w, h, 34)


def test_single_frame_operator_on_page_locked_planes(ctx, oracle):
    """A large frame whose planes live in page-locked caller memory is DMA'd where it lies (no staging copy), pitched rows
    included; pinned and pageable planes may be mixed within one frame.  Results and untouched row padding as always."""
    from gpu_video_codec_amd import synth
    for (w, h) in ((1920, 1088), (352, 288)):  # the strip pipeline, and the small-frame path (kernel on the planes themselves)
        _page_locked_case(ctx, oracle, synth, w, h)


def _page_locked_case(ctx, oracle, synth, w, h):
    y, u, v = synth.blocky_yuv420(w, h, seed=61)
    want = oracle.split_yuv420(oracle.filter_yuv420(oracle.join_yuv420(y, u, v), w, h, 33), w, h)
    for mix in ((True, True, True), (True, False, True), (False, True, False)):
        bufs, views = [], []
        for p, pin in zip((y, u, v), mix):
            shape = (p.shape[0], p.shape[1] + 64)
            b = ctx.pinned_array(shape, np.uint8) if pin else np.empty(shape, np.uint8)
            b[:] = 0x3C
            b[:, :p.shape[1]] = p
            bufs.append(b)
            views.append(b[:, :p.shape[1]])
        t = ctx.filter_frame(*views, qp=33)
        assert t["pipelined_s"] > 0
        for vw, wnt, b in zip(views, want, bufs):
            assert np.array_equal(vw, wnt), mix
            assert (b[:, -64:] == 0x3C).all(), mix
        for b, pin in zip(bufs, mix):
            if pin:
                ctx.free_pinned(b)


def test_strip_pipeline_on_large_frames_all_kernels(ctx, oracle):
    """Frames large enough for several strips per plane through hevc_deblocking_filter: 8-bit (packed kernels), 10-bit
    (packed 16-bit kernels) and 12-bit (32-bit kernels), caller luma bS and a per-CTU QP map -- the block-row range
    launches of every kernel family must reproduce the whole-plane result."""
    from gpu_video_codec_amd import synth
    w, h = 1920, 1088
    for bd in (8, 10, 12):
        y, u, v = synth.blocky_yuv420(w, h, seed=70 + bd, bit_depth=bd)
        vb, hb = oracle.lcg_bs(w, h, bd)
        for qmap in (None, synth.ctu_qp_map(w, h, seed=bd, lo=24, hi=46)):
            qp = 0 if qmap is not None else 37
            want = [oracle.filter_plane(y, qp, bit_depth=bd, vert_bs=vb, hor_bs=hb, qp_map=qmap),
                    oracle.filter_plane(u, qp, bit_depth=bd, is_chroma=True, qp_map=qmap),
                    oracle.filter_plane(v, qp, bit_depth=bd, is_chroma=True, qp_map=qmap)]
            gy, gu, gv = y.copy(), u.copy(), v.copy()
            ctx.filter_frame(gy, gu, gv, qp=qp, bit_depth=bd, vert_bs=vb, hor_bs=hb, qp_map=qmap)
            for g, wnt, nm in zip((gy, gu, gv), want, "YUV"):
                assert np.array_equal(g, wnt), (bd, qmap is not None, nm)


def test_staging_crew_sizes_on_large_pageable_frames(ctx, oracle):
    """hevc_deblocking_filter on large frames in ordinary pageable memory with 1, 2, 4 and 8 copying threads (the caller
    alone ... the crew of host_crew.h): a 4K luma frame in tight rows (block copies with streaming stores) and a 1080p 4:2:0
    frame in pitched rows; results, the untouched row padding, and the strip record of the call (hevcdbk_last_frame_trace:
    the strips tile every plane exactly, first strips shorter than later ones, all phases in order)."""
    from gpu_video_codec_amd import synth
    # three DIFFERENT frames go through the same HBM / ring buffers call after call: a stale line of the previous frame anywhere
    # between the host's writes and the kernel's reads (the crew writes HBM through the BAR, behind the GPU's caches) would show
    y4ks = [synth.blocky_plane(3840, 2160, seed=81 + 5 * i) for i in range(3)]
    want4ks = [oracle.filter_plane(y, 32, threads=8) for y in y4ks]
    y4k = y4ks[0]
    y, u, v = synth.blocky_yuv420(1920, 1088, seed=82)
    want = oracle.split_yuv420(oracle.filter_yuv420(oracle.join_yuv420(y, u, v), 1920, 1088, 36), 1920, 1088)
    try:
        for n in (1, 2, 4, 8):
            ctx.set_host_threads(n)
            assert ctx.host_threads() == n
            for rep in (0, 1, 2, 1, 0, 2):  # the crew sleeps between calls and is woken again
                g = y4ks[rep].copy()
                t = ctx.filter_frame(g, qp=32)
                assert np.array_equal(g, want4ks[rep]), (n, rep)
            tr = ctx.last_frame_trace()
            assert len(tr) >= 4 and tr[0]["row_begin"] == 0 and tr[-1]["row_end"] == 2160
            assert all(a["row_end"] == b["row_begin"] for a, b in zip(tr, tr[1:]))
            assert sum(s["bytes"] for s in tr) == y4k.nbytes and tr[0]["bytes"] < tr[-2]["bytes"]
            for s in tr:
                assert 0 < s["stage_begin_s"] <= s["stage_end_s"] <= s["enqueue_end_s"] <= s["d2h_seen_s"] <= s["unstage_end_s"] <= t["pipelined_s"], s
                assert s["kernel_ms"] > 0 and s["d2h_ms"] == 0   # the kernel stores into the ring itself: no D2H DMA
            bufs = []
            for p in (y, u, v):
                b = np.full((p.shape[0], p.shape[1] + 48), 0x3C, np.uint8)
                b[:, :p.shape[1]] = p
                bufs.append(b)
            ctx.filter_frame(*[b[:, :p.shape[1]] for b, p in zip(bufs, (y, u, v))], qp=36)
            for b, p, wnt in zip(bufs, (y, u, v), want):
                assert np.array_equal(b[:, :p.shape[1]], wnt), n
                assert (b[:, p.shape[1]:] == 0x3C).all(), n
            assert {s["plane"] for s in ctx.last_frame_trace()} == {0, 1, 2}
        ctx.filter_frame(*[p.copy() for p in synth.blocky_yuv420(352, 288, seed=3)], qp=30)
        assert ctx.last_frame_trace() == []   # a small frame has no strips
    finally:
        ctx.set_host_threads(0)


def test_registered_caller_memory_is_filtered_in_place(ctx, oracle):
    """hevcdbk_host_register: a caller that reuses its buffers page-locks them once; the host-frame operator then DMAs the planes
    where they lie or stores into them from the kernel (the strip record carries no un-stage times), tight and pitched, several frames through
    the same buffer; unregistering twice is an argument error, and the buffer still works as pageable memory afterwards."""
    from gpu_video_codec_amd import synth, deblock, _lib
    w, h = 3840, 2160
    buf = np.empty((h, w + 128), np.uint8)
    ctx.host_register(buf)
    try:
        for seed, view in ((91, buf[:, :w]), (92, buf.reshape(-1)[: w * h].reshape(h, w)), (93, buf[:, :w])):
            y = synth.blocky_plane(w, h, seed=seed)
            buf[:] = 0x77
            view[:] = y
            ctx.filter_frame(view, qp=34)
            assert np.array_equal(view, oracle.filter_plane(y, 34, threads=8)), seed
            tr = ctx.last_frame_trace()
            assert tr and all(s["unstage_end_s"] == 0 for s in tr), seed   # results are never copied out of a ring by the host
            if view.strides[0] != w:
                assert (buf[:, w:] == 0x77).all()
    finally:
        ctx.host_unregister(buf)
    with pytest.raises(deblock.DeblockError) as e:
        ctx.host_unregister(buf)
    assert e.value.code == _lib.ERR_ARG
    y = synth.blocky_plane(w, h, seed=94)
    buf[:, :w] = y
    ctx.filter_frame(buf[:, :w], qp=34)
    assert np.array_equal(buf[:, :w], oracle.filter_plane(y, 34, threads=8))
    assert all(s["unstage_end_s"] > 0 for s in ctx.last_frame_trace())   # pageable again: results come back through the ring


def test_cpp_class_mirror_against_golden_manifest(manifest, tmp_path):
    """examples/read_yuv_frame.cpp = the reference's ExecuteCpu body on hevcdbk::ReadYuvFrame (include/hevc_deblock.hpp):
    ctor -> [SetBoundaryStrenght with the seeded generator] -> DeblockingFilter -> Save, against the reference's own hashes."""
    import subprocess
    from conftest import ROOT
    from gpu_video_codec_amd import _lib
    exe = str(tmp_path / "ryf")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "read_yuv_frame.cpp"),
                           "-L", libdir, "-lhevcdbk", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    files = {"image1": "image1_352x288_yv12.yuv", "mother-daughter": "mother-daughter_352x288_yv12.yuv", "image2": "image2_768x576.yuv"}
    checked = 0
    for name, ent in manifest["images"].items():
        for c in ent["cases"][::4]:
            out = tmp_path / "o.yuv"
            cmd = [exe, os.path.join(GOLDEN, files[name]), str(out), str(ent["width"]), str(ent["height"]), str(c["qp"])]
            if c["bs_seed"] is not None:
                cmd.append(str(c["bs_seed"]))
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, (cmd, r.stderr)
            assert sha256(out.read_bytes()) == c["sha256"], (name, c)
            checked += 1
    assert checked >= 12


def test_plain_c_host_frames_example(oracle, tmp_path):
    """examples/host_frames.c on the GPU: a malloc'ed 4:2:0 4K frame, the same buffer registered, and a sequence of six, from
    strict C99 through the C ABI; the example checks that the three ways agree byte for byte."""
    import subprocess
    from gpu_video_codec_amd import _lib
    exe = str(tmp_path / "host_frames")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "host_frames.c"), "-L", libdir, "-lhevcdbk", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all three ways agree byte for byte" in r.stdout, (r.stdout, r.stderr)
    assert "strip  0 plane 0 rows    0.." in r.stdout


def test_host_frame_paths_a_large_bar_device_does_not_take(oracle):
    """On MI355X the large-frame host call writes through the BAR and lets the kernels store into page-locked memory; a device
    without a large BAR takes ring + H2D DMA instead, and a build without direct stores adds the D2H DMA.  Both forms stay in the
    library, so both stay tested: the diagnostic build (the same sources; it reads HEVCDBK_HOST_PUSH / _DIRECT_OUT) filters a 4K
    luma frame each way in a child process and compares with the oracle."""
    import json
    import subprocess
    tool = os.path.join(ROOT, "tools", "host_frame_4k.py")
    for env in ({"HEVCDBK_HOST_PUSH": "0"}, {"HEVCDBK_HOST_PUSH": "0", "HEVCDBK_HOST_DIRECT_OUT": "0"}, {"HEVCDBK_HOST_STREAM_STORES": "0"},
                {"HEVCDBK_HOST_AFFINITY": "0", "HEVCDBK_HOST_THREADS": "2"}):
        r = subprocess.run([sys.executable, tool, "--calls", "3", "--diag", "--check"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["luma_bit_exact_vs_oracle"] is True, env
        strips = d["last_call_strips"]
        assert strips and (env.get("HEVCDBK_HOST_DIRECT_OUT") == "0") == any(s["d2h_ms"] > 0 for s in strips), env
        assert (env.get("HEVCDBK_HOST_PUSH") == "0") == any(s["h2d_ms"] > 0 for s in strips), env


def test_probed_destination_pool(ctx, oracle):
    """hevcdbk_device_malloc_probed: a destination pool picked among candidate allocations by timing the filter on each.  The
    pool it returns is ordinary device memory of the size the launch writes: the launch into it equals the oracle (8-bit and
    10-bit batches, one candidate and several), best <= worst, and bad arguments are refused before anything is allocated."""
    from gpu_video_codec_amd import deblock, synth, _lib
    for (w, h, bd, n, cands) in ((640, 360 // 8 * 8, 8, 3, 1), (1280, 720, 10, 4, 5)):
        fr = np.stack([synth.blocky_plane(w, h, seed=60 + f + bd, bit_depth=bd) for f in range(n)])
        b = deblock.DeviceBatch(ctx, w, h, n, bit_depth=bd, per_frame_bs=False)
        b.upload_all(fr)
        pool, best, worst = ctx.alloc_probed(b.planes(), 33, cands)
        assert pool.ptr and pool.nbytes == b.frame_bytes * n and 0 < best <= worst
        p = b.planes()
        p.dst = pool.ptr
        pool.upload(np.zeros(pool.nbytes, np.uint8))
        ctx.filter_device(p, 33)
        ctx.synchronize()
        got = pool.download(dtype=fr.dtype).reshape(n, h, w)
        for f in range(n):
            assert np.array_equal(got[f], oracle.filter_plane(fr[f], 33, bit_depth=bd, threads=4)), (bd, f)
        for bad in (0, 17):
            with pytest.raises(deblock.DeblockError) as e:
                ctx.alloc_probed(b.planes(), 33, bad)
            assert e.value.code == _lib.ERR_ARG
        pool.free()
        b.free()


def test_large_host_frames_of_odd_shapes(ctx, oracle):
    """The strip schedule of the large-frame host call on shapes far from a video frame: planes 8 .. 24 samples wide and a hundred
    thousand rows tall, very wide and flat ones, a width that is not a multiple of 64, 8 and 10 bit, 4:2:0 -- the strips must tile
    every plane exactly (whole block rows, nothing twice, nothing left out) and the result must equal the oracle, with the crew
    and with the caller alone."""
    from gpu_video_codec_amd import synth
    try:
        for (w, h, bd, chroma) in ((16, 163840, 8, False), (8, 300000, 8, False), (8192, 512, 8, False), (2000, 1200, 8, False),
                                   (24, 90000, 10, False), (32, 70000, 8, True), (16384, 256, 10, False)):
            pl = synth.blocky_yuv420(w, h, seed=w + h, bit_depth=bd) if chroma else (synth.blocky_plane(w, h, seed=w + h, bit_depth=bd),)
            want = [oracle.filter_plane(p, 37, bit_depth=bd, is_chroma=k > 0, threads=8) for k, p in enumerate(pl)]
            for threads in (4, 1):
                ctx.set_host_threads(threads)
                got = [p.copy() for p in pl]
                ctx.filter_frame(*got, qp=37, bit_depth=bd)
                for g, wnt in zip(got, want):
                    assert np.array_equal(g, wnt), (w, h, bd, chroma, threads)
                tr = ctx.last_frame_trace()
                assert tr, (w, h)
                for k, p in enumerate(pl):
                    mine = [s for s in tr if s["plane"] == k]
                    assert mine[0]["row_begin"] == 0 and mine[-1]["row_end"] == p.shape[0], (w, h, k)
                    assert all(a["row_end"] == b["row_begin"] for a, b in zip(mine, mine[1:])), (w, h, k)
                    assert all((s["row_begin"] + 4) % 8 == 0 for s in mine[1:]), (w, h, k)   # strips start on block rows: 8b - 4
                    assert sum(s["bytes"] for s in mine) == p.nbytes
    finally:
        ctx.set_host_threads(0)
