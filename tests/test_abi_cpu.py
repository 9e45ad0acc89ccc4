"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol declared in
include/hevc_deblock.h, its host-only helpers agree with the oracle, and -- with no GPU -- every
compute entry point fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def L():
    from gpu_video_codec_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def test_exports_match_header(L):
    from gpu_video_codec_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "hevc_deblock.h")).read()
    declared = set(re.findall(r"\b(hevcdbk_[a-z0-9_]+|hevc_sao_filter_device|hevc_deblock_sao(?:_h265)?_device(?:_planes)?|hevc_deblocking_filter(?:_h265_device|_h265|_device_planes|_device|_sequence)?)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for s in declared:
        assert hasattr(L, s), s


def test_cpp_shim_symbol_present():
    """main.cu's forward declaration (main.cu:87-90) must resolve against the library."""
    import subprocess
    from gpu_video_codec_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", _lib.LIB_PATH]).decode()
    assert "ExecuteGpu(std::" in out


def test_no_oracle_or_cpu_filter_in_product():
    """The product library must not link or embed the checker."""
    import subprocess
    from gpu_video_codec_amd import _lib
    out = subprocess.check_output(["nm", "-D", _lib.LIB_PATH]).decode()
    assert "dbko_" not in out and "ref_frame" not in out
    needed = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH]).decode()
    assert "liboracle" not in needed and "libref_oracle" not in needed and "libgomp" not in needed
    for fn in os.listdir(os.path.join(ROOT, "gpu_video_codec_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "gpu_video_codec_amd", fn)).read()
            assert "oracle" not in src.replace("the oracle", "").lower() or fn == "deblock.py" and "import oracle" not in src


def test_product_library_reads_no_environment_and_has_no_diagnostics():
    """The diagnostics (copy variant, timing-only ablations with WRONG pixels, the LDS-queue kernel, launch knobs) live in
    libhevcdbk_diag.so only: the shipped library does not import getenv, carries none of the knob names or diagnostic
    kernels, and does not export the knob setter."""
    import subprocess
    from gpu_video_codec_amd import _lib
    und = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH]).decode()
    assert "getenv" not in und
    blob = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"HEVCDBK_TUNE", b"HEVCDBK_WG", b"dbk_packed_q_kernel", b"nostrong", b"hevcdbk_diag_set"):
        assert needle not in blob, needle
    assert os.path.exists(_lib.DIAG_LIB_PATH)
    dsyms = subprocess.check_output(["nm", "-D", "--defined-only", _lib.DIAG_LIB_PATH]).decode()
    assert "hevcdbk_diag_set" in dsyms
    assert b"dbk_packed_q_kernel" in open(_lib.DIAG_LIB_PATH, "rb").read()


def test_tables_and_bs_helpers_match_oracle(L, oracle):
    from gpu_video_codec_amd import deblock
    tc, beta = deblock.default_tables()
    otc, obeta = oracle.tables()
    assert np.array_equal(tc, otc) and np.array_equal(beta, obeta)
    for (w, h) in [(8, 8), (16, 8), (352, 288), (176, 144), (768, 576), (3840, 2160), (7680, 4320)]:
        assert deblock.num_vert_bs(w, h) == oracle.num_vert_bs(w, h)
        assert deblock.num_hor_bs(w, h) == oracle.num_hor_bs(w, h)
        v, hh = deblock.default_bs(w, h)
        ov, oh = oracle.default_bs(w, h)
        assert np.array_equal(v, ov) and np.array_equal(hh, oh)
    assert deblock.num_vert_bs(3840, 2160) == 129870 and deblock.num_hor_bs(3840, 2160) == 130080


def test_error_strings_are_the_reference_messages(L):
    from gpu_video_codec_amd import _lib
    assert L.hevcdbk_strerror(_lib.ERR_FILE_SIZE) == b"Incorrect file size"
    assert L.hevcdbk_strerror(_lib.ERR_DIMENSIONS) == b"Width and height of image must be multiplier of sample block size"
    assert L.hevcdbk_strerror(_lib.ERR_BS_SIZE) == b"Incorrect size of input boundary strenght array"


def test_fails_loudly_without_gpu(L, tmp_path):
    """No device => HEVCDBK_ERR_HIP from create and from ExecuteGpu; never a silent CPU result."""
    from gpu_video_codec_amd import _lib, deblock
    if deblock.device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert L.hevcdbk_create(0, C.byref(h)) == _lib.ERR_HIP
    with pytest.raises(deblock.DeblockError):
        deblock.Context(0)
    src = os.path.join(ROOT, "tests", "golden", "image1_352x288_yv12.yuv")
    out = tmp_path / "o.yuv"
    assert L.hevcdbk_execute_gpu(src.encode(), str(out).encode(), 352, 288, 30, 20, 20, 20, 20, 0) == _lib.ERR_HIP
    assert not out.exists()


def test_round3_entries_validate_arguments_without_a_gpu(L):
    """The entries added in round 3 -- steady-state replay, PCI id, multi-plane deblocking + SAO -- refuse bad arguments
    before they touch a device (no compute call is made here)."""
    from gpu_video_codec_amd import _lib
    planes = (_lib.DevicePlanes * 2)()
    sao = (_lib.SaoPlane * 2)()
    r = _lib.Replay(steps=3)
    ms = (C.c_float * 3)()
    assert L.hevcdbk_device_replay(None, planes, 1, 30, None, 0, C.byref(r), ms) == _lib.ERR_ARG
    assert L.hevcdbk_device_pci_bus_id(None, C.create_string_buffer(64), 64) == _lib.ERR_ARG
    assert L.hevc_deblock_sao_device_planes(None, planes, 2, 30, None, sao, _lib.FUSED_AUTO, None) == _lib.ERR_ARG
    assert L.hevc_deblock_sao_h265_device_planes(None, planes, 2, 30, None, sao, _lib.FUSED_AUTO, None) == _lib.ERR_ARG
    # the struct layouts the bindings assume (include/hevc_deblock.h: hevcdbk_replay, hevcdbk_sao_plane)
    assert C.sizeof(_lib.Replay) == 3 * 8 + 3 * 4 + 2 * 4 + 4 + 6 * 8 or C.sizeof(_lib.Replay) % 8 == 0
    assert C.sizeof(_lib.SaoPlane) == 56


def test_execute_gpu_validation_order(L, tmp_path):
    """Size check before divisibility check, as gpu.cu:1082-1087 / cpu.h:43-48."""
    from gpu_video_codec_amd import _lib
    p = tmp_path / "short.yuv"
    p.write_bytes(b"\0" * 100)
    assert L.hevcdbk_execute_gpu(str(p).encode(), b"/dev/null", 352, 288, 30, 0, 0, 0, 0, 0) == _lib.ERR_FILE_SIZE
    p2 = tmp_path / "odd.yuv"
    p2.write_bytes(b"\0" * (3 * 20 * 20 // 2))
    assert L.hevcdbk_execute_gpu(str(p2).encode(), b"/dev/null", 20, 20, 30, 0, 0, 0, 0, 0) == _lib.ERR_DIMENSIONS
    assert L.hevcdbk_execute_gpu(b"/nonexistent.yuv", b"/dev/null", 16, 16, 30, 0, 0, 0, 0, 0) == _lib.ERR_IO


def test_read_yuv_frame_mirror_validation(tmp_path):
    from gpu_video_codec_amd import deblock, _lib
    p = tmp_path / "short.yuv"
    p.write_bytes(b"\0" * 100)
    with pytest.raises(deblock.DeblockError) as e:
        deblock.ReadYuvFrame(str(p), 352, 288, 30, ctx=object())
    assert e.value.code == _lib.ERR_FILE_SIZE
    p2 = tmp_path / "ok.yuv"
    p2.write_bytes(b"\0" * (3 * 16 * 16 // 2))
    f = deblock.ReadYuvFrame(str(p2), 16, 16, 30, ctx=object())
    with pytest.raises(deblock.DeblockError) as e:
        f.SetBoundaryStrenght(np.zeros(3, np.uint8), np.zeros(4, np.uint8))
    assert e.value.code == _lib.ERR_BS_SIZE


def test_sharded_file_operator_argument_errors_need_no_gpu(L, tmp_path):
    """The argument / file checks of hevcdbk_filter_yuv_file_multi run before any device is touched."""
    import ctypes as C
    from gpu_video_codec_amd import _lib
    dev = (C.c_int * 2)(0, 1)
    rag = tmp_path / "ragged.yuv"
    rag.write_bytes(b"\0" * (352 * 288 * 3 // 2 + 5))
    out = str(tmp_path / "o.yuv").encode()
    call = lambda inp, w, h, nd=2: L.hevcdbk_filter_yuv_file_multi(dev, nd, inp, out, w, h, 30, None, None, None, None)
    assert call(str(rag).encode(), 352, 288) == _lib.ERR_FILE_SIZE
    assert call(str(tmp_path / "missing.yuv").encode(), 352, 288) == _lib.ERR_IO
    assert call(out, 352, 288) == _lib.ERR_ARG              # in == out
    assert call(str(rag).encode(), 352, 288, 0) == _lib.ERR_ARG
    odd = tmp_path / "odd.yuv"
    odd.write_bytes(b"\0" * (24 * 24 * 3 // 2))
    assert call(str(odd).encode(), 24, 24) == _lib.ERR_DIMENSIONS


def test_header_is_plain_c_and_example_links(L, tmp_path):
    """include/hevc_deblock.h must be usable from C (the boundary is a C ABI): the decoder-loop example is compiled as
    strict C99 and linked against the library; without a GPU it must stop at hevcdbk_create with its own exit code."""
    import subprocess
    from gpu_video_codec_amd import _lib
    src = os.path.join(ROOT, "examples", "decoder_loop.c")
    exe = str(tmp_path / "decoder_loop")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src,
                           "-L", libdir, "-lhevcdbk", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    from gpu_video_codec_amd import deblock
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if deblock.device_count() > 0:
        assert r.returncode == 0 and "deblocking and SAO on the GPU" in r.stdout, r.stderr
    else:
        assert r.returncode == 2 and "no CPU path" in r.stderr


def test_host_frames_example_is_plain_c(L, tmp_path):
    """examples/host_frames.c -- the reference-shaped host call with the round-4 entries (hevcdbk_set_host_threads,
    hevcdbk_host_register, hevcdbk_last_frame_trace, the sequence operator) from strict C99; without a GPU it stops at
    hevcdbk_create with its own exit code."""
    import subprocess
    from gpu_video_codec_amd import _lib, deblock
    exe = str(tmp_path / "host_frames")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "host_frames.c"), "-L", libdir, "-lhevcdbk", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if deblock.device_count() > 0:
        assert r.returncode == 0 and "all three ways agree byte for byte" in r.stdout, (r.stdout, r.stderr)
    else:
        assert r.returncode == 2 and "no CPU path" in r.stderr


def test_cpp_class_mirror_compiles_and_keeps_the_reference_errors(L, tmp_path):
    """include/hevc_deblock.hpp (the ReadYuvFrame surface in C++): builds warning-free, and the reference's two constructor
    checks throw the reference's texts before any device is needed (cpu.h:43-48)."""
    import subprocess
    from gpu_video_codec_amd import _lib
    exe = str(tmp_path / "ryf")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "read_yuv_frame.cpp"), "-L", libdir, "-lhevcdbk", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    img = os.path.join(ROOT, "tests", "golden", "image1_352x288_yv12.yuv")
    out = str(tmp_path / "o.yuv")
    r = subprocess.run([exe, img, out, "352", "280", "30"], capture_output=True, text=True)
    assert r.returncode == 1 and "Incorrect file size" in r.stderr
    odd = tmp_path / "odd.yuv"
    odd.write_bytes(b"\0" * (12 * 8 * 3 // 2))
    r = subprocess.run([exe, str(odd), out, "12", "8", "30"], capture_output=True, text=True)
    assert r.returncode == 1 and "multiplier of sample block size" in r.stderr


def test_only_the_api_leaves_the_library():
    """Built with hidden visibility: the dynamic symbol table holds the declared entry points, the C++ ExecuteGpu symbol and
    nothing of the internal launch interface (generic names such as filter_chunk must not be interposable)."""
    import subprocess
    from gpu_video_codec_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode().split("\n")
    names = [l.split()[-1] for l in out if l.strip()]
    internal = [n for n in names if n.startswith("dbk_") or n in ("filter_chunk", "h265_args", "launch_h265") or "dbk_launch" in n]
    assert not internal, internal
    assert set(_lib.EXPORTS) <= set(names)


REF_MAIN = "/root/reference/hevc_deblocking_filter/main.cu"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference lives in the build container only")
def test_reference_driver_links_against_the_library(tmp_path):
    """INTEGRATION.md section 1, executed: tools/port_main_cu.py applies the listed edits to a copy of the reference's
    main.cu made here at test time (drop the four CUDA includes main.cu:20-21,27-28, the addWithCuda declaration
    main.cu:85, swap the body of GetGpuDeviceInfo main.cu:92-107); the result is plain C++ that compiles against the
    reference's own CPU header and links against libhevcdbk.so -- ExecuteGpu (main.cu:87-90) resolves to the library's
    C++ symbol.  Run in a directory holding the bundled input: ExecuteCpu (the reference's code) writes the reference
    output; without a GPU the following ExecuteGpu call fails loudly (no CPU fallback).  On the GPU box, where the
    reference is absent, the committed hevc_deblock_main driver covers the same call (tests/test_gpu_parity.py)."""
    import shutil
    import subprocess
    import sys
    from gpu_video_codec_amd import _lib
    from conftest import GOLDEN, sha256
    ported = tmp_path / "main_ported.cpp"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "port_main_cu.py"), REF_MAIN, str(ported)])
    text = ported.read_text(errors="replace")
    assert "cuda" not in text.lower().replace("addwithcuda", "") and "hevcdbk_get_device_info" in text
    exe = tmp_path / "deblock"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["g++", "-x", "c++", "-std=c++14", "-O2", "-fopenmp", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.dirname(REF_MAIN), str(ported), "-L", libdir, "-lhevcdbk",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    und = subprocess.check_output(["nm", "-C", "--undefined-only", str(exe)]).decode()
    assert "ExecuteGpu(std::" in und and "hevcdbk_create" in und   # resolved by libhevcdbk.so at load time
    shutil.copy(os.path.join(GOLDEN, "mother-daughter_352x288_yv12.yuv"), tmp_path)
    run = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    cpu_out = tmp_path / "mother-daughter_352x288_yv12_filtered.yuv"      # main.cu:129
    assert cpu_out.exists(), run.stdout[-2000:] + run.stderr[-2000:]
    with open(os.path.join(GOLDEN, "mother-daughter_qp35.ref.yuv"), "rb") as fh:
        assert sha256(cpu_out.read_bytes()) == sha256(fh.read())
    gpu_out = tmp_path / "mother-daughter_352x288_yv12_filtered_gpu.yuv"  # main.cu:130
    from gpu_video_codec_amd import deblock
    if deblock.device_count() > 0:
        assert run.returncode == 0 and sha256(gpu_out.read_bytes()) == sha256(cpu_out.read_bytes())
    else:
        assert run.returncode != 0 and not gpu_out.exists()   # fails loudly: there is no CPU path in the library


def test_no_wide_store_is_overwritten_in_its_shadow():
    """Round 3 found a 16-byte buffer store whose data registers the very next VALU instruction overwrote (legal by the
    compiler's hazard table when soffset is an SGPR; on MI355X the first launch of a process wrote garbage;
    profiles/r04/store_hazard.md).  tools/check_store_hazard.py scans the gfx950 code objects inside the SHIPPED libraries
    (product and diagnostic) and the tools/ubench programs, following both sides of every branch; `make all` runs it too and
    fails on a finding.  Its self-test (block-ending stores, branch targets, objdump and compiler listing forms) runs first."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "check_store_hazard.py")
    r = subprocess.run([sys.executable, tool, "--self-test"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "self-test passed" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 suspicious" in r.stdout and "libhevcdbk" not in r.stdout.replace("0 suspicious", "")
    m = re.search(r"in (\d+) code object\(s\) of (\d+) file", r.stdout)
    assert m and int(m.group(1)) >= 6 and int(m.group(2)) >= 2   # both libraries, three code objects each


def test_toolchain_hazard_table_is_what_the_guard_assumes(tmp_path):
    """tools/ubench/store_hazard_probe.hip, compile only: for a 16-byte buffer store whose data the next VALU instruction
    overwrites, hipcc inserts two wait states on gfx950 when soffset is an immediate and NOTHING when it is an SGPR -- the
    exempted form the kernels guard by hand with `s_nop 1`.  If a toolchain changes either fact, the guard's reasoning
    (profiles/r04/store_hazard.md) needs another look."""
    import subprocess
    out = tmp_path / "probe.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "tools", "ubench", "store_hazard_probe.hip")], stderr=subprocess.DEVNULL)
    text = out.read_text()

    def after_first_store(kernel):
        body = text[text.index(kernel + ":"):]
        lines = [l.strip() for l in body[:body.index("s_endpgm")].split("\n")]
        i = next(k for k, l in enumerate(lines) if l.startswith("buffer_store_dwordx4"))
        return lines[i], lines[i + 1]
    st, nxt = after_first_store("_Z5k_immPhiPKj")
    assert st.split(",")[-1].strip().startswith("0 ") and nxt == "s_nop 1", (st, nxt)
    st, nxt = after_first_store("_Z6k_sgprPhiPKji")
    assert re.search(r", s\d+ offen", st) and not nxt.startswith("s_nop"), (st, nxt)


def test_staging_crew_on_cpu_plain_and_under_thread_sanitizer():
    """csrc/host_crew.h, the host threads that stage a large pageable frame of hevc_deblocking_filter: tests/host_crew/crew_test.cpp
    runs strip-shaped copies (tight and pitched rows, 600 jobs through the 256-slot ring, 0..7 crew threads, streaming stores on
    and off) and checks every byte, once as a plain build and once under -fsanitize=thread."""
    import subprocess
    d = os.path.join(ROOT, "tests", "host_crew")
    subprocess.check_call(["make", "-s", "-C", d])
    r = subprocess.run([os.path.join(d, "crew_test"), "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 failing rounds" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([os.path.join(d, "crew_test_tsan"), "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 failing rounds" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]


def test_host_side_entries_reject_bad_arguments_without_a_device(L):
    """The round-4 host-side entries take a context; with none they return HEVCDBK_ERR_ARG (no crash, no CPU path)."""
    from gpu_video_codec_amd import _lib
    n = C.c_uint(7)
    assert L.hevcdbk_set_host_threads(None, 4) == _lib.ERR_ARG
    assert L.hevcdbk_get_host_threads(None) == 0
    assert L.hevcdbk_host_register(None, None, 0) == _lib.ERR_ARG
    assert L.hevcdbk_host_unregister(None, None) == _lib.ERR_ARG
    assert L.hevcdbk_last_frame_trace(None, None, 0, C.byref(n)) == _lib.ERR_ARG
    p = C.c_void_p(5)
    assert L.hevcdbk_device_malloc_probed(None, None, 30, None, 4, C.byref(p), None, None) == _lib.ERR_ARG
