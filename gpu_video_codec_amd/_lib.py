"""ctypes loader for the product library libhevcdbk.so (C ABI: include/hevc_deblock.h).

Fails loudly when the library is missing: there is no Python or CPU fallback for the filter.
The library is linked against /opt/rocm's HIP runtime (RUNPATH); loading it before `import torch`
makes that runtime (the one hipcc / rocprofv3 match) the single HIP runtime of the process.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhevcdbk.so")

OK, ERR_FILE_SIZE, ERR_DIMENSIONS, ERR_BS_SIZE, ERR_HIP, ERR_ARG, ERR_NOMEM, ERR_IO, ERR_UNSUPPORTED = \
    0, -1, -2, -3, -4, -5, -6, -7, -8
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_PACKED = 0, 1, 2
MAP_AUTO, MAP_ROWS, MAP_LINEAR = 0x000, 0x100, 0x200   # OR-ed into the kernel selector (HEVCDBK_MAP_*)
# diagnostic library only (csrc/hevcdbk_diag.h; use_diagnostic_library() below): the product library refuses it
DIAG_KERNEL_COPY = 100
DIAG_MAP_STRIPE = 0x300
DIAG_MAP_TILES = 0x400
DIAG_MAP_PIPE = 0x500
DIAG_MAP_GROUP = 0x600
FUSED_AUTO, FUSED_OFF, FUSED_ON = 0, 1, 2   # hevc_deblock_sao_*_device: one kernel for both stages / two launches
DIAG_LIB_PATH = os.path.join(_HERE, "libhevcdbk_diag.so")

# every symbol include/hevc_deblock.h declares (tests check the library exports all of them)
EXPORTS = [
    "hevcdbk_strerror", "hevcdbk_device_count", "hevcdbk_create", "hevcdbk_destroy", "hevcdbk_last_error",
    "hevcdbk_get_device_info", "hevcdbk_default_tc_table", "hevcdbk_default_beta_table",
    "hevcdbk_num_vert_bs", "hevcdbk_num_hor_bs", "hevcdbk_default_bs",
    "hevc_deblocking_filter", "hevc_deblocking_filter_device", "hevc_deblocking_filter_device_planes",
    "hevc_deblocking_filter_sequence",
    "hevcdbk_device_malloc", "hevcdbk_device_free", "hevcdbk_host_malloc_pinned", "hevcdbk_host_free_pinned",
    "hevcdbk_memcpy_h2d", "hevcdbk_memcpy_d2h", "hevcdbk_memcpy_d2d", "hevcdbk_memset_d",
    "hevcdbk_synchronize", "hevcdbk_compute_stream", "hevcdbk_device_run_timed", "hevcdbk_device_replay",
    "hevcdbk_device_pci_bus_id", "hevcdbk_execute_gpu",
    "hevcdbk_filter_yuv_file", "hevcdbk_filter_yuv_file_multi",
    "hevcdbk_h265_num_vert_bs", "hevcdbk_h265_num_hor_bs", "hevcdbk_h265_derive_bs_device",
    "hevc_deblocking_filter_h265_device", "hevc_deblocking_filter_h265", "hevc_sao_filter_device",
    "hevc_deblock_sao_device", "hevc_deblock_sao_h265_device",
    "hevc_deblock_sao_device_planes", "hevc_deblock_sao_h265_device_planes",
    "hevcdbk_set_host_threads", "hevcdbk_get_host_threads", "hevcdbk_host_register", "hevcdbk_host_unregister",
    "hevcdbk_last_frame_trace", "hevcdbk_device_malloc_probed",
]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint), ("height", C.c_uint), ("bit_depth", C.c_uint), ("sample_bytes", C.c_uint),
                ("plane", C.c_void_p * 3), ("pitch", C.c_size_t * 3)]


class Bs(C.Structure):
    _fields_ = [("vert", C.c_void_p), ("n_vert", C.c_size_t), ("hor", C.c_void_p), ("n_hor", C.c_size_t),
                ("chroma_vert", C.c_void_p), ("n_chroma_vert", C.c_size_t),
                ("chroma_hor", C.c_void_p), ("n_chroma_hor", C.c_size_t)]


class Qp(C.Structure):
    _fields_ = [("qp", C.c_uint), ("map", C.c_void_p), ("map_stride", C.c_uint), ("ctu_log2", C.c_uint)]


class Tables(C.Structure):
    _fields_ = [("tc", C.c_void_p), ("beta", C.c_void_p)]


class StripTrace(C.Structure):
    """hevcdbk_strip_trace: one strip of the last large-frame hevc_deblocking_filter call"""
    _fields_ = [("plane", C.c_int), ("row_begin", C.c_uint), ("row_end", C.c_uint), ("bytes", C.c_size_t),
                ("stage_begin_s", C.c_double), ("stage_end_s", C.c_double), ("enqueue_begin_s", C.c_double),
                ("enqueue_end_s", C.c_double), ("d2h_seen_s", C.c_double), ("unstage_begin_s", C.c_double),
                ("unstage_end_s", C.c_double), ("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double)]


class SaoPlane(C.Structure):
    """hevcdbk_sao_plane: the SAO operands of one plane of a multi-plane deblocking + SAO call (device pointers)"""
    _fields_ = [("params", C.c_void_p), ("params_stride", C.c_uint), ("params_frame_stride", C.c_size_t), ("ctb_log2", C.c_uint),
                ("keep", C.c_void_p), ("keep_stride", C.c_uint), ("keep_frame_stride", C.c_size_t)]


class Replay(C.Structure):
    """hevcdbk_replay: in = settle / warm-up / steps, out = what the settle phase did and the timed window's brackets"""
    _fields_ = [("settle_min_ms", C.c_double), ("settle_max_ms", C.c_double), ("settle_tolerance", C.c_double),
                ("settle_window", C.c_uint), ("warmup", C.c_uint), ("steps", C.c_uint),
                ("settle_launches", C.c_uint), ("settled", C.c_int), ("settle_ms", C.c_double),
                ("settle_tail_mean_ms", C.c_double), ("t_begin", C.c_double), ("t_end", C.c_double),
                ("wall_ms", C.c_double), ("span_ms", C.c_double)]


class Timing(C.Structure):
    _fields_ = [("exec_s", C.c_double), ("total_s", C.c_double), ("copy_s", C.c_double), ("pipelined_s", C.c_double)]


class DevicePlanes(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("pitch", C.c_size_t), ("frame_stride", C.c_size_t),
                ("n_frames", C.c_uint), ("plane_w", C.c_uint), ("plane_h", C.c_uint),
                ("bit_depth", C.c_uint), ("sample_bytes", C.c_uint), ("is_chroma", C.c_int),
                ("vert_bs", C.c_void_p), ("hor_bs", C.c_void_p),
                ("vert_bs_stride", C.c_size_t), ("hor_bs_stride", C.c_size_t),
                ("qp_map", C.c_void_p), ("qp_map_stride", C.c_uint), ("ctu_log2", C.c_uint),
                ("qp_map_frame_stride", C.c_size_t)]


class H265Params(C.Structure):
    _fields_ = [("tc_offset_div2", C.c_int), ("beta_offset_div2", C.c_int), ("cb_qp_offset", C.c_int), ("cr_qp_offset", C.c_int)]


class H265Units(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("mv0", C.c_void_p), ("mv1", C.c_void_p), ("ref0", C.c_void_p), ("ref1", C.c_void_p)]


H265_BS_MASK, H265_KEEP_P, H265_KEEP_Q = 3, 4, 8
U_INTRA, U_CBF, U_TU_LEFT, U_TU_TOP, U_PU_LEFT, U_PU_TOP = 1, 2, 4, 8, 16, 32
U_KEEP, U_DBK_OFF, U_PRED_L0, U_PRED_L1, U_NOX_LEFT, U_NOX_TOP = 64, 128, 256, 512, 1024, 2048


SAO_CTB_DTYPE = [("type", "u1"), ("cls", "u1"), ("offset", "i1", (4,))]  # numpy dtype of hevcdbk_sao_ctb


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("gcn_arch", C.c_char * 64), ("compute_units", C.c_int),
                ("wavefront_size", C.c_int), ("max_threads_per_block", C.c_int),
                ("total_global_mem", C.c_size_t), ("shared_mem_per_block", C.c_size_t),
                ("total_const_mem", C.c_size_t)]


_lib = None


def use_diagnostic_library(spec=None):
    """Tools and ablation tests only: make this process load libhevcdbk_diag.so (the same sources built with
    -DHEVCDBK_DIAG: copy variant, timing-only ablations, launch knobs) instead of the product library, and set its
    knobs (hevcdbk_diag_set).  Must be called before the first use of the library; never called by the product path."""
    global LIB_PATH
    if _lib is not None and LIB_PATH != DIAG_LIB_PATH:
        raise RuntimeError("the product library is already loaded in this process")
    LIB_PATH = DIAG_LIB_PATH
    L = lib()
    L.hevcdbk_diag_set.argtypes = [C.c_char_p]
    rc = L.hevcdbk_diag_set(None if spec is None else spec.encode())
    if rc != OK:
        raise ValueError("hevcdbk_diag_set(%r) -> %d" % (spec, rc))
    return L


def lib():
    """The loaded C library.  Raises if it has not been built (see __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libhevcdbk.so is missing (%s): build it with `make -C gpu_video_codec_amd/csrc` or "
                "__graft_entry__.build(); the deblocking filter has no CPU/Python fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L.hevcdbk_strerror.restype = C.c_char_p
        L.hevcdbk_last_error.restype = C.c_char_p
        L.hevcdbk_last_error.argtypes = [C.c_void_p]
        L.hevcdbk_default_tc_table.restype = C.POINTER(C.c_uint * 52)
        L.hevcdbk_default_beta_table.restype = C.POINTER(C.c_uint * 52)
        L.hevcdbk_num_vert_bs.restype = C.c_size_t
        L.hevcdbk_num_hor_bs.restype = C.c_size_t
        L.hevcdbk_num_vert_bs.argtypes = [C.c_uint, C.c_uint]
        L.hevcdbk_num_hor_bs.argtypes = [C.c_uint, C.c_uint]
        L.hevcdbk_default_bs.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
        L.hevcdbk_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.hevcdbk_destroy.argtypes = [C.c_void_p]
        L.hevcdbk_destroy.restype = None
        L.hevcdbk_get_device_info.argtypes = [C.c_void_p, C.POINTER(DeviceInfo)]
        L.hevc_deblocking_filter.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(Bs), C.POINTER(Qp),
                                             C.POINTER(Tables), C.POINTER(Timing)]
        L.hevc_deblocking_filter_sequence.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_uint, C.POINTER(Bs), C.POINTER(Qp),
                                                      C.POINTER(Tables), C.POINTER(Timing)]
        L.hevc_deblocking_filter_device.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint,
                                                    C.POINTER(Tables), C.c_int, C.c_void_p]
        L.hevc_deblocking_filter_device_planes.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint, C.c_uint,
                                                           C.POINTER(Tables), C.c_int, C.c_void_p]
        L.hevcdbk_device_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.hevcdbk_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.hevcdbk_host_malloc_pinned.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.hevcdbk_host_free_pinned.argtypes = [C.c_void_p, C.c_void_p]
        L.hevcdbk_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.hevcdbk_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.hevcdbk_memcpy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.hevcdbk_memset_d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]
        L.hevcdbk_synchronize.argtypes = [C.c_void_p]
        L.hevcdbk_compute_stream.argtypes = [C.c_void_p]
        L.hevcdbk_compute_stream.restype = C.c_void_p
        L.hevcdbk_device_run_timed.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint, C.c_uint,
                                               C.POINTER(Tables), C.c_int, C.c_uint, C.POINTER(C.c_float)]
        L.hevcdbk_device_replay.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint, C.c_uint,
                                            C.POINTER(Tables), C.c_int, C.POINTER(Replay), C.POINTER(C.c_float)]
        L.hevcdbk_device_pci_bus_id.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.hevcdbk_execute_gpu.argtypes = [C.c_char_p, C.c_char_p, C.c_uint, C.c_uint, C.c_uint,
                                          C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int]
        L.hevcdbk_filter_yuv_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint, C.c_uint, C.c_uint,
                                              C.POINTER(Bs), C.POINTER(Tables), C.POINTER(C.c_uint), C.POINTER(Timing)]
        L.hevcdbk_filter_yuv_file_multi.argtypes = [C.POINTER(C.c_int), C.c_uint, C.c_char_p, C.c_char_p, C.c_uint, C.c_uint,
                                                    C.c_uint, C.POINTER(Bs), C.POINTER(Tables), C.POINTER(C.c_uint),
                                                    C.POINTER(Timing)]
        L.hevcdbk_h265_num_vert_bs.restype = C.c_size_t
        L.hevcdbk_h265_num_hor_bs.restype = C.c_size_t
        L.hevcdbk_h265_num_vert_bs.argtypes = [C.c_uint, C.c_uint]
        L.hevcdbk_h265_num_hor_bs.argtypes = [C.c_uint, C.c_uint]
        L.hevcdbk_h265_derive_bs_device.argtypes = [C.c_void_p, C.POINTER(H265Units), C.c_uint, C.c_uint, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hevc_deblocking_filter_h265_device.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_int, C.c_uint,
                                                         C.POINTER(H265Params), C.c_int, C.c_void_p]
        L.hevc_deblocking_filter_h265.argtypes = [C.c_void_p, C.POINTER(Frame), C.POINTER(H265Units), C.POINTER(Bs),
                                                  C.POINTER(Qp), C.POINTER(H265Params), C.POINTER(Timing)]
        L.hevc_sao_filter_device.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_void_p, C.c_uint, C.c_size_t, C.c_uint,
                                             C.c_void_p, C.c_uint, C.c_size_t, C.c_void_p]
        L.hevc_deblock_sao_device.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint, C.POINTER(Tables), C.c_void_p, C.c_uint,
                                              C.c_size_t, C.c_uint, C.c_void_p, C.c_uint, C.c_size_t, C.c_int, C.c_void_p]
        L.hevc_deblock_sao_h265_device.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_int, C.c_uint, C.POINTER(H265Params),
                                                   C.c_void_p, C.c_uint, C.c_size_t, C.c_uint, C.c_void_p, C.c_uint, C.c_size_t,
                                                   C.c_int, C.c_void_p]
        L.hevc_deblock_sao_device_planes.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint, C.c_uint, C.POINTER(Tables),
                                                     C.POINTER(SaoPlane), C.c_int, C.c_void_p]
        L.hevc_deblock_sao_h265_device_planes.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint, C.c_uint, C.POINTER(H265Params),
                                                          C.POINTER(SaoPlane), C.c_int, C.c_void_p]
        L.hevcdbk_set_host_threads.argtypes = [C.c_void_p, C.c_uint]
        L.hevcdbk_get_host_threads.argtypes = [C.c_void_p]
        L.hevcdbk_get_host_threads.restype = C.c_uint
        L.hevcdbk_host_register.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.hevcdbk_host_unregister.argtypes = [C.c_void_p, C.c_void_p]
        L.hevcdbk_last_frame_trace.argtypes = [C.c_void_p, C.POINTER(StripTrace), C.c_uint, C.POINTER(C.c_uint)]
        L.hevcdbk_device_malloc_probed.argtypes = [C.c_void_p, C.POINTER(DevicePlanes), C.c_uint, C.POINTER(Tables), C.c_uint,
                                                   C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = L
    return _lib


class DeblockError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().hevcdbk_strerror(code).decode()
        super().__init__("%s (code %d)%s" % (msg, code, (": " + detail) if detail else ""))
