/*
 * execute_gpu_shim.cpp -- the C++ symbol main.cu links against.
 *
 * main.cu forward-declares ExecuteGpu (main.cu:87-90) and calls it at main.cu:138; the reference
 * defines it in hevc_deblocking_filter_gpu.cu:1230-1232.  Linking main.cu against libhevcdbk.so
 * instead of gpu.cu resolves the same symbol to this definition, which forwards to the C ABI.
 * Errors surface the way the reference's do: a thrown `const char *` (gpu.cu:1082-1087).
 */
#include <string>

#include "../../include/hevc_deblock.h"

HEVCDBK_API void ExecuteGpu(std::string const &input_file_name, std::string const &output_file_name,
                unsigned int width, unsigned int height, unsigned int Qp,
                unsigned dimx1, unsigned int dimy1, unsigned dimx2, unsigned int dimy2)
{
    const int rc = hevcdbk_execute_gpu(input_file_name.c_str(), output_file_name.c_str(), width, height, Qp,
                                       dimx1, dimy1, dimx2, dimy2, /*device=*/0 /* main.cu:93 */);
    if (rc != HEVCDBK_OK) throw hevcdbk_strerror(rc);
}
