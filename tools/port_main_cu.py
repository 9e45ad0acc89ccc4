#!/usr/bin/env python3
"""The edits INTEGRATION.md section 1 lists, applied mechanically: turns a copy of the reference's driver
(hevc_deblocking_filter/main.cu) into a plain C++ translation unit that links against libhevcdbk.so instead of the
reference's CUDA translation unit.

    python3 tools/port_main_cu.py <path/to/main.cu> <out.cpp>

Nothing of the reference is stored in this repository: the script reads the file where the maintainer has it and writes
the patched copy where they say.  Edits (line numbers of the reference snapshot this was written against):
  1. main.cu:20-21, 27-28   drop the four CUDA includes (cuda_runtime.h, device_launch_parameters.h, each twice);
  2. main.cu:85             drop the stray `addWithCuda` declaration (a CUDA template leftover, never defined or called);
  3. main.cu:92-107         GetGpuDeviceInfo(): same printout from hevcdbk_get_device_info() instead of
                            cudaGetDeviceProperties();
  4. add `#include "hevc_deblock.h"` (for edit 3).
ExecuteCpu (main.cu:36-83), the forward declaration of ExecuteGpu (main.cu:87-90) and main() (main.cu:109-141) stay
byte for byte: ExecuteGpu resolves against the C++ symbol the library exports (csrc/execute_gpu_shim.cpp).
"""
import re
import sys

NEW_INFO = r'''void GetGpuDeviceInfo() {
	hevcdbk_context *ctx = NULL;
	hevcdbk_device_info i;
	if (hevcdbk_create(0, &ctx) != HEVCDBK_OK || hevcdbk_get_device_info(ctx, &i) != HEVCDBK_OK) {
		printf("no HIP device\n");
		if (ctx) hevcdbk_destroy(ctx);
		return;
	}
	printf("==============================================\n");
	printf("Device %d: %s\n", 0, i.name);
	printf("Number of multiprocessors: %d\n", i.compute_units);
	printf("Total amount of constant memory: %4.2f KB\n", i.total_const_mem / 1024.0);
	printf("Total amount of global memory: %4.2f KB\n", i.total_global_mem / 1024.0);
	printf("Total amount of shared memory per block: %4.2f KB\n", i.shared_mem_per_block / 1024.0);
	printf("Warp size: %d\n", i.wavefront_size);
	printf("Maximum number of threads per block: %d\n", i.max_threads_per_block);
	printf("==============================================\n\n\n");
	hevcdbk_destroy(ctx);
}
'''


def port(src):
    out, n_inc = [], 0
    lines = src.split("\n")
    i = 0
    edits = {"includes": 0, "addWithCuda": 0, "GetGpuDeviceInfo": 0}
    while i < len(lines):
        l = lines[i]
        if re.match(r'\s*#include\s+"(cuda_runtime|device_launch_parameters)\.h"', l):
            edits["includes"] += 1
            i += 1
            continue
        if re.match(r"\s*cudaError_t\s+addWithCuda\s*\(", l):
            edits["addWithCuda"] += 1
            i += 1
            continue
        if re.match(r"\s*void\s+GetGpuDeviceInfo\s*\(\s*\)\s*\{", l):
            depth = 0
            while i < len(lines):  # skip the old body: to the brace that closes the function
                depth += lines[i].count("{") - lines[i].count("}")
                i += 1
                if depth == 0:
                    break
            out.append(NEW_INFO.rstrip("\n"))
            edits["GetGpuDeviceInfo"] += 1
            continue
        if n_inc == 0 and re.match(r'\s*#include\s+"hevc_deblocking_filter_cpu\.h"', l):
            out.append('#include "hevc_deblock.h"')
            n_inc = 1
        out.append(l)
        i += 1
    if edits["includes"] != 4 or edits["addWithCuda"] != 1 or edits["GetGpuDeviceInfo"] != 1 or n_inc != 1:
        raise SystemExit("port_main_cu: this main.cu does not look like the snapshot the edit list was written for: %s" % edits)
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    with open(sys.argv[1], encoding="utf-8", errors="replace") as fh:
        text = fh.read()
    with open(sys.argv[2], "w", encoding="utf-8") as fh:
        fh.write(port(text))
