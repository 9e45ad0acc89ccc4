/*
 * hevc_deblock_main.cpp -- main.cu-shaped driver (main.cu:109-141) on top of libhevcdbk.so.
 *
 *   hevc_deblock_main [in.yuv out.yuv width height qp [device]]
 *   hevc_deblock_main --sequence in.yuv out.yuv width height qp [device[,device...]]   (any number of frames in the file;
 *                                                                                    several devices = frame-parallel shard)
 *
 * Without arguments it runs the configuration main.cu ships with (main.cu:128-133:
 * mother-daughter 352x288, QP 35).  Prints the GetGpuDeviceInfo block (main.cu:92-107) and the
 * reference's three GPU timing lines.  The CPU legs of main.cu (ExecuteCpu, main.cu:36-83) stay
 * with the reference's own header; this library has no CPU path.
 */
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/hevc_deblock.h"

void ExecuteGpu(std::string const &input_file_name, std::string const &output_file_name,
                unsigned int width, unsigned int height, unsigned int Qp,
                unsigned dimx1, unsigned int dimy1, unsigned dimx2, unsigned int dimy2);

static void GetGpuDeviceInfo(int device)
{
    hevcdbk_context *ctx = nullptr;
    if (hevcdbk_create(device, &ctx) != HEVCDBK_OK) {
        std::printf("No usable HIP device %d\n", device);
        return;
    }
    hevcdbk_device_info i;
    if (hevcdbk_get_device_info(ctx, &i) == HEVCDBK_OK) {
        std::printf("==============================================\n");
        std::printf("Device %d: %s (%s)\n", device, i.name, i.gcn_arch);
        std::printf("Number of compute units: %d\n", i.compute_units);
        std::printf("Total amount of constant memory: %4.2f KB\n", i.total_const_mem / 1024.0);
        std::printf("Total amount of global memory: %4.2f KB\n", i.total_global_mem / 1024.0);
        std::printf("Total amount of shared memory per block: %4.2f KB\n", i.shared_mem_per_block / 1024.0);
        std::printf("Wavefront size: %d\n", i.wavefront_size);
        std::printf("Maximum number of threads per block: %d\n", i.max_threads_per_block);
        std::printf("==============================================\n\n\n");
    }
    hevcdbk_destroy(ctx);
}

int main(int argc, char **argv)
{
    std::string in = "mother-daughter_352x288_yv12.yuv";
    std::string out = "mother-daughter_352x288_yv12_filtered_gpu.yuv";
    unsigned width = 352, height = 288, Qp = 35;
    int device = 0;
    if (argc >= 7 && std::string(argv[1]) == "--sequence") {
        /* beyond main.cu: a multi-frame file through the streaming pipeline (hevcdbk_filter_yuv_file) */
        /* optional last argument: one device, or a comma-separated list = shard the file over those GPUs */
        std::vector<int> devs;
        if (argc >= 8) {
            for (const char *p = argv[7]; *p;) {
                devs.push_back(std::atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
        if (devs.empty()) devs.push_back(0);
        unsigned n = 0;
        hevcdbk_timing t;
        const int rc = hevcdbk_filter_yuv_file_multi(devs.data(), (unsigned)devs.size(), argv[2], argv[3], (unsigned)std::atoi(argv[4]),
                                                     (unsigned)std::atoi(argv[5]), (unsigned)std::atoi(argv[6]), nullptr, nullptr,
                                                     &n, &t);
        if (rc != HEVCDBK_OK) {
            std::fprintf(stderr, "error: %s\n", hevcdbk_strerror(rc));
            return 1;
        }
        std::printf("Frames: %u on %zu device worker(s)\nExecution Time with file I/O on GPU: %gs (%g frames/s)\n", n, devs.size(),
                    t.pipelined_s, t.pipelined_s > 0 ? n / t.pipelined_s : 0.0);
        return 0;
    }
    if (argc >= 6) {
        in = argv[1]; out = argv[2];
        width = (unsigned)std::atoi(argv[3]); height = (unsigned)std::atoi(argv[4]); Qp = (unsigned)std::atoi(argv[5]);
        if (argc >= 7) device = std::atoi(argv[6]);
    } else if (argc != 1) {
        std::fprintf(stderr, "usage: %s [in.yuv out.yuv width height qp [device]]\n", argv[0]);
        return 2;
    }
    GetGpuDeviceInfo(device);
    try {
        if (device == 0) {
            ExecuteGpu(in, out, width, height, Qp, 20, 20, 20, 20); /* main.cu:138 */
        } else {
            const int rc = hevcdbk_execute_gpu(in.c_str(), out.c_str(), width, height, Qp, 20, 20, 20, 20, device);
            if (rc) throw hevcdbk_strerror(rc);
        }
    } catch (const char *m) {
        std::fprintf(stderr, "error: %s\n", m);
        return 1;
    }
    return 0;
}
