// read_yuv_frame.cpp -- the body of the reference's ExecuteCpu (main.cu:36-83) against include/hevc_deblock.hpp: the same four
// calls, the same catch of `const char *`, executed on the MI355X.
//   read_yuv_frame in.yuv out.yuv width height qp [bs_seed]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hevc_deblock.hpp"

int main(int argc, char **argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s in.yuv out.yuv width height qp [bs_seed]\n", argv[0]);
        return 2;
    }
    const unsigned w = (unsigned)std::atoi(argv[3]), h = (unsigned)std::atoi(argv[4]), qp = (unsigned)std::atoi(argv[5]);
    try {
        hevcdbk::ReadYuvFrame frame(argv[1], w, h, qp);
        if (argc >= 7) { /* the seeded bS generator of BASELINE config 3: s = s*1664525 + 1013904223, value (s >> 16) % 3 */
            uint32_t s = (uint32_t)std::atoi(argv[6]);
            std::vector<unsigned char> vert(hevcdbk_num_vert_bs(w, h)), hor(hevcdbk_num_hor_bs(w, h));
            for (auto &b : vert) { s = s * 1664525u + 1013904223u; b = (unsigned char)((s >> 16) % 3); }
            for (auto &b : hor) { s = s * 1664525u + 1013904223u; b = (unsigned char)((s >> 16) % 3); }
            frame.SetBoundaryStrenght(vert.data(), (unsigned)vert.size(), hor.data(), (unsigned)hor.size());
        }
        frame.DeblockingFilter(1);
        std::printf("Execution Time without copy on GPU: %gs\n", frame.timing().exec_s);
        frame.Save(argv[2]);
    } catch (const char *m) {
        std::fprintf(stderr, "error: %s\n", m);
        return 1;
    }
    return 0;
}
