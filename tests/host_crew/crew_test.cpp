// tests/host_crew/crew_test.cpp -- CPU test of csrc/host_crew.h, the staging crew of the host-frame operator: many rounds of
// strip-shaped copies (tight and pitched rows, more jobs than the ring holds, 0..7 crew threads), every byte checked, built
// plain and with -fsanitize=thread by tests/test_abi_cpu.py.  No HIP here: the crew never makes a HIP call.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../../gpu_video_codec_amd/csrc/host_crew.h"

using namespace dbkh;

static uint32_t lcg(uint32_t &s) { return s = s * 1664525u + 1013904223u; }

static int round_of(StageCrew &crew, unsigned W, unsigned H, size_t user_pitch, unsigned strips, unsigned pieces, bool back, uint32_t seed, bool to_device = false)
{
    std::vector<uint8_t> user(user_pitch * H), ring((size_t)W * H, 0xEE), want;
    uint32_t s = seed;
    for (auto &b : user) b = (uint8_t)(lcg(s) >> 24);
    want = user;
    std::unique_ptr<CopyGroup[]> g(new CopyGroup[strips]);
    crew.begin();
    for (unsigned k = 0; k < strips; k++) {
        const unsigned r0 = H * k / strips, r1 = H * (k + 1) / strips, rows = r1 - r0;
        const unsigned pc = rows < pieces ? (rows ? rows : 1) : pieces;
        g[k].pending.store((int)pc);
        for (unsigned p = 0; p < pc; p++) {
            const unsigned a = r0 + rows * p / pc, b = r0 + rows * (p + 1) / pc;
            crew.submit({ring.data() + (size_t)a * W, user.data() + a * user_pitch, W, user_pitch, W, b - a, &g[k], to_device});
        }
    }
    for (unsigned k = 0; k < strips; k++) crew.wait(g[k]);
    int bad = 0;
    for (unsigned y = 0; y < H && !bad; y++)
        for (unsigned x = 0; x < W; x++)
            if (ring[(size_t)y * W + x] != user[y * user_pitch + x]) { bad = 1; break; }
    if (back) { /* the way out: ring -> caller plane; bytes between the rows of a pitched plane stay as they were */
        for (auto &b : ring) b = (uint8_t)(b ^ 0x5A);
        std::unique_ptr<CopyGroup[]> o(new CopyGroup[strips]);
        for (unsigned k = 0; k < strips; k++) {
            const unsigned r0 = H * k / strips, r1 = H * (k + 1) / strips;
            o[k].pending.store(1);
            crew.submit({user.data() + r0 * user_pitch, ring.data() + (size_t)r0 * W, user_pitch, W, W, r1 - r0, &o[k]}, StageCrew::LANE_OUT);
        }
        for (unsigned k = 0; k < strips; k++) crew.wait(o[k]);
        for (unsigned y = 0; y < H && !bad; y++)
            for (size_t x = 0; x < user_pitch; x++) {
                const uint8_t w = x < W ? (uint8_t)(want[y * user_pitch + x] ^ 0x5A) : want[y * user_pitch + x];
                if (user[y * user_pitch + x] != w) { bad = 2; break; }
            }
        for (unsigned k = 0; k < strips; k++)
            if (o[k].done_ns.load() == 0 || o[k].first_ns.load() == 0 || o[k].done_ns.load() < o[k].first_ns.load()) bad = bad ? bad : 3;
    }
    crew.end();
    return bad;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 6;
    int fails = 0;
    for (unsigned workers : {0u, 1u, 3u, 7u})
        for (int stream = 0; stream < 2; stream++) {
            StageCrew crew(workers, stream != 0);
            for (int r = 0; r < rounds; r++) {
                /* a 1920x1080-like plane in tight rows (block copies, streaming stores above 256 KiB), a pitched one, and a round
                 * of 600 single-row jobs: more than the ring's 256 slots, so the producer has to work through a full ring */
                fails += round_of(crew, 1920, 536, 1920, 5, workers + 1, true, 11u + (uint32_t)r) != 0;
                fails += round_of(crew, 720, 288, 736, 7, 3, true, 23u + (uint32_t)r) != 0;
                fails += round_of(crew, 64, 600, 80, 600, 1, r & 1, 37u + (uint32_t)r) != 0;
                /* the form used for HBM behind the PCIe BAR (streaming stores row by row, flush read), rows that start off 16-byte
                 * boundaries: 1000-byte rows, tight and pitched */
                fails += round_of(crew, 1000, 300, 1000, 4, workers + 1, false, 41u + (uint32_t)r, true) != 0;
                fails += round_of(crew, 1000, 300, 1016, 4, 2, false, 43u + (uint32_t)r, true) != 0;
            }
        }
    std::printf("host_crew: %d failing rounds\n", fails);
    return fails ? 1 : 0;
}
