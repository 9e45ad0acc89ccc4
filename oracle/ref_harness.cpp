/*
 * ref_harness.cpp -- thin C entry points around the REFERENCE's own CPU implementation.
 *
 * TEST INFRASTRUCTURE ONLY.  This translation unit #includes the reference header where it
 * lies (/root/reference/hevc_deblocking_filter/hevc_deblocking_filter_cpu.h, never copied)
 * and is built by oracle/Makefile into oracle/_ref/ (git-ignored, but it travels to the GPU
 * box as a built binary).  It pins the restatement in deblock_oracle.c and is the
 * cpu_baseline of kind "reference" in bench.py.  Recipe = SURVEY.md Appendix A.
 *
 * The replacement allocation functions below make `new unsigned char[n]` return zeroed
 * memory, which pins the reference's never-initialised plane padding to 0 (the
 * natural build gives the same bytes in a fresh process), and route every delete through
 * free() because the reference releases new[] memory through unique_ptr<unsigned char>
 * (cpu.h:57,1041).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

void *operator new[](std::size_t n)
{
    void *p = std::calloc(n ? n : 1, 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(std::size_t n)
{
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void operator delete(void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

#include "/root/reference/hevc_deblocking_filter/hevc_deblocking_filter_cpu.h"

extern "C" {

/* error codes shared with deblock_oracle.h */
enum { REF_OK = 0, REF_ERR_FILE_SIZE = -1, REF_ERR_DIMENSIONS = -2, REF_ERR_BS_SIZE = -3, REF_ERR_OTHER = -9 };

static int map_throw(const char *m)
{
    if (std::strstr(m, "file size")) return REF_ERR_FILE_SIZE;
    if (std::strstr(m, "multiplier")) return REF_ERR_DIMENSIONS;
    if (std::strstr(m, "boundary strenght")) return REF_ERR_BS_SIZE;
    return REF_ERR_OTHER;
}

/* ReadYuvFrame(file, w, h, Qp) -- cpu.h:35 */
int ref_frame_create(void **out, const char *path, unsigned w, unsigned h, unsigned qp)
{
    try {
        *out = new ReadYuvFrame(path, w, h, qp);
        return REF_OK;
    } catch (const char *m) {
        *out = nullptr;
        return map_throw(m);
    }
}

/* SetBoundaryStrenght -- cpu.h:120 */
int ref_frame_set_bs(void *f, const unsigned char *vert, unsigned n_vert, const unsigned char *hor, unsigned n_hor)
{
    try {
        static_cast<ReadYuvFrame *>(f)->SetBoundaryStrenght(const_cast<unsigned char *>(vert), n_vert,
                                                            const_cast<unsigned char *>(hor), n_hor);
        return REF_OK;
    } catch (const char *m) {
        return map_throw(m);
    }
}

/* DeblockingFilter(num_threads) -- cpu.h:134 */
int ref_frame_filter(void *f, unsigned num_threads)
{
    try {
        static_cast<ReadYuvFrame *>(f)->DeblockingFilter(num_threads);
        return REF_OK;
    } catch (const char *m) {
        return map_throw(m);
    }
}

/* Save -- cpu.h:995 */
int ref_frame_save(void *f, const char *path)
{
    static_cast<ReadYuvFrame *>(f)->Save(path);
    return REF_OK;
}

void ref_frame_destroy(void *f) { delete static_cast<ReadYuvFrame *>(f); }

} /* extern "C" */

#ifdef REF_HARNESS_MAIN
/* CLI: ref_oracle in w h qp threads out [seed]   (seed => LCG bS in {0,1,2}, vert then hor) */
int main(int argc, char **argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s in w h qp threads out [bs_seed]\n", argv[0]);
        return 2;
    }
    unsigned w = std::atoi(argv[2]), h = std::atoi(argv[3]), qp = std::atoi(argv[4]), nt = std::atoi(argv[5]);
    void *f = nullptr;
    int rc = ref_frame_create(&f, argv[1], w, h, qp);
    if (rc) { std::fprintf(stderr, "create failed: %d\n", rc); return 1; }
    if (argc > 7) {
        unsigned s = (unsigned)std::strtoul(argv[7], nullptr, 10);
        unsigned nv = (w / 8 + 1) * h / 8, nh = (h / 8 + 1) * w / 8;
        unsigned char *v = new unsigned char[nv], *hh = new unsigned char[nh];
        for (unsigned i = 0; i < nv; i++) { s = s * 1664525u + 1013904223u; v[i] = (s >> 16) % 3; }
        for (unsigned i = 0; i < nh; i++) { s = s * 1664525u + 1013904223u; hh[i] = (s >> 16) % 3; }
        rc = ref_frame_set_bs(f, v, nv, hh, nh);
        delete[] v; delete[] hh;
        if (rc) { std::fprintf(stderr, "set_bs failed: %d\n", rc); return 1; }
    }
    ref_frame_filter(f, nt);
    ref_frame_save(f, argv[6]);
    ref_frame_destroy(f);
    return 0;
}
#endif
