/*
 * hevc_deblock.hpp -- header-only C++ mirror of the reference's class surface on top of the C ABI (hevc_deblock.h).
 *
 * A caller written against the reference's
 *     ReadYuvFrame frame(file, w, h, Qp);                         cpu.h:35
 *     frame.SetBoundaryStrenght(vert, n_vert, hor, n_hor);        cpu.h:120
 *     frame.DeblockingFilter(num_threads);                        cpu.h:134
 *     frame.Save(out);                                            cpu.h:995
 * keeps compiling and keeps its error handling (the reference throws `const char *` with the three messages of
 * cpu.h:43-48, 122-123) when it switches `#include "hevc_deblocking_filter_cpu.h"` for this header and the namespace
 * hevcdbk:: -- the work then runs on the MI355X.  Differences, all forced by the hardware boundary:
 *   - DeblockingFilter's argument is accepted and ignored (there are no CPU threads to set; cpu.h:135 sets OpenMP's);
 *   - planes are kept un-padded (the reference's 4-sample zero padding is implicit in the kernels);
 *   - HIP failures, which the reference can only print (gpu.cu:1271-1288), throw the library's error text as well.
 * Not thread-safe per object, like the reference; one context per object unless one is passed in.
 */
#ifndef HEVC_DEBLOCK_HPP
#define HEVC_DEBLOCK_HPP

#include <cstdio>
#include <vector>

#include "hevc_deblock.h"

namespace hevcdbk {

class ReadYuvFrame {
public:
    /* cpu.h:35-118: load a planar 8-bit 4:2:0 file; same checks in the same order */
    ReadYuvFrame(char const *file_name, unsigned int width, unsigned int height, unsigned int Qp = 20,
                 hevcdbk_context *shared_ctx = nullptr, int device = 0)
        : _width(width), _height(height), _Qp(Qp), _ctx(shared_ctx), _own_ctx(shared_ctx == nullptr)
    {
        std::FILE *fp = std::fopen(file_name, "rb");
        long length = -1;
        if (fp) {
            std::fseek(fp, 0, SEEK_END);
            length = std::ftell(fp);
            std::fseek(fp, 0, SEEK_SET);
        }
        if (length < 0 || (unsigned long)length != 3ul * width * height / 2) {
            if (fp) std::fclose(fp);
            throw hevcdbk_strerror(HEVCDBK_ERR_FILE_SIZE); /* "Incorrect file size" */
        }
        if (width % 8 != 0 || height % 8 != 0) {
            std::fclose(fp);
            throw hevcdbk_strerror(HEVCDBK_ERR_DIMENSIONS);
        }
        _buf.resize((size_t)length);
        const size_t got = std::fread(_buf.data(), 1, _buf.size(), fp);
        std::fclose(fp);
        if (got != _buf.size()) throw hevcdbk_strerror(HEVCDBK_ERR_IO);
        if (_own_ctx) check(hevcdbk_create(device, &_ctx));
    }
    ~ReadYuvFrame()
    {
        if (_own_ctx && _ctx) hevcdbk_destroy(_ctx);
    }
    ReadYuvFrame(const ReadYuvFrame &) = delete;
    ReadYuvFrame &operator=(const ReadYuvFrame &) = delete;

    /* cpu.h:120-132: luma bS only, sizes checked, arrays copied */
    void SetBoundaryStrenght(unsigned char *vert_bs, unsigned int num_vert_bs, unsigned char *hor_bs, unsigned int num_hor_bs)
    {
        if (hevcdbk_num_hor_bs(_width, _height) != num_hor_bs || hevcdbk_num_vert_bs(_width, _height) != num_vert_bs)
            throw hevcdbk_strerror(HEVCDBK_ERR_BS_SIZE);
        _vert.assign(vert_bs, vert_bs + num_vert_bs);
        _hor.assign(hor_bs, hor_bs + num_hor_bs);
    }

    /* cpu.h:134-993 */
    void DeblockingFilter(unsigned int /*num_threads*/ = 1)
    {
        const size_t ysz = (size_t)_width * _height, csz = ysz / 4;
        hevcdbk_frame fr = {_width, _height, 8, 1, {_buf.data(), _buf.data() + ysz, _buf.data() + ysz + csz},
                            {_width, _width / 2, _width / 2}};
        hevcdbk_bs bs = {_vert.empty() ? nullptr : _vert.data(), _vert.size(), _hor.empty() ? nullptr : _hor.data(), _hor.size(),
                         nullptr, 0, nullptr, 0};
        hevcdbk_qp qp = {_Qp, nullptr, 0, 6};
        check(hevc_deblocking_filter(_ctx, &fr, &bs, &qp, nullptr, &_timing));
    }

    /* cpu.h:995-1018: Y, U, V interiors in that order */
    void Save(char const *output_file_name)
    {
        std::FILE *fo = std::fopen(output_file_name, "wb");
        if (!fo) throw hevcdbk_strerror(HEVCDBK_ERR_IO);
        const size_t put = std::fwrite(_buf.data(), 1, _buf.size(), fo);
        if (std::fclose(fo) != 0 || put != _buf.size()) throw hevcdbk_strerror(HEVCDBK_ERR_IO);
    }

    /* what ExecuteGpu prints (gpu.cu:1292, 1302-1303), for the last DeblockingFilter() */
    const hevcdbk_timing &timing() const { return _timing; }

private:
    void check(int rc)
    {
        if (rc == HEVCDBK_OK) return;
        /* The reference throws `const char *` (cpu.h:43-48), so this does too.  The HIP detail text lives in the context,
         * which ~ReadYuvFrame destroys while the exception unwinds: copy it into storage that outlives the object (one
         * buffer per thread) before throwing; hevcdbk_strerror() strings are static. */
        if (rc == HEVCDBK_ERR_HIP && _ctx && hevcdbk_last_error(_ctx)[0]) {
            static thread_local char detail[512];
            std::snprintf(detail, sizeof(detail), "%s", hevcdbk_last_error(_ctx));
            throw static_cast<const char *>(detail);
        }
        throw hevcdbk_strerror(rc);
    }
    unsigned int _width, _height, _Qp;
    hevcdbk_context *_ctx;
    bool _own_ctx;
    std::vector<unsigned char> _buf, _vert, _hor;
    hevcdbk_timing _timing = {0, 0, 0, 0};
};

} /* namespace hevcdbk */
#endif
