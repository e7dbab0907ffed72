// svt_hip_picture.hip — picture input, device side: the layout of a frame in the encoder's padded picture buffers,
// border generation and the HME decimations (kernel_picture.h).  The y4m header / frame reader is host-only code: csrc/y4m_reader.cpp.
#include "host_common.h"
#include "kernel_picture.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace svtdev;
using namespace svthost;

// ---- device side ---------------------------------------------------------------------------------------------------
extern "C" int svt_hip_picture_import(const void* d_frame, uint32_t width, uint32_t height, int ss_x, int ss_y, int is_16bit,
                                      void* d_y, uint32_t stride_y, void* d_cb, uint32_t stride_cb, void* d_cr, uint32_t stride_cr,
                                      uint32_t origin_x, uint32_t origin_y, uint32_t pad_right, uint32_t pad_bottom, void* stream) {
    if (int rc = require_init()) return rc;
    if (!d_frame || !d_y) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return set_err(SVT_HIP_ERR_INVALID, "picture %ux%u", width, height);
    if ((ss_x | ss_y) & ~1) return set_err(SVT_HIP_ERR_INVALID, "subsampling %d,%d", ss_x, ss_y);
    if ((d_cb == nullptr) != (d_cr == nullptr)) return set_err(SVT_HIP_ERR_INVALID, "one chroma buffer without the other");
    PicImport d;
    memset(&d, 0, sizeof(d));
    d.nplanes = d_cb ? 3 : 1;
    const uint32_t cw = (width + (uint32_t)ss_x) >> ss_x, ch = (height + (uint32_t)ss_y) >> ss_y;
    uint32_t row = 0, max_w = 0;
    for (uint32_t i = 0; i < d.nplanes; i++) {
        PicPlane& p = d.p[i];
        const int sx = i ? ss_x : 0, sy = i ? ss_y : 0;
        p.buf = i == 0 ? d_y : (i == 1 ? d_cb : d_cr);
        p.stride = i == 0 ? stride_y : (i == 1 ? stride_cb : stride_cr);
        p.w = i ? cw : width; p.h = i ? ch : height;
        p.ox = origin_x >> sx; p.oy = origin_y >> sy;
        // PadPictureToMultipleOfMinCuSizeDimensions (EbPictureAnalysisProcess.c:4818): pad_right >> subsampling_x for chroma
        p.full_w = p.w + (pad_right >> sx) + 2 * p.ox;
        p.full_h = p.h + (pad_bottom >> sy) + 2 * p.oy;
        if (p.stride < p.full_w) return set_err(SVT_HIP_ERR_INVALID, "plane %u: stride %u < %u", i, p.stride, p.full_w);
        p.src_off = i == 0 ? 0 : width * height + (i - 1) * cw * ch;
        p.row0 = row;
        row += p.full_h;
        if (p.full_w > max_w) max_w = p.full_w;
    }
    d.rows_total = row;
    if (row > 65535) return set_err(SVT_HIP_ERR_INVALID, "%u buffer rows in one launch", row);
    const uint32_t npl = is_16bit ? 8 : 16, gx = ((max_w + npl - 1) / npl + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (is_16bit) hipLaunchKernelGGL(picture_import_kernel<uint16_t>, dim3(gx, row), dim3(256), 0, s, (const uint16_t*)d_frame, d);
    else hipLaunchKernelGGL(picture_import_kernel<uint8_t>, dim3(gx, row), dim3(256), 0, s, (const uint8_t*)d_frame, d);
    return launch_status("picture_import");
}

extern "C" int svt_hip_picture_pad(void* d_buf, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_w, uint32_t pad_h,
                                   int is_16bit, void* stream) {
    if (int rc = require_init()) return rc;
    if (!d_buf) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0) return set_err(SVT_HIP_ERR_INVALID, "picture %ux%u", width, height);
    if (stride < width + 2 * pad_w) return set_err(SVT_HIP_ERR_INVALID, "stride %u < %u", stride, width + 2 * pad_w);
    const uint32_t rows = height + 2 * pad_h;
    if (rows > 65535) return set_err(SVT_HIP_ERR_INVALID, "%u buffer rows in one launch", rows);
    if (pad_w == 0 && pad_h == 0) return SVT_HIP_OK;
    const uint32_t npl = is_16bit ? 8 : 16, gx = ((width + 2 * pad_w + npl - 1) / npl + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (is_16bit) hipLaunchKernelGGL(picture_pad_kernel<uint16_t>, dim3(gx, rows), dim3(256), 0, s, (uint16_t*)d_buf, stride, (int)width, (int)height, (int)pad_w, (int)pad_h);
    else hipLaunchKernelGGL(picture_pad_kernel<uint8_t>, dim3(gx, rows), dim3(256), 0, s, (uint8_t*)d_buf, stride, (int)width, (int)height, (int)pad_w, (int)pad_h);
    return launch_status("picture_pad");
}

extern "C" int svt_hip_picture_luma8(const uint16_t* d_in, uint32_t in_stride, uint8_t* d_out, uint32_t out_stride, uint32_t cols, uint32_t rows,
                                     int bd, void* stream) {
    if (int rc = require_init()) return rc;
    if (!d_in || !d_out) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (bd < 9 || bd > 16) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bd);
    if (cols == 0 || rows == 0 || rows > 65535) return set_err(SVT_HIP_ERR_INVALID, "buffer %ux%u", cols, rows);
    if (in_stride < cols || out_stride < cols) return set_err(SVT_HIP_ERR_INVALID, "stride below %u columns", cols);
    hipLaunchKernelGGL(picture_luma8_kernel, dim3(((cols + 15) / 16 + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, d_in, in_stride, d_out,
                       out_stride, (int)cols, bd - 8);
    return launch_status("picture_luma8");
}

extern "C" int svt_hip_picture_decimate(const uint8_t* d_luma, uint32_t luma_stride, uint32_t width, uint32_t height,
                                        uint8_t* d_quarter, uint32_t q_stride, uint32_t q_origin_x, uint32_t q_origin_y,
                                        uint8_t* d_sixteenth, uint32_t s_stride, uint32_t s_origin_x, uint32_t s_origin_y, void* stream) {
    if (int rc = require_init()) return rc;
    if (!d_luma || (!d_quarter && !d_sixteenth)) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0) return set_err(SVT_HIP_ERR_INVALID, "picture %ux%u", width, height);
    PicDecim q, x;
    memset(&q, 0, sizeof(q)); memset(&x, 0, sizeof(x));
    uint32_t row = 0, max_w = 0;
    if (d_quarter) {
        q = PicDecim{d_quarter, q_stride, (width + 1) / 2, (height + 1) / 2, q_origin_x, q_origin_y, 0};
        if (q_stride < q.w + 2 * q.ox) return set_err(SVT_HIP_ERR_INVALID, "quarter stride %u < %u", q_stride, q.w + 2 * q.ox);
        row += q.h + 2 * q.oy; max_w = q.w + 2 * q.ox;
    }
    if (d_sixteenth) {
        x = PicDecim{d_sixteenth, s_stride, (width + 3) / 4, (height + 3) / 4, s_origin_x, s_origin_y, row};
        if (s_stride < x.w + 2 * x.ox) return set_err(SVT_HIP_ERR_INVALID, "sixteenth stride %u < %u", s_stride, x.w + 2 * x.ox);
        row += x.h + 2 * x.oy;
        if (x.w + 2 * x.ox > max_w) max_w = x.w + 2 * x.ox;
    }
    if (row > 65535) return set_err(SVT_HIP_ERR_INVALID, "%u buffer rows in one launch", row);
    // without a quarter picture the sixteenth's rows start at 0 and the kernel's "row >= s.row0" picks it for every row
    hipLaunchKernelGGL(picture_decimate_kernel, dim3(((max_w + 15) / 16 + 255) / 256, row), dim3(256), 0, (hipStream_t)stream, d_luma, luma_stride,
                       (int)width, q, x);
    return launch_status("picture_decimate");
}
