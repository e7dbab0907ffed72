// kernel_picture.h — the picture-input side of the path: a frame as the y4m file holds it (planes back to back,
// no padding) laid out in the encoder's picture buffers, the replicated borders those buffers carry, and the 1/4 and 1/16
// pictures the hierarchical motion estimation searches.  Reference: pad_input_picture (EbMcp.c:273-317) +
// generate_padding{,16_bit} (:176-267) as PadPictureToMultipleOfMinCuSizeDimensions / PadPictureToMultipleOfLcuDimensions
// (EbPictureAnalysisProcess.c:4818-4902) apply them, and Decimation2D + generate_padding as DecimateInputPicture does (:4907-4958).
//
// Both reference steps replicate the nearest picture sample, first along rows and then whole rows, so every sample of a
// padded buffer has the closed form  buf[y][x] = pic[clamp(y - oy, 0, H - 1)][clamp(x - ox, 0, W - 1)]  (checked against the
// reference in tests/test_oracle_vs_ref.py); the kernels evaluate that form directly, one 16-byte chunk of output per lane,
// no ordering between lanes.  All three are HBM streams: bytes in + bytes out per sample of the padded buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svtdev {

struct PicPlane {                  // one plane of a padded picture buffer
    void* buf;                     // first sample of the buffer (NOT the picture origin)
    uint32_t stride;               // samples per buffer row
    uint32_t w, h;                 // picture size in samples (what the frame holds: before any right / bottom extension)
    uint32_t full_w, full_h;       // buffer columns / rows to fill: picture + extension + both borders
    uint32_t ox, oy;               // picture origin inside the buffer
    uint32_t src_off;              // plane's first sample in the packed frame (samples)
    uint32_t row0;                 // first row of this plane in the launch's row space
};
struct PicImport { PicPlane p[3]; uint32_t nplanes, rows_total; };

// out chunk = NPL samples of buffer row y starting at column x0, from a source whose rows are `w` samples apart (packed) or
// `src_stride` apart (in place); the interior of a row is a straight (unaligned) 16-byte load
template <typename PixT>
__device__ __forceinline__ void pic_fill_chunk(PixT* __restrict__ out, const PixT* __restrict__ srow, int x0, int ox, int w, int full_w) {
    constexpr int NPL = 16 / (int)sizeof(PixT);
    const int sx0 = x0 - ox;
    if (sx0 >= 0 && sx0 + NPL <= w && x0 + NPL <= full_w) {
        uint4 v;
        __builtin_memcpy(&v, srow + sx0, 16);
        __builtin_memcpy(out, &v, 16);
    } else {
#pragma unroll
        for (int k = 0; k < NPL; k++)
            if (x0 + k < full_w) out[k] = srow[min(max(sx0 + k, 0), w - 1)];
    }
}

// frame (packed planes) -> up to three padded plane buffers, one launch: grid.y = buffer rows of all planes, grid.x = chunks
template <typename PixT>
__global__ __launch_bounds__(256) void picture_import_kernel(const PixT* __restrict__ frame, const PicImport d) {
    constexpr int NPL = 16 / (int)sizeof(PixT);
    const uint32_t row = blockIdx.y;
    const int pi = (d.nplanes > 2 && row >= d.p[2].row0) ? 2 : ((d.nplanes > 1 && row >= d.p[1].row0) ? 1 : 0);
    const PicPlane& p = d.p[pi];
    const int y = (int)(row - p.row0);
    const int x0 = (int)(blockIdx.x * 256u + threadIdx.x) * NPL;
    if (x0 >= (int)p.full_w) return;
    const int sy = min(max(y - (int)p.oy, 0), (int)p.h - 1);
    pic_fill_chunk<PixT>(reinterpret_cast<PixT*>(p.buf) + (size_t)y * p.stride + x0, frame + p.src_off + (size_t)sy * p.w, x0, (int)p.ox,
                         (int)p.w, (int)p.full_w);
}

// generate_padding in place: lanes whose chunk lies wholly inside the picture return without touching memory; the others read
// picture samples only and write border samples only, so no lane reads what another writes
template <typename PixT>
__global__ __launch_bounds__(256) void picture_pad_kernel(PixT* __restrict__ buf, uint32_t stride, int w, int h, int pad_w, int pad_h) {
    constexpr int NPL = 16 / (int)sizeof(PixT);
    const int y = (int)blockIdx.y, full_w = w + 2 * pad_w;
    const int x0 = (int)(blockIdx.x * 256u + threadIdx.x) * NPL;
    if (x0 >= full_w) return;
    const bool row_inside = y >= pad_h && y < pad_h + h;
    if (row_inside && x0 >= pad_w && x0 + NPL <= pad_w + w) return;
    const int sy = min(max(y - pad_h, 0), h - 1);
    const PixT* srow = buf + (size_t)(pad_h + sy) * stride + pad_w;
    PixT* out = buf + (size_t)y * stride + x0;
#pragma unroll
    for (int k = 0; k < NPL; k++) {
        const int x = x0 + k, sx = x - pad_w;
        if (x < full_w && !(row_inside && sx >= 0 && sx < w)) out[k] = srow[min(max(sx, 0), w - 1)];
    }
}

// Decimation2D at step 2 and / or 4 + the decimated picture's own borders: out[y][x] = luma[STEP * cy][STEP * cx] with (cx, cy)
// the clamped decimated coordinates.  16 output bytes per lane; the interior gathers every STEP-th byte of 16 * STEP loaded ones.
struct PicDecim { uint8_t* buf; uint32_t stride, w, h, ox, oy, row0; };      // w, h: decimated picture size = ceil(luma / step)
template <int STEP>
__device__ __forceinline__ void pic_decim_chunk(uint8_t* __restrict__ out, const uint8_t* __restrict__ srow, int x0, int ox, int w, int full_w,
                                                int luma_w) {
    const int cx0 = x0 - ox;
    if (cx0 >= 0 && cx0 + 16 <= w && x0 + 16 <= full_w && (cx0 + 16) * STEP <= luma_w) {
        uint32_t in[4 * STEP], o[4];
        __builtin_memcpy(in, srow + (size_t)cx0 * STEP, 16 * STEP);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (STEP == 2) o[q] = __builtin_amdgcn_perm(in[2 * q + 1], in[2 * q], 0x06040200u);       // bytes 0, 2 of each dword
            else o[q] = (in[4 * q] & 0xffu) | ((in[4 * q + 1] & 0xffu) << 8) | ((in[4 * q + 2] & 0xffu) << 16) | (in[4 * q + 3] << 24);
        }
        __builtin_memcpy(out, o, 16);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (x0 + k < full_w) out[k] = srow[(size_t)min(max(cx0 + k, 0), w - 1) * STEP];
    }
}
__global__ __launch_bounds__(256) void picture_decimate_kernel(const uint8_t* __restrict__ luma, uint32_t luma_stride, int luma_w, const PicDecim q,
                                                               const PicDecim s) {
    const uint32_t row = blockIdx.y;
    const bool six = s.buf != nullptr && row >= s.row0;
    const PicDecim& d = six ? s : q;
    const int y = (int)(row - d.row0), full_w = (int)(d.w + 2 * d.ox);
    const int x0 = (int)(blockIdx.x * 256u + threadIdx.x) * 16;
    if (x0 >= full_w) return;
    const int cy = min(max(y - (int)d.oy, 0), (int)d.h - 1);
    uint8_t* out = d.buf + (size_t)y * d.stride + x0;
    if (six) pic_decim_chunk<4>(out, luma + (size_t)cy * 4 * luma_stride, x0, (int)d.ox, (int)d.w, full_w, luma_w);
    else pic_decim_chunk<2>(out, luma + (size_t)cy * 2 * luma_stride, x0, (int)d.ox, (int)d.w, full_w, luma_w);
}

// The 8-bit picture the analysis stages (HME, ME, open-loop intra search) read when the input is deeper: the reference keeps a
// 10-bit picture as an 8-bit plane (the samples' top 8 bits) + a 2-bit plane and searches on the former (EbPictureBufferDesc:
// buffer_y / bufferBitIncY; unpack in EbPackUnPack_C.c).  Here the picture is 16-bit samples, so the 8-bit plane is v >> (bd - 8),
// over the whole padded buffer (borders included: replication commutes with the shift).  16 samples per lane.
__global__ __launch_bounds__(256) void picture_luma8_kernel(const uint16_t* __restrict__ in, uint32_t in_stride, uint8_t* __restrict__ out,
                                                            uint32_t out_stride, int cols, int shift) {
    const int y = (int)blockIdx.y, x0 = (int)(blockIdx.x * 256u + threadIdx.x) * 16;
    if (x0 >= cols) return;
    const uint16_t* src = in + (size_t)y * in_stride + x0;
    uint8_t* dst = out + (size_t)y * out_stride + x0;
    if (x0 + 16 <= cols) {
        uint32_t v[8], o[4];
        __builtin_memcpy(v, src, 32);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t a = v[2 * q], b = v[2 * q + 1];
            o[q] = (((a & 0xffffu) >> shift) & 0xffu) | ((((a >> 16) >> shift) & 0xffu) << 8) | ((((b & 0xffffu) >> shift) & 0xffu) << 16) | (((b >> 16) >> shift) << 24);
        }
        __builtin_memcpy(dst, o, 16);
    } else {
        for (int k = 0; x0 + k < cols; k++) dst[k] = (uint8_t)(src[k] >> shift);
    }
}

}  // namespace svtdev
