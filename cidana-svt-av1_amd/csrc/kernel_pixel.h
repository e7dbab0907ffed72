// kernel_pixel.h — batched quantize_b, SAD, SAD search, SSE, residual kernels.
#pragma once
#include <type_traits>
#include "dev_common.h"

namespace svtdev {

// ---------------------------------------------------------------------------
// plain NxM SAD / SSE of block pairs (fast_loop_nx_m_sad_kernel,
// C_DEFAULT/EbComputeSAD_C.c:48; spatial_full_distortion_kernel,
// C_DEFAULT/EbPictureOperators_C.c:40).  Blocks dense or at offsets; one
// LPB-lane group per block, lane = (row, 4-pixel group) walking the block.
// ---------------------------------------------------------------------------
// MODE 0: SAD (fast_loop_nx_m_sad_kernel / aom_sadMxN_c), 1: SSE (spatial_full_distortion_kernel), 2: SAD against the
// rounded average of two references (combined_averaging_sad, C_DEFAULT/EbComputeSAD_C.c:13-40: avg = (r1 + r2 + 1) >> 1).
// Addressing: block i of `a` is at a + (a_offs ? a_offs[i >> a_shift] : (i >> a_shift) * a_block_pitch) - a_shift = 2 shares
// one source block between the four references of aom_sadMxNx4d - and block i of `b` at b + (b_offs ? b_offs[i] : i * pitch).
template <int MODE>
__global__ __launch_bounds__(256) void sad_sse_kernel(
    const uint8_t* __restrict__ a, uint32_t a_stride, size_t a_block_pitch, const uint32_t* __restrict__ a_offs, int a_shift,
    const uint8_t* __restrict__ b, uint32_t b_stride, size_t b_block_pitch, const uint32_t* __restrict__ b_offs,
    const uint8_t* __restrict__ c2, uint32_t c_stride, size_t c_block_pitch,
    uint32_t width, uint32_t height, void* __restrict__ out, uint32_t nblocks) {
    // 16 lanes per block, 4 blocks per wave; a lane walks (row, chunk) items with chunks of
    // cs = 16 / 8 / 4 / 1 bytes fetched by one unaligned load each.  SAD: v_sad_u8 per dword;
    // SSE: sum (x-y)^2 = x.x + y.y - 2 x.y with three v_dot4_u32_u8 per dword (exact in u32).
    constexpr bool SSE = MODE == 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, l = lane & 15;
    const uint32_t blk = (blockIdx.x * 4 + wave) * 4 + sub;
    const bool valid = blk < nblocks;
    unsigned long long acc = 0;
    if (valid) {
        const uint32_t ai = blk >> a_shift;
        const uint8_t* pa = a + (a_offs ? (size_t)a_offs[ai] : (size_t)ai * a_block_pitch);
        const uint8_t* pb = b + (b_offs ? (size_t)b_offs[blk] : (size_t)blk * b_block_pitch);
        const uint8_t* pc = MODE == 2 ? c2 + (size_t)blk * c_block_pitch : nullptr;
        const uint32_t cs = (width & 15) == 0 ? 16u : ((width & 7) == 0 ? 8u : ((width & 3) == 0 ? 4u : 1u));
        const uint32_t cpr = width / cs, items = cpr * height;
        uint32_t y = l / cpr, c = l - y * cpr;                 // advance (y, c) by 16 items without dividing again
        const uint32_t dy16 = 16 / cpr, dc16 = 16 - dy16 * cpr;
        for (uint32_t i = l; i < items; i += 16) {
            uint32_t va[4] = {0, 0, 0, 0}, vb[4] = {0, 0, 0, 0}, vc[4] = {0, 0, 0, 0};
            const uint8_t* qa = pa + (size_t)y * a_stride + c * cs;
            const uint8_t* qb = pb + (size_t)y * b_stride + c * cs;
            if (cs == 16) { __builtin_memcpy(va, qa, 16); __builtin_memcpy(vb, qb, 16); }
            else if (cs == 8) { __builtin_memcpy(va, qa, 8); __builtin_memcpy(vb, qb, 8); }
            else if (cs == 4) { __builtin_memcpy(va, qa, 4); __builtin_memcpy(vb, qb, 4); }
            else { va[0] = qa[0]; vb[0] = qb[0]; }
            if (MODE == 2) {
                const uint8_t* qc = pc + (size_t)y * c_stride + c * cs;
                if (cs == 16) __builtin_memcpy(vc, qc, 16);
                else if (cs == 8) __builtin_memcpy(vc, qc, 8);
                else if (cs == 4) __builtin_memcpy(vc, qc, 4);
                else vc[0] = qc[0];
#pragma unroll
                for (int k = 0; k < 4; k++)      // per-byte (x + y + 1) >> 1 = (x | y) - ((x ^ y) >> 1)
                    vb[k] = (vb[k] | vc[k]) - (((vb[k] ^ vc[k]) >> 1) & 0x7f7f7f7fu);
            }
            unsigned t = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (SSE) {
                    const unsigned aa = __builtin_amdgcn_udot4(va[k], va[k], 0u, false);
                    const unsigned bb = __builtin_amdgcn_udot4(vb[k], vb[k], 0u, false);
                    const unsigned ab = __builtin_amdgcn_udot4(va[k], vb[k], 0u, false);
                    t += aa + bb - 2u * ab;
                } else {
                    t = __builtin_amdgcn_sad_u8(va[k], vb[k], t);
                }
            }
            acc += t;
            y += dy16; c += dc16;
            if (c >= cpr) { c -= cpr; y++; }
        }
    }
    acc = group_sum64<16>(acc);
    if (valid && l == 0) {
        if (SSE) reinterpret_cast<unsigned long long*>(out)[blk] = acc;
        else reinterpret_cast<uint32_t*>(out)[blk] = (uint32_t)acc;
    }
}

// residual_kernel_c (EbPictureOperators.c:166): int16 res = src - pred.  One item = CS pixels of one
// row (one wide load per input, 2*CS bytes of output), ONE item per lane and a grid as large as the
// job (tools/probe/store_probe.hip: grid-stride loops cost 15-40 % of the store bandwidth).
// POW2: items per row and per block are powers of two (every AV1 block size), so the item -> (block,
// row, chunk) split is shifts and masks; otherwise 64-bit divisions.
// PixT uint16_t: residual_kernel16bit (EbPictureOperators.c:134-164), CS counts samples.
template <int CS, bool POW2, typename PixT = uint8_t>
__global__ __launch_bounds__(256) void residual_kernel(
    const PixT* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const PixT* __restrict__ pred, uint32_t pred_stride, size_t pred_block_pitch,
    int16_t* __restrict__ res, uint32_t res_stride, size_t res_block_pitch, uint32_t width,
    uint32_t height, uint32_t nblocks) {
    const uint32_t cpr = width / CS;
    const size_t per = (size_t)cpr * height, total = per * nblocks;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    size_t blk; uint32_t y, c;
    if (POW2) {
        const int cs = __builtin_ctz(cpr), ps = __builtin_ctzll((unsigned long long)per);
        blk = i >> ps;
        const uint32_t j = (uint32_t)(i & (per - 1));
        y = j >> cs; c = j & (cpr - 1);
    } else {
        blk = i / per;
        const uint32_t j = (uint32_t)(i - blk * per);
        y = j / cpr; c = j - y * cpr;
    }
    const PixT* ps_ = src + blk * src_block_pitch + (size_t)y * src_stride + c * CS;
    const PixT* pp = pred + blk * pred_block_pitch + (size_t)y * pred_stride + c * CS;
    int16_t* pr = res + blk * res_block_pitch + (size_t)y * res_stride + c * CS;
    PixT vs[CS], vp[CS];
    __builtin_memcpy(vs, ps_, CS * sizeof(PixT)); __builtin_memcpy(vp, pp, CS * sizeof(PixT));
    int16_t o[CS];
#pragma unroll
    for (int k = 0; k < CS; k++) o[k] = (int16_t)((int)vs[k] - (int)vp[k]);
    __builtin_memcpy(pr, o, 2 * CS);
}

// ---------------------------------------------------------------------------
// SAD search (sad_loop_kernel, C_DEFAULT/EbComputeSAD_C.c:72-120): for every
// candidate (x, y) of a search_area_width x search_area_height window, SAD of
// the W x H source block against ref + x + y*ref_stride_raw; first strict
// minimum in raster order wins.  One wave per block: the source block and the
// reference window are staged in LDS; lane t evaluates candidates t, t+64, ...
// v_sad_u8 on dwords rebuilt from aligned LDS words with v_alignbyte.
// The argmin key (sad << 32 | candidate index) reproduces the tie-break.
// ---------------------------------------------------------------------------
// LDS per wave = src_lds_bytes + ref_lds_bytes (host-computed, 16-B multiples);
// blockDim.x / 64 waves per workgroup, dynamic LDS sized accordingly.
__global__ __launch_bounds__(256) void sad_search_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, uint32_t ref_stride_raw, size_t ref_block_pitch,
    uint32_t width, uint32_t height, int search_w, int search_h,
    unsigned long long* __restrict__ best_sad, int16_t* __restrict__ best_x, int16_t* __restrict__ best_y,
    uint32_t src_lds_bytes, uint32_t ref_lds_bytes, const uint32_t* __restrict__ src_offs,
    const uint32_t* __restrict__ ref_offs, uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blk = blockIdx.x * (blockDim.x >> 6) + wave;
    // window geometry in LDS: rows of `wpitch` bytes (multiple of 4, + 4 spare)
    const uint32_t win_w = width + search_w - 1;
    // rows of the window that can be touched: candidate row ys adds ys*ref_stride_raw,
    // block row y adds y*ref_stride; stage (row, offset) pairs as needed.
    const uint32_t wpitch = (win_w + 3 + 8) & ~3u;   // host uses the same formula
    const uint32_t spitch = (width + 3) & ~3u;
    uint8_t* s_src = smem + (size_t)wave * (src_lds_bytes + ref_lds_bytes);
    uint8_t* s_ref = s_src + src_lds_bytes;
    if (blk >= nblocks) return;   // whole wave exits together (blk is wave-uniform)
    const uint8_t* gs = src + (src_offs ? (size_t)src_offs[blk] : (size_t)blk * src_block_pitch);
    const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[blk] : (size_t)blk * ref_block_pitch);
    // stage the source block
    for (uint32_t i = lane; i < spitch * height; i += 64) {
        const uint32_t y = i / spitch, x = i - y * spitch;
        s_src[i] = x < width ? gs[(size_t)y * src_stride + x] : 0;
    }
    // stage the reference rows.  Row index space: rr = ys * height + y  ->
    // address ys*ref_stride_raw + y*ref_stride (covers line-skipping callers where
    // ref_stride = 2*ref_stride_raw).  When ref_stride == ref_stride_raw the rows
    // overlap and only search_h + height - 1 distinct rows are staged.
    const bool plain = (ref_stride == ref_stride_raw);
    const uint32_t nrows = plain ? (uint32_t)(search_h + height - 1) : (uint32_t)search_h * height;
    for (uint32_t i = lane; i < wpitch * nrows; i += 64) {
        const uint32_t rr = i / wpitch, x = i - rr * wpitch;
        size_t off;
        if (plain) off = (size_t)rr * ref_stride_raw;
        else off = (size_t)(rr / height) * ref_stride_raw + (size_t)(rr % height) * ref_stride;
        s_ref[i] = x < win_w ? gr[off + x] : 0;
    }
    wave_lds_fence();
    unsigned long long best = ~0ull;
    const int ncand = search_w * search_h;
    const uint32_t wq = spitch >> 2;
    for (int cand = lane; cand < ncand; cand += 64) {
        const int ys = cand / search_w, xs = cand - ys * search_w;
        unsigned acc = 0;
        for (uint32_t y = 0; y < height; y++) {
            const uint32_t rr = plain ? (uint32_t)ys + y : (uint32_t)ys * height + y;
            const uint32_t* rrow = reinterpret_cast<const uint32_t*>(s_ref + rr * wpitch) + (xs >> 2);
            const uint32_t* srow = reinterpret_cast<const uint32_t*>(s_src + y * spitch);
            const unsigned sh = (unsigned)(xs & 3);
            uint32_t lo = rrow[0];
            for (uint32_t q = 0; q < wq; q++) {
                const uint32_t hi = rrow[q + 1];
                const uint32_t rv = __builtin_amdgcn_alignbyte(hi, lo, sh);
                uint32_t sv = srow[q];
                uint32_t rvm = rv;
                const uint32_t rem = width - q * 4;      // < 4 only on a ragged last group
                if (rem < 4) { const uint32_t m = (1u << (rem * 8)) - 1; rvm &= m; sv &= m; }
                acc = __builtin_amdgcn_sad_u8(sv, rvm, acc);
                lo = hi;
            }
        }
        const unsigned long long key = ((unsigned long long)acc << 32) | (unsigned)cand;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(best, m, 64);
        best = o < best ? o : best;
    }
    if (lane == 0) {
        const unsigned sadv = (unsigned)(best >> 32);
        const int cand = (int)(best & 0xffffffffu);
        // reference initialises best_sad = 0xffffff and only updates on strict '<'
        if (ncand > 0 && sadv < 0xffffffu) {
            best_sad[blk] = sadv;
            best_x[blk] = (int16_t)(cand % search_w);
            best_y[blk] = (int16_t)(cand / search_w);
        } else {
            best_sad[blk] = 0xffffffu;   // x/y untouched, as in the reference
        }
    }
}

// ---------------------------------------------------------------------------
// SAD search, quad-SAD form (block width multiple of 4 — every AV1 width): a lane owns
// FOUR horizontally adjacent candidates (xs0 .. xs0+3, xs0 multiple of 4) of one search
// row, so every reference dword pair it needs is ALIGNED in the staged window and one
// v_qsad_pk_u16_u8 yields 4 candidates x 4 pixels.  lpb (power of two) lanes share a
// block, 64/lpb blocks per wave.  The packed u16 accumulators are flushed into u32
// before they can overflow (every floor(257/W) rows).  Tie-break as above.
// ---------------------------------------------------------------------------
template <int CW, int CH>
__global__ __launch_bounds__(256) void sad_search_q_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, uint32_t ref_stride_raw, size_t ref_block_pitch,
    uint32_t width_rt, uint32_t height_rt, int search_w, int search_h,
    unsigned long long* __restrict__ best_sad, int16_t* __restrict__ best_x, int16_t* __restrict__ best_y,
    uint32_t src_lds_bytes, uint32_t ref_lds_bytes, uint32_t lpb, const uint32_t* __restrict__ src_offs,
    const uint32_t* __restrict__ ref_offs, uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t width = CW ? (uint32_t)CW : width_rt, height = CH ? (uint32_t)CH : height_rt;   // compile-time when specialised
    const uint32_t tid = threadIdx.x;
    const uint32_t slot = tid / lpb, l = tid % lpb;               // block slot in the workgroup, lane in the block
    const uint32_t slots = blockDim.x / lpb;
    const uint32_t blk = blockIdx.x * slots + slot;
    const bool valid = blk < nblocks;
    const uint32_t win_w = width + search_w - 1;
    const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;      // whole 16-B chunks + slack for the sliding window
    const bool plain = (ref_stride == ref_stride_raw);
    const uint32_t nrows = plain ? (uint32_t)(search_h + height - 1) : (uint32_t)search_h * height;
    uint8_t* s_src = smem + (size_t)slot * (src_lds_bytes + ref_lds_bytes);
    uint8_t* s_ref = s_src + src_lds_bytes;
    const uint32_t wq = width >> 2;
    if (valid) {
        const uint8_t* gs = src + (src_offs ? (size_t)src_offs[blk] : (size_t)blk * src_block_pitch);
        const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[blk] : (size_t)blk * ref_block_pitch);
        // Staging: 2-D lane arrangement (power-of-two row length: no divisions), up to 4 wide
        // unaligned loads in flight per lane (a one-load-per-iteration loop is latency-bound).
        {   // source block: rows of `width` bytes in chunks of cs = 16 / 8 / 4 bytes
            const uint32_t cs = (width & 15) == 0 ? 16u : ((width & 7) == 0 ? 8u : 4u);
            const uint32_t cpr = width / cs;
            uint32_t lxs = 1;
            while (lxs < cpr && lxs < lpb) lxs <<= 1;
            const uint32_t lx = l & (lxs - 1), ly = l / lxs, lys = lpb / lxs;
            // (the chunk size is a compile-time constant inside each branch: with `cs` as a run-time memcpy length the generic
            // <0, 0> instantiation kept v[] in SCRATCH, 80 bytes per lane)
            auto stage_src = [&](auto CSC) {
                constexpr uint32_t CS = decltype(CSC)::value;
                for (uint32_t y0 = ly; y0 < height; y0 += 4 * lys)
                    for (uint32_t c = lx; c < cpr; c += lxs) {
                        uint4 v[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t y = y0 + k * lys;
                            v[k] = make_uint4(0, 0, 0, 0);
                            if (y < height) __builtin_memcpy(&v[k], gs + (size_t)y * src_stride + c * CS, CS);
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t y = y0 + k * lys;
                            if (y < height) __builtin_memcpy(s_src + y * width + c * CS, &v[k], CS);
                        }
                    }
            };
            if (cs == 16) stage_src(std::integral_constant<uint32_t, 16>{});
            else if (cs == 8) stage_src(std::integral_constant<uint32_t, 8>{});
            else stage_src(std::integral_constant<uint32_t, 4>{});
        }
        {   // reference window: rows of wpitch bytes in 16-B chunks; a chunk is loaded wide when it
            // ends inside the window's own footprint (row tails over-read the next row: harmless, only
            // excluded candidates can touch those bytes), else byte-wise with zero fill
            const uint32_t cpr = (win_w + 15) >> 4;
            const size_t span = plain ? (size_t)(nrows - 1) * ref_stride_raw + win_w
                                      : (size_t)(search_h - 1) * ref_stride_raw + (size_t)(height - 1) * ref_stride + win_w;
            uint32_t rxs = 1;
            while (rxs < cpr && rxs < lpb) rxs <<= 1;
            const uint32_t lx = l & (rxs - 1), ly = l / rxs, lys = lpb / rxs;
            for (uint32_t r0 = ly; r0 < nrows; r0 += 4 * lys)
                for (uint32_t c = lx; c < cpr; c += rxs) {
                    uint4 v[4];
                    uint32_t back[4];       // a chunk that would end past the footprint is fetched `back` bytes earlier
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t rr = r0 + k * lys;
                        v[k] = make_uint4(0, 0, 0, 0);
                        back[k] = 0;
                        if (rr < nrows) {
                            const size_t off = (plain ? (size_t)rr * ref_stride_raw
                                                      : (size_t)(rr / height) * ref_stride_raw + (size_t)(rr % height) * ref_stride) + c * 16;
                            if (off + 16 <= span) {
                                __builtin_memcpy(&v[k], gr + off, 16);
                            } else if (span >= 16 && off < span && (plain || off - (span - 16) <= c * 16)) {
                                // the last 16 bytes of the footprint, stored shifted: the bytes re-written before
                                // the chunk are the same data (or row slack); byte-wise global loads here cost one
                                // memory latency EACH and ran for every block of a dense window array
                                back[k] = (uint32_t)(off - (span - 16));
                                __builtin_memcpy(&v[k], gr + (span - 16), 16);
                            } else {
                                uint8_t* vb = reinterpret_cast<uint8_t*>(&v[k]);
                                for (uint32_t b = 0; b < 16; b++) if (off + b < span) vb[b] = gr[off + b];
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t rr = r0 + k * lys;
                        if (rr < nrows) {
                            if (back[k] == 0) *reinterpret_cast<uint4*>(s_ref + rr * wpitch + c * 16) = v[k];
                            else {
                                const uint8_t* vb = reinterpret_cast<const uint8_t*>(&v[k]);
                                uint8_t* d = s_ref + rr * wpitch + c * 16 - back[k];
#pragma unroll
                                for (int b = 0; b < 16; b++) d[b] = vb[b];
                            }
                        }
                    }
                }
        }
    }
    __syncthreads();            // staging -> search (a block's lanes share a wave, lpb <= 64; the barrier also orders the tail byte stores)
    unsigned long long best = ~0ull;
    if (valid) {
        const int gx = (search_w + 3) >> 2;
        const int ngroups = gx * search_h;
        const uint32_t flush = 257u / width > 0 ? 257u / width : 1u;
        constexpr int ROW_UNROLL = CH ? (CH > 16 ? 4 : CH) : 1, COL_UNROLL = CW ? CW / 4 : 1;
        for (int g = (int)l; g < ngroups; g += (int)lpb) {
            const int ys = g / gx, xg = g - ys * gx;
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            unsigned long long acc = 0;
            uint32_t since = 0;
#pragma unroll ROW_UNROLL
            for (uint32_t y = 0; y < height; y++) {
                const uint32_t rr = plain ? (uint32_t)ys + y : (uint32_t)ys * height + y;
                const uint32_t* rrow = reinterpret_cast<const uint32_t*>(s_ref + rr * wpitch) + xg;
                const uint32_t* srow = reinterpret_cast<const uint32_t*>(s_src) + y * wq;
                uint32_t lo = rrow[0];
#pragma unroll COL_UNROLL
                for (uint32_t q = 0; q < wq; q++) {
                    const uint32_t hi = rrow[q + 1];
                    acc = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)hi << 32) | lo, srow[q], acc);
                    lo = hi;
                }
                if (++since == flush || y + 1 == height) {
                    a0 += (uint32_t)(acc & 0xffffu); a1 += (uint32_t)((acc >> 16) & 0xffffu);
                    a2 += (uint32_t)((acc >> 32) & 0xffffu); a3 += (uint32_t)(acc >> 48);
                    acc = 0; since = 0;
                }
            }
            const int xs0 = xg * 4, cbase = ys * search_w + xs0;
            const uint32_t sads[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (xs0 + j < search_w) {
                    const unsigned long long key = ((unsigned long long)sads[j] << 32) | (unsigned)(cbase + j);
                    best = key < best ? key : best;
                }
        }
    }
    for (uint32_t m = lpb >> 1; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(best, (int)m, 64);
        best = o < best ? o : best;
    }
    // results of the workgroup's `slots` consecutive blocks leave through LDS so that the three
    // output arrays get one contiguous store each (scattered 2-/8-byte stores cost ~350 B of HBM
    // write traffic apiece: measured WRITE_SIZE 1.07 GB for 12 MB of results)
    __syncthreads();
    unsigned long long* s_out = reinterpret_cast<unsigned long long*>(smem);     // staging is dead
    if (l == 0) s_out[slot] = valid ? best : ~0ull;
    __syncthreads();
    if (tid < slots) {
        const uint32_t ob = blockIdx.x * slots + tid;
        if (ob < nblocks) {
            const unsigned long long key = s_out[tid];
            const unsigned sadv = (unsigned)(key >> 32);
            const int cand = (int)(key & 0xffffffffu);
            // reference initialises best_sad = 0xffffff and only updates on strict '<'
            if (sadv < 0xffffffu) {
                best_sad[ob] = sadv;
                best_x[ob] = (int16_t)(cand % search_w);
                best_y[ob] = (int16_t)(cand / search_w);
            } else {
                best_sad[ob] = 0xffffffu;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// SAD search for small blocks (CW*CH <= 256 pixels: the whole source block fits in 64 VGPRs), plain
// reference window.  A lane owns FOUR horizontally adjacent candidates of TWO vertically adjacent
// search rows: every reference row it reads from LDS (CW/4+1 dwords) feeds 2*CW/4 v_qsad_pk_u16_u8
// against two source rows held in registers, instead of CW/4 qsads per CW/4+1 reference and CW/4
// source dwords in sad_search_q_kernel - the LDS pipe, not v_qsad, was that kernel's limiter at
// 16x16 (profiles/r01_valu_issue_cost_2.txt: v_qsad costs 6 plain instructions, a ds_read_b32 two
// LDS cycles).  A candidate's whole SAD fits its packed u16 accumulator (256 * 255 < 2^16).
// Argmin key and outputs as in sad_search_q_kernel.
// ---------------------------------------------------------------------------
template <int CW, int CH, int Q2_SU>
__global__ __launch_bounds__(256) void sad_search_q2_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, size_t ref_block_pitch, int search_w, int search_h,
    unsigned long long* __restrict__ best_sad, int16_t* __restrict__ best_x, int16_t* __restrict__ best_y,
    uint32_t ref_lds_bytes, uint32_t lpb, uint32_t cpr_magic, const uint32_t* __restrict__ src_offs,
    const uint32_t* __restrict__ ref_offs, uint32_t nblocks) {
    static_assert(CW % 4 == 0 && CW * CH <= 256, "source block must fit 64 VGPRs");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int WQ = CW / 4;
    constexpr uint32_t SRC_BYTES = (CW * CH + 15) & ~15;
    const uint32_t tid = threadIdx.x;
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slot = tid >> lsh, l = tid & (lpb - 1);
    const uint32_t slots = blockDim.x >> lsh;
    const uint32_t blk = blockIdx.x * slots + slot;
    const bool valid = blk < nblocks;
    const uint32_t win_w = CW + search_w - 1;
    const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;
    const uint32_t nrows = (uint32_t)(search_h + CH - 1);
    // LDS: all source blocks first (16-B aligned), then the windows at a pitch of ref_lds_bytes == 8 (mod 32)
    // bytes: with the 48-B row pitch of a 23-wide window the 32 lanes of a ds_read_b32 group (4 blocks x
    // 4 row pairs x 2 column groups) then hit 32 different banks (measured before: 56 % of the LDS
    // cycles were bank conflicts).  Window chunks are therefore written as two 8-B halves.
    uint8_t* s_src = smem + (size_t)slot * SRC_BYTES;
    uint8_t* s_ref = smem + (size_t)slots * SRC_BYTES + (size_t)slot * ref_lds_bytes;
    if (valid) {
        const uint8_t* gs = src + (src_offs ? (size_t)src_offs[blk] : (size_t)blk * src_block_pitch);
        const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[blk] : (size_t)blk * ref_block_pitch);
        // ---- stage the source block and the window.  One flat list of 16-B chunks (source rows first,
        // then window rows of cpr chunks); a lane issues up to SU loads back to back and only then
        // writes LDS, so the block costs ONE memory latency (three dependent load -> LDS phases made a
        // wave live 9 us for 1 us of arithmetic).  Chunks are loaded whole when they end inside the
        // window's own footprint (row tails over-read into the next row: harmless, only excluded
        // candidates can touch those bytes); the few that do not are fetched byte-wise afterwards.
        constexpr uint32_t CS = CW % 16 == 0 ? 16 : (CW % 8 == 0 ? 8 : 4);
        constexpr uint32_t NSRC = CW * CH / CS;
        constexpr int SU = Q2_SU;
        const uint32_t cpr = (win_w + 15) >> 4;
        const uint32_t nref = nrows * cpr;
        const size_t span = (size_t)(nrows - 1) * ref_stride + win_w;
        // every load below is unconditional (clamped index / clamped offset), so nothing separates them
        for (uint32_t i0 = l; i0 < NSRC || i0 < nref; i0 += SU * lpb) {
            uint4 vs[SU], vr[SU];
            uint32_t rdst[SU], tdst[SU];
#pragma unroll
            for (int k = 0; k < SU; k++) {
                const uint32_t i = min(i0 + k * lpb, NSRC - 1);
                vs[k] = make_uint4(0, 0, 0, 0);
                __builtin_memcpy(&vs[k], gs + (size_t)(i / (CW / CS)) * src_stride + (i % (CW / CS)) * CS, CS);
            }
#pragma unroll
            for (int k = 0; k < SU; k++) {
                const uint32_t j = min(i0 + k * lpb, nref - 1);
                const uint32_t rr = cpr == 1 ? j : __umulhi(j, cpr_magic), c = j - rr * cpr;     // j / cpr, j % cpr
                const size_t off = (size_t)rr * ref_stride + c * 16;
                // a chunk that would end past the window's footprint is fetched as the LAST 16 bytes of the
                // footprint instead and stored `delta` bytes earlier (the bytes it re-writes are the same data)
                const size_t offc = off + 16 <= span ? off : span - 16;
                __builtin_memcpy(&vr[k], gr + offc, 16);
                const bool want = i0 + k * lpb < nref;
                rdst[k] = (want && off == offc) ? rr * wpitch + c * 16 : ~0u;
                tdst[k] = (want && off != offc) ? rr * wpitch + c * 16 - (uint32_t)(off - offc) : ~0u;
            }
            // All loads are consumed here, so they are issued above and waited for once; without this the
            // predicated LDS writes below let LLVM sink each load into its branch (one latency per chunk).
            static_assert(SU == 4 || SU == 8, "operand lists below");
            if constexpr (SU == 4)
                asm volatile("" ::"v"(vs[0].x), "v"(vs[1].x), "v"(vs[2].x), "v"(vs[3].x), "v"(vr[0].x), "v"(vr[1].x), "v"(vr[2].x), "v"(vr[3].x));
            else
                asm volatile("" ::"v"(vs[0].x), "v"(vs[1].x), "v"(vs[2].x), "v"(vs[3].x), "v"(vr[0].x), "v"(vr[1].x), "v"(vr[2].x), "v"(vr[3].x),
                             "v"(vs[4 % SU].x), "v"(vs[5 % SU].x), "v"(vs[6 % SU].x), "v"(vs[7 % SU].x), "v"(vr[4 % SU].x), "v"(vr[5 % SU].x), "v"(vr[6 % SU].x), "v"(vr[7 % SU].x));
#pragma unroll
            for (int k = 0; k < SU; k++) {
                const uint32_t i = i0 + k * lpb;
                if (i < NSRC) __builtin_memcpy(s_src + i * CS, &vs[k], CS);
                if (rdst[k] != ~0u) {
                    uint2* rd = reinterpret_cast<uint2*>(s_ref + rdst[k]);
                    rd[0] = make_uint2(vr[k].x, vr[k].y); rd[1] = make_uint2(vr[k].z, vr[k].w);
                }
            }
#pragma unroll
            for (int k = 0; k < SU; k++)
                if (tdst[k] != ~0u) {                      // footprint tail: byte-granular LDS store, no global access
                    const uint8_t* vb = reinterpret_cast<const uint8_t*>(&vr[k]);
                    for (int bb = 0; bb < 16; bb++) s_ref[tdst[k] + bb] = vb[bb];
                }
        }
    }
    __syncthreads();
    unsigned long long best = ~0ull;
    if (valid) {
        // the whole source block into registers (broadcast reads: the lanes of a block share addresses)
        uint32_t sreg[CH][WQ];
#pragma unroll
        for (int y = 0; y < CH; y++) {
            if constexpr (WQ % 4 == 0) {
#pragma unroll
                for (int i = 0; i < WQ / 4; i++) {
                    const uint4 a = *reinterpret_cast<const uint4*>(s_src + y * CW + 16 * i);
                    sreg[y][4 * i] = a.x; sreg[y][4 * i + 1] = a.y; sreg[y][4 * i + 2] = a.z; sreg[y][4 * i + 3] = a.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < WQ; q++) sreg[y][q] = *reinterpret_cast<const uint32_t*>(s_src + y * CW + 4 * q);
            }
        }
        const int gx = (search_w + 3) >> 2, gy = (search_h + 1) >> 1;
        const int ngroups = gx * gy;
        for (int g = (int)l; g < ngroups; g += (int)lpb) {
            const int yp = g / gx, xg = g - yp * gx;
            const int ysA = 2 * yp;
            const bool hasB = ysA + 1 < search_h;
            unsigned long long accA = 0, accB = 0;
            const uint8_t* rbase = s_ref + (size_t)ysA * wpitch + xg * 4;
#pragma unroll
            for (int r = 0; r <= CH; r++) {
                // row ysA + CH is past the window when the pair has no second row: re-read the last one
                const uint32_t* rrow = reinterpret_cast<const uint32_t*>(rbase + (size_t)((r == CH && !hasB) ? CH - 1 : r) * wpitch);
                uint32_t rw[WQ + 1];
#pragma unroll
                for (int q = 0; q <= WQ; q++) rw[q] = rrow[q];
#pragma unroll
                for (int q = 0; q < WQ; q++) {
                    const unsigned long long pr = ((unsigned long long)rw[q + 1] << 32) | rw[q];
                    if (r < CH) accA = __builtin_amdgcn_qsad_pk_u16_u8(pr, sreg[r < CH ? r : 0][q], accA);
                    if (r >= 1) accB = __builtin_amdgcn_qsad_pk_u16_u8(pr, sreg[r >= 1 ? r - 1 : 0][q], accB);
                }
            }
            const int xs0 = xg * 4;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                if (xs0 + jj < search_w) {
                    const unsigned sa = (unsigned)((accA >> (16 * jj)) & 0xffffu), sb = (unsigned)((accB >> (16 * jj)) & 0xffffu);
                    const unsigned long long ka = ((unsigned long long)sa << 32) | (unsigned)(ysA * search_w + xs0 + jj);
                    best = ka < best ? ka : best;
                    if (hasB) {
                        const unsigned long long kb = ((unsigned long long)sb << 32) | (unsigned)((ysA + 1) * search_w + xs0 + jj);
                        best = kb < best ? kb : best;
                    }
                }
            }
        }
    }
    for (uint32_t m = lpb >> 1; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(best, (int)m, 64);
        best = o < best ? o : best;
    }
    __syncthreads();
    unsigned long long* s_out = reinterpret_cast<unsigned long long*>(smem);     // staging is dead
    if (l == 0) s_out[slot] = valid ? best : ~0ull;
    __syncthreads();
    if (tid < slots) {
        const uint32_t ob = blockIdx.x * slots + tid;
        if (ob < nblocks) {
            const unsigned long long key = s_out[tid];
            const unsigned sadv = (unsigned)(key >> 32);
            const int cand = (int)(key & 0xffffffffu);
            if (sadv < 0xffffffu) {          // reference initialises best_sad = 0xffffff, strict '<'
                best_sad[ob] = sadv;
                best_x[ob] = (int16_t)(cand % search_w);
                best_y[ob] = (int16_t)(cand / search_w);
            } else {
                best_sad[ob] = 0xffffffu;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// sad_search_q2p_kernel: sad_search_q2_kernel as a PERSISTENT, software-pipelined kernel.  The one-shot form is latency
// bound (3 waves / SIMD by LDS, a wave lives ~8 us for ~1.3 us of v_qsad work: staging loads, a workgroup barrier, then
// the search, nothing overlapping inside a wave).  Here a wave owns its LDS region (no workgroup barrier), loops over
// sets of 64 / lpb blocks, and fetches the NEXT set's source + window chunks into registers (at most 8 x 16 B per lane)
// right after it has handed the current set's chunks to LDS - so every wave keeps global loads in flight for the whole
// time it spends in v_qsad.  Same staging layout, search loop, argmin key and outputs as sad_search_q2_kernel; a wave's
// results leave as one contiguous store per output array (lanes 0 .. bpw-1).
// Host precondition: (source chunks + window chunks) <= 8 * lpb.
// ---------------------------------------------------------------------------
template <int CW, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void sad_search_q2p_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, size_t ref_block_pitch, int search_w, int search_h,
    unsigned long long* __restrict__ best_sad, int16_t* __restrict__ best_x, int16_t* __restrict__ best_y,
    uint32_t wstride, uint32_t wpitch, uint32_t lpb, uint32_t cpr_magic, const uint32_t* __restrict__ src_offs,
    const uint32_t* __restrict__ ref_offs, uint32_t nblocks) {
    static_assert(CW % 4 == 0 && CW * CH <= 256, "source block must fit 64 VGPRs");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int WQ = CW / 4, SU = 8;
    // source blocks sit 16 B apart from a multiple of 256 B: the broadcast b128 reads of a row (one address per block) then
    // fall on different banks (at a stride of 256 B all eight blocks of a wave hit the same four: PMC showed 38 % of the LDS
    // cycles as bank conflicts)
    constexpr uint32_t SRC_BYTES = ((CW * CH + 15) & ~15) + 16;
    constexpr uint32_t CS = CW % 16 == 0 ? 16 : (CW % 8 == 0 ? 8 : 4);
    constexpr uint32_t NSRC = CW * CH / CS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lsh = __builtin_ctz(lpb), bpw = 64u >> lsh;
    const uint32_t wslot = lane >> lsh, l = lane & (lpb - 1);
    const uint32_t win_w = CW + search_w - 1;
    const uint32_t nrows = (uint32_t)(search_h + CH - 1);
    const uint32_t cpr = (win_w + 15) >> 4;
    const uint32_t nref = nrows * cpr;
    const size_t span = (size_t)(nrows - 1) * ref_stride + win_w;
    // LDS of a wave: its source blocks, then its windows at a row pitch of wpitch = 16 * cpr + 8 bytes and a block stride
    // wstride == 64 (mod 128) bytes, every second pair of blocks 8 bytes further on: the window origins of four consecutive
    // blocks then sit at dword residues {0, 16, 2, 18} (mod 32), and with the 40-B pitch of a 23-wide window the 32 lanes of a
    // ds_read_b32 group (4 blocks x 4 row pairs x 2 column groups) hit 32 different banks.  920 B per 16x16 / 8x8-search
    // window instead of 1 288: 4 workgroups (16 waves) per CU instead of 3.
    uint8_t* wbase = smem + (size_t)wave * bpw * (SRC_BYTES + wstride);
    uint8_t* s_src = wbase + (size_t)wslot * SRC_BYTES;
    uint8_t* s_ref = wbase + (size_t)bpw * SRC_BYTES + (size_t)wslot * wstride + 8u * ((wslot >> 1) & 1u);
    const uint32_t nsets = (nblocks + bpw - 1) >> __builtin_ctz(bpw);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    uint32_t set = blockIdx.x * (blockDim.x >> 6) + wave;

    // chunk k of this lane is list entry i = l + k * lpb: entries [0, NSRC) are source chunks of CS bytes, the rest the
    // window's 16-B chunks (row-major, cpr per row).  Where a chunk comes from (byte offset from the block's source / window
    // origin) and where it goes in LDS depend on the lane only: computed once, 2 VGPRs per chunk.  Loads are unconditional
    // (entries past the list repeat the last one and are not stored).
    uint32_t goff[SU], ldst[SU];             // ldst: byte offset in smem | kind in bits 30-31 (0 none, 1 source, 2 window, 3 window tail)
#pragma unroll
    for (int k = 0; k < SU; k++) {
        const uint32_t iw = l + k * lpb, i = min(iw, NSRC + nref - 1);
        const bool want = iw < NSRC + nref;
        if (i < NSRC) {
            goff[k] = (i / (CW / CS)) * src_stride + (i % (CW / CS)) * CS;
            ldst[k] = (uint32_t)(s_src + i * CS - smem) | (want ? 1u << 30 : 0u);
        } else {
            const uint32_t j = i - NSRC;
            const uint32_t rr = cpr == 1 ? j : __umulhi(j, cpr_magic), c = j - rr * cpr;
            const size_t off = (size_t)rr * ref_stride + c * 16;
            const bool tail = off + 16 > span;              // footprint tail: the LAST 16 bytes of the footprint instead, stored earlier
            const uint32_t delta = tail ? (uint32_t)(off + 16 - span) : 0u;
            goff[k] = (uint32_t)off - delta;
            ldst[k] = (uint32_t)(s_ref + rr * wpitch + c * 16 - delta - smem) | (want ? (tail ? 3u : 2u) << 30 : 0u);
        }
    }
    // the kind of chunk k is the same in every lane for most k (a block's source chunks fill whole rounds of lanes, the tail
    // and the end of the list touch one round each): kept as a scalar so that the hand-over below branches without VALU work
    uint32_t ukind[SU];
#pragma unroll
    for (int k = 0; k < SU; k++) {
        const uint32_t kd = ldst[k] >> 30, k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)kd);
        ukind[k] = __builtin_amdgcn_ballot_w64(kd != k0) == 0 ? k0 : 4u;
        ldst[k] &= 0x3fffffffu;
        if (ukind[k] == 4u) ldst[k] |= kd << 30;
    }
    const bool k_is_src[SU] = {l < NSRC, l + lpb < NSRC, l + 2 * lpb < NSRC, l + 3 * lpb < NSRC, l + 4 * lpb < NSRC, l + 5 * lpb < NSRC, l + 6 * lpb < NSRC, l + 7 * lpb < NSRC};
    uint4 v[SU];
    auto issue = [&](uint32_t st) {
        const uint32_t b = min(st * bpw + wslot, nblocks - 1);
        const uint8_t* gs = src + (src_offs ? (size_t)src_offs[b] : (size_t)b * src_block_pitch);
        const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[b] : (size_t)b * ref_block_pitch);
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint8_t* p = (k_is_src[k] ? gs : gr) + goff[k];
            if constexpr (CS == 16) {
                __builtin_memcpy(&v[k], p, 16);
            } else {                                        // 8-B source chunks: never read past the source block
                v[k] = make_uint4(0, 0, 0, 0);
                if (k_is_src[k]) __builtin_memcpy(&v[k], p, CS); else __builtin_memcpy(&v[k], p, 16);
            }
        }
    };
    if (set < nsets) issue(set);
    for (; set < nsets; set += nwaves) {
        // ---- hand the prefetched chunks to LDS (the previous set's reads are complete: same wave, program order) ----
        asm volatile("" ::"v"(v[0].x), "v"(v[1].x), "v"(v[2].x), "v"(v[3].x), "v"(v[4].x), "v"(v[5].x), "v"(v[6].x), "v"(v[7].x));
#pragma unroll
        for (int k = 0; k < SU; k++) {
            auto put = [&](uint32_t kind, uint8_t* d) {
                if (kind == 1) {
                    __builtin_memcpy(d, &v[k], CS);
                } else if (kind == 2) {                    // two 8-B halves (window origins are 8-B aligned only)
                    uint2* rd = reinterpret_cast<uint2*>(d);
                    rd[0] = make_uint2(v[k].x, v[k].y); rd[1] = make_uint2(v[k].z, v[k].w);
                } else if (kind == 3) {                    // byte-granular address: gfx950 LDS takes the unaligned 16-B store
                    struct __attribute__((packed, aligned(1))) U4 { uint32_t a, b, c, e; };
                    *reinterpret_cast<U4*>(d) = U4{v[k].x, v[k].y, v[k].z, v[k].w};
                }
            };
            const uint32_t uk = (uint32_t)__builtin_amdgcn_readfirstlane((int)ukind[k]);      // scalar: s_cmp + s_cbranch below
            if (uk == 1u) put(1u, smem + ldst[k]);
            else if (uk == 2u) put(2u, smem + ldst[k]);
            else if (uk == 3u) put(3u, smem + ldst[k]);
            else if (uk == 4u) put(ldst[k] >> 30, smem + (ldst[k] & 0x3fffffffu));
        }
        wave_lds_fence();
        // ---- next set's chunks: in flight during the search below ----------------------------------------------------
        if (set + nwaves < nsets) issue(set + nwaves);
        // ---- search (sad_search_q2_kernel) -------------------------------------------------------------------------------
        const uint32_t blk = set * bpw + wslot;
        const bool valid = blk < nblocks;
        // argmin key = sad << 16 | y << 8 | x of the candidate (a 16x16 SAD is at most 65 280; the host sends search areas
        // wider or taller than 256 to the one-shot kernel): one v_min_u32 per candidate, first strict minimum as the reference
        uint32_t best = 0xffffffffu;
        {
            const int gx = (search_w + 3) >> 2, gy = (search_h + 1) >> 1;
            const int ngroups = gx * gy;
            for (int g = (int)l; g < ngroups; g += (int)lpb) {
                const int yp = g / gx, xg = g - yp * gx;
                const int ysA = 2 * yp;
                const bool hasB = ysA + 1 < search_h;
                unsigned long long accA = 0, accB = 0;
                const uint8_t* rbase = s_ref + (size_t)ysA * wpitch + xg * 4;
                // source rows are streamed from LDS (broadcast reads, the lanes of a block share the address): two rows live
                // instead of the whole block in 64 VGPRs - the registers hold the next set's chunks meanwhile
                uint32_t sprev[WQ];
                auto load_src = [&](int r, uint32_t (&d)[WQ]) {
                    if constexpr (WQ % 4 == 0) {
#pragma unroll
                        for (int i = 0; i < WQ / 4; i++) {
                            const uint4 a = *reinterpret_cast<const uint4*>(s_src + r * CW + 16 * i);
                            d[4 * i] = a.x; d[4 * i + 1] = a.y; d[4 * i + 2] = a.z; d[4 * i + 3] = a.w;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < WQ; q++) d[q] = *reinterpret_cast<const uint32_t*>(s_src + r * CW + 4 * q);
                    }
                };
                {   // window row 0 feeds search row A only
                    const uint32_t* rrow = reinterpret_cast<const uint32_t*>(rbase);
                    uint32_t rw[WQ + 1];
#pragma unroll
                    for (int q = 0; q <= WQ; q++) rw[q] = rrow[q];
                    load_src(0, sprev);
#pragma unroll
                    for (int q = 0; q < WQ; q++) accA = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)rw[q + 1] << 32) | rw[q], sprev[q], accA);
                }
                // (a bounded unroll: fully unrolled, the scheduler hoists all 17 rows' LDS reads and the kernel needs 280 VGPRs)
#pragma unroll 4
                for (int r = 1; r < CH; r++) {
                    const uint32_t* rrow = reinterpret_cast<const uint32_t*>(rbase + (size_t)r * wpitch);
                    uint32_t rw[WQ + 1], scur[WQ];
#pragma unroll
                    for (int q = 0; q <= WQ; q++) rw[q] = rrow[q];
                    load_src(r, scur);
#pragma unroll
                    for (int q = 0; q < WQ; q++) {
                        const unsigned long long pr = ((unsigned long long)rw[q + 1] << 32) | rw[q];
                        accA = __builtin_amdgcn_qsad_pk_u16_u8(pr, scur[q], accA);
                        accB = __builtin_amdgcn_qsad_pk_u16_u8(pr, sprev[q], accB);
                    }
#pragma unroll
                    for (int q = 0; q < WQ; q++) sprev[q] = scur[q];
                }
                {   // window row CH feeds search row B only (past the window when the pair has no second row: re-read the last one)
                    const uint32_t* rrow = reinterpret_cast<const uint32_t*>(rbase + (size_t)(hasB ? CH : CH - 1) * wpitch);
                    uint32_t rw[WQ + 1];
#pragma unroll
                    for (int q = 0; q <= WQ; q++) rw[q] = rrow[q];
#pragma unroll
                    for (int q = 0; q < WQ; q++) accB = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)rw[q + 1] << 32) | rw[q], sprev[q], accB);
                }
                const int xs0 = xg * 4;
                const uint32_t idxA = (uint32_t)((ysA << 8) | xs0), idxB = idxA + 256u;      // y << 8 | x: raster order, no division
                const uint32_t aw[2] = {(uint32_t)accA, (uint32_t)(accA >> 32)}, bw[2] = {(uint32_t)accB, (uint32_t)(accB >> 32)};
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const bool in = xs0 + jj < search_w;
                    const uint32_t sa = (jj & 1) ? (aw[jj >> 1] & 0xffff0000u) : (aw[jj >> 1] << 16);
                    const uint32_t sb = (jj & 1) ? (bw[jj >> 1] & 0xffff0000u) : (bw[jj >> 1] << 16);
                    const uint32_t ka = in ? (sa | (idxA + jj)) : 0xffffffffu;
                    const uint32_t kb = (in && hasB) ? (sb | (idxB + jj)) : 0xffffffffu;
                    best = min(best, min(ka, kb));
                }
            }
        }
        for (uint32_t m = lpb >> 1; m >= 1; m >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, (int)m, 64));
        // lane t < bpw takes block slot t's result: contiguous stores
        const uint32_t key = (uint32_t)__shfl((int)best, (int)((lane & (bpw - 1)) << lsh), 64);
        const uint32_t ob = set * bpw + lane;
        if (lane < bpw && ob < nblocks) {
            best_sad[ob] = key >> 16;                     // (a SAD below the reference's initial best 0xffffff always exists)
            best_x[ob] = (int16_t)(key & 0xffu);
            best_y[ob] = (int16_t)((key >> 8) & 0xffu);
        }
        (void)valid;
        wave_lds_fence();                                  // the search's LDS reads are done before the next set overwrites
    }
}

// ---------------------------------------------------------------------------
// SAD search for 16-, 32- and 64-wide blocks (any height that is a multiple of 256 / CW), plain reference
// window.  sad_search_q_kernel issues two ds_read_b32 per v_qsad_pk_u16_u8 (a reference and a source
// dword), which makes the LDS pipe - shared by the CU's four SIMDs, 128 B/clk - its limiter at about
// a third of the v_qsad rate.  Here a lane owns SIXTEEN horizontally adjacent candidates of one search
// row (as me_sb_search16_kernel does): per block row it reads CW/16 + 1 ALIGNED ds_read_b128 of the
// window and CW/16 broadcast ds_read_b128 of the source row and feeds CW v_qsad from them - 2.25 LDS
// bytes per qsad instead of 8.  Packed u16 sums are widened every 256 / CW rows (CW * 255 * 256 / CW
// < 2^16).  The window's row pitch is an odd multiple of 16 bytes so that the lanes of consecutive
// search rows read different bank groups.
// ---------------------------------------------------------------------------
template <int CW>
__global__ __launch_bounds__(256) void sad_search_q16_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, size_t ref_block_pitch, uint32_t height, int search_w, int search_h,
    unsigned long long* __restrict__ best_sad, int16_t* __restrict__ best_x, int16_t* __restrict__ best_y,
    uint32_t wpitch, uint32_t ref_lds_bytes, uint32_t lpb, uint32_t tsh, uint32_t cpr_magic,
    const uint32_t* __restrict__ src_offs, const uint32_t* __restrict__ ref_offs, uint32_t nblocks) {
    static_assert(CW == 16 || CW == 32 || CW == 64, "block width");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int WQ = CW / 4, WC = CW / 16, FL = 256 / CW;
    const uint32_t tid = threadIdx.x;
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slot = tid >> lsh, l = tid & (lpb - 1);
    const uint32_t slots = blockDim.x >> lsh;
    const uint32_t blk = blockIdx.x * slots + slot;
    const bool valid = blk < nblocks;
    const uint32_t win_w = CW + search_w - 1;
    const uint32_t nrows = (uint32_t)search_h + height - 1;
    const uint32_t src_bytes = CW * height;
    uint8_t* s_src = smem + (size_t)slot * (src_bytes + ref_lds_bytes);
    uint8_t* s_ref = s_src + src_bytes;
    if (valid) {
        const uint8_t* gs = src + (src_offs ? (size_t)src_offs[blk] : (size_t)blk * src_block_pitch);
        const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[blk] : (size_t)blk * ref_block_pitch);
        // staging as in sad_search_q2_kernel: one flat list of 16-B chunks, SU unconditional (clamped) loads
        // per lane issued back to back, then the LDS writes - one memory latency per SU chunks
        constexpr int SU = 4;
        const uint32_t nsrc = WC * height;
        const uint32_t cpr = (win_w + 15) >> 4;
        const uint32_t nref = nrows * cpr;
        const size_t span = (size_t)(nrows - 1) * ref_stride + win_w;
        for (uint32_t i0 = l; i0 < nsrc || i0 < nref; i0 += SU * lpb) {
            uint4 vs[SU], vr[SU];
            uint32_t rdst[SU], tdst[SU];
#pragma unroll
            for (int k = 0; k < SU; k++) {
                const uint32_t i = min(i0 + k * lpb, nsrc - 1);
                __builtin_memcpy(&vs[k], gs + (size_t)(i / WC) * src_stride + (i % WC) * 16, 16);
            }
#pragma unroll
            for (int k = 0; k < SU; k++) {
                const uint32_t j = min(i0 + k * lpb, nref - 1);
                const uint32_t rr = __umulhi(j, cpr_magic), c = j - rr * cpr;     // j / cpr, j % cpr (cpr >= 2)
                const size_t off = (size_t)rr * ref_stride + c * 16;
                // a chunk that would end past the window's footprint is fetched as the LAST 16 bytes of the
                // footprint instead and stored `delta` bytes earlier (the bytes it re-writes are the same data)
                const size_t offc = off + 16 <= span ? off : span - 16;
                __builtin_memcpy(&vr[k], gr + offc, 16);
                const bool want = i0 + k * lpb < nref;
                rdst[k] = (want && off == offc) ? rr * wpitch + c * 16 : ~0u;
                tdst[k] = (want && off != offc) ? rr * wpitch + c * 16 - (uint32_t)(off - offc) : ~0u;
            }
            static_assert(SU == 4, "operand list below");
            asm volatile("" ::"v"(vs[0].x), "v"(vs[1].x), "v"(vs[2].x), "v"(vs[3].x), "v"(vr[0].x), "v"(vr[1].x), "v"(vr[2].x), "v"(vr[3].x));
#pragma unroll
            for (int k = 0; k < SU; k++) {
                const uint32_t i = i0 + k * lpb;
                if (i < nsrc) *reinterpret_cast<uint4*>(s_src + i * 16) = vs[k];
                if (rdst[k] != ~0u) *reinterpret_cast<uint4*>(s_ref + rdst[k]) = vr[k];
            }
#pragma unroll
            for (int k = 0; k < SU; k++)
                if (tdst[k] != ~0u) {                      // footprint tail: byte-granular LDS store, no global access
                    const uint8_t* vb = reinterpret_cast<const uint8_t*>(&vr[k]);
                    for (int bb = 0; bb < 16; bb++) s_ref[tdst[k] + bb] = vb[bb];
                }
        }
    }
    __syncthreads();
    unsigned long long best = ~0ull;
    if (valid) {
        // lane = row split * TP + task: the lanes of one 8-lane LDS group are consecutive search rows (window
        // pitch = odd multiple of 16 B: conflict-free), and a block with few tasks still fills 64 lanes by
        // splitting the block's row groups over RS lanes per task (sums are added across them below)
        const uint32_t TP = 1u << tsh, RS = lpb >> tsh;
        const uint32_t t0 = l & (TP - 1), rs = l >> tsh;
        const int xqn = (search_w + 15) >> 4;
        const int ntasks = xqn * search_h;
        for (int tb = 0; tb < ntasks; tb += (int)TP) {
            const int t = tb + (int)t0;
            const bool act = t < ntasks;
            const int tc = act ? t : 0;
            const int ys = tc / xqn, xs0 = (tc - ys * xqn) * 16;
            const uint8_t* rbase = s_ref + (size_t)ys * wpitch + xs0;
            uint32_t sum[16];
#pragma unroll
            for (int i = 0; i < 16; i++) sum[i] = 0;
            for (uint32_t y0 = rs * FL; y0 < height; y0 += RS * FL) {
                unsigned long long acc[4] = {0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < FL; r++) {
                    const uint4* sp = reinterpret_cast<const uint4*>(s_src + (size_t)(y0 + r) * CW);
                    const uint4* rp = reinterpret_cast<const uint4*>(rbase + (size_t)(y0 + r) * wpitch);
                    uint32_t sw[WQ], rw[WQ + 4];
#pragma unroll
                    for (int i = 0; i < WC; i++) { const uint4 a = sp[i]; sw[4 * i] = a.x; sw[4 * i + 1] = a.y; sw[4 * i + 2] = a.z; sw[4 * i + 3] = a.w; }
#pragma unroll
                    for (int i = 0; i < WC + 1; i++) { const uint4 a = rp[i]; rw[4 * i] = a.x; rw[4 * i + 1] = a.y; rw[4 * i + 2] = a.z; rw[4 * i + 3] = a.w; }
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int q = 0; q < WQ; q++)
                            acc[g] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)rw[g + q + 1] << 32) | rw[g + q], sw[q], acc[g]);
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    sum[4 * g] += (uint32_t)(acc[g] & 0xffffu); sum[4 * g + 1] += (uint32_t)((acc[g] >> 16) & 0xffffu);
                    sum[4 * g + 2] += (uint32_t)((acc[g] >> 32) & 0xffffu); sum[4 * g + 3] += (uint32_t)(acc[g] >> 48);
                }
            }
            for (uint32_t m = TP; m < lpb; m <<= 1) {
#pragma unroll
                for (int i = 0; i < 16; i++) sum[i] += (uint32_t)__shfl_xor((int)sum[i], (int)m, 64);
            }
            const int cbase = ys * search_w + xs0;
            if (act) {
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (xs0 + j < search_w) {
                        const unsigned long long key = ((unsigned long long)sum[j] << 32) | (unsigned)(cbase + j);
                        best = key < best ? key : best;
                    }
            }
        }
    }
    for (uint32_t m = lpb >> 1; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(best, (int)m, 64);
        best = o < best ? o : best;
    }
    __syncthreads();
    unsigned long long* s_out = reinterpret_cast<unsigned long long*>(smem);     // staging is dead
    if (l == 0) s_out[slot] = valid ? best : ~0ull;
    __syncthreads();
    if (tid < slots) {
        const uint32_t ob = blockIdx.x * slots + tid;
        if (ob < nblocks) {
            const unsigned long long key = s_out[tid];
            const unsigned sadv = (unsigned)(key >> 32);
            const int cand = (int)(key & 0xffffffffu);
            if (sadv < 0xffffffu) {          // reference initialises best_sad = 0xffffff, strict '<'
                best_sad[ob] = sadv;
                best_x[ob] = (int16_t)(cand % search_w);
                best_y[ob] = (int16_t)(cand / search_w);
            } else {
                best_sad[ob] = 0xffffffu;
            }
        }
    }
}

// full_distortion_kernel32_bits / _cbf_zero32_bits (EbPictureOperators.c:283-346):
// out[blk][0] = sum (c - r)^2 (or sum c^2 when cbf_zero), out[blk][1] = sum c^2.
// 16 lanes per block, 4 blocks per wave.  nz != NULL: per-block choice as picture_full_distortion32_bits makes it
// (:377-395: count_non_zero_coeffs == 0 -> the cbf_zero kernel).  AVX2: the residual term as the reference's AVX2 kernel
// accumulates it (EbPictureOperators_Intrinsic_AVX2.c:1955-2011): four 64-bit lanes by column mod 4, the square taken from
// the low 32 bits of the difference, low and high halves of a lane summed separately with 32-bit adds (no carry).
template <bool AVX2>
__global__ __launch_bounds__(256) void full_distortion32_kernel(
    const int32_t* __restrict__ coeff, uint32_t coeff_stride, size_t coeff_block_pitch,
    const int32_t* __restrict__ recon, uint32_t recon_stride, size_t recon_block_pitch, uint32_t width,
    uint32_t height, int cbf_zero, const uint32_t* __restrict__ nz, unsigned long long* __restrict__ out,
    uint32_t nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, l = lane & 15;
    const uint32_t blk = (blockIdx.x * 4 + wave) * 4 + sub;
    const bool valid = blk < nblocks;
    unsigned long long resid = 0, pred = 0;
    unsigned lo = 0, hi = 0;
    const bool zero = cbf_zero || (nz && valid && nz[blk] == 0);
    if (valid) {
        const int32_t* pc = coeff + (size_t)blk * coeff_block_pitch;
        const int32_t* pr = zero ? nullptr : recon + (size_t)blk * recon_block_pitch;
        const uint32_t total = width * height;
        for (uint32_t i = l; i < total; i += 16) {
            const uint32_t y = i / width, x = i - y * width;
            const long long c = pc[(size_t)y * coeff_stride + x];
            pred += (unsigned long long)(c * c);
            if (!zero) {
                const long long d = c - (long long)pr[(size_t)y * recon_stride + x];
                if (AVX2) {
                    const long long dl = (int)(unsigned)(unsigned long long)d;
                    const unsigned long long sq = (unsigned long long)(dl * dl);
                    lo += (unsigned)sq; hi += (unsigned)(sq >> 32);       // width % 4 == 0: all of a lane's x share x & 3 = l & 3
                } else {
                    resid += (unsigned long long)(d * d);
                }
            }
        }
    }
    if (AVX2) {
        // lanes l, l ^ 4, l ^ 8, l ^ 12 hold the same column class: wrapping 32-bit sums, then 64-bit over the 4 classes
        lo += (unsigned)__shfl_xor((int)lo, 4, 64); hi += (unsigned)__shfl_xor((int)hi, 4, 64);
        lo += (unsigned)__shfl_xor((int)lo, 8, 64); hi += (unsigned)__shfl_xor((int)hi, 8, 64);
        resid = ((unsigned long long)hi << 32) | lo;
        resid += __shfl_xor(resid, 1, 64);
        resid += __shfl_xor(resid, 2, 64);
    } else {
        resid = group_sum64<16>(resid);
    }
    pred = group_sum64<16>(pred);
    if (valid && l == 0) {
        out[(size_t)blk * 2 + 0] = zero ? pred : resid;
        out[(size_t)blk * 2 + 1] = pred;
    }
}

}  // namespace svtdev
