// kernel_fused32.h — tuned 32x32 kernels (the sizes the headline metric and
// BASELINE.json configs[1] are quoted on).
//
//   fwd32_kernel<IN_U8, QUANT, WITH_SAD, ...>
//     IN_U8 + QUANT + WITH_SAD   headline: residual(src,pred) -> FwdTxfm2d -> quantize_b_32x32 -> SAD
//     int16 in + QUANT           configs[1]: FwdTxfm2d + quantize on a residual batch
//     int16 in, no QUANT         plain av1_fwd_txfm2d_32x32
//   inv32_kernel<PixT>           av1_inv_txfm2d_add_32x32 / av1_inv_txfm_add (8-bit recon)
//
// Replaces, per block, the reference call sequence
//   ResidualKernel                (EbCodingLoop.c:617  -> EbPictureOperators.c:166)
//   av1_fwd_txfm2d_32x32          (EbFullLoop.c:763    -> EbTransforms.c:4466 / AVX2 :4080)
//   aom_highbd_quantize_b_32x32   (EbFullLoop.c:780    -> EbFullLoop.c:239 / AVX2 :422)
//   NxMSadKernel 32x32            (EbProductCodingLoop.c:1259 -> EbComputeSAD_C.c:48)
//   av1_inv_txfm2d_add_32x32      (EbTransforms.c:8293 -> inv_txfm2d_add_c :8180)
//
// Mapping (CDNA4, wave64): one wave owns TWO blocks (lanes 0-31 / 32-63).  In
// the first pass lane c holds column c in 32 VGPRs and runs the straight-line
// generated DCT32; a swizzled LDS tile transposes; in the second pass lane r holds
// row r.  A second swizzled tile re-orders into the linear block layout so that
// quantisation happens on, and every 4 KB output is stored from, fully coalesced
// 16-B-per-lane positions.  Each wave uses a private LDS region: no workgroup
// barrier anywhere.  All LDS access patterns are conflict-free
// (SQ_LDS_BANK_CONFLICT = 0 measured, profiles/r01_a_pmc.json).
//
// HBM traffic per block (headline) = 2 x 1024 B in + 3 x 4096 + 2 + 4 B out =
// 14 342 B — the algorithmic minimum; everything else is VGPR/LDS.
#pragma once
#include "dev_common.h"
#include "gen/txfm1d_gen.h"

namespace svtdev {

constexpr int F32_WAVES = 4;                 // waves per workgroup
constexpr int F32_TILE_WORDS = 1024;         // one 32x32 int32 tile per block
constexpr int F32_COS_BIT = 12;              // fwd_cos_bit_col/row[3][3] (EbTransforms.h:141-156)

// LDS byte offset of 16-B slot `s` (0..7) of row `r` in a 32x32 int32 tile whose
// slots are XOR-swizzled by f: conflict-free for the access pairs used below.
__device__ __forceinline__ int tile_slot(int r, int s, int f) { return r * 128 + ((s ^ f) << 4); }

template <bool IN_U8, bool QUANT, bool WITH_SAD, int MIN_WAVES_PER_SIMD = 1, bool NT = false, int QMODE = 2>
__global__ __launch_bounds__(F32_WAVES * 64, MIN_WAVES_PER_SIMD) void fwd32_kernel(
    const void* __restrict__ src_v, const uint8_t* __restrict__ pred, int32_t* __restrict__ coeff,
    int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob,
    uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, QParams qp, int is_idtx, uint32_t nblocks) {
    __shared__ __attribute__((aligned(16))) int32_t lds[F32_WAVES * 2 * F32_TILE_WORDS];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;      // which of the wave's two blocks
    const int li = lane & 31;        // column index (pass 1) / row index (pass 2)
    char* tile = reinterpret_cast<char*>(lds + (wave * 2 + half) * F32_TILE_WORDS);

    // iscan+1 for the 32 linear positions this lane quantises: position
    // (k*32 + li)*4 + j, k = 0..7, j = 0..3  (same for every block).
    uint2 isc[8];
    if (QUANT) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            isc[k] = *reinterpret_cast<const uint2*>(iscan + (k * 32 + li) * 4);
            isc[k].x += 0x00010001u;   // iscan <= 1023: no carry between the packed halves
            isc[k].y += 0x00010001u;
        }
    }

    const uint32_t npairs = (nblocks + 1) >> 1;
    const uint32_t wave_stride = gridDim.x * F32_WAVES;
    for (uint32_t pair = blockIdx.x * F32_WAVES + wave; pair < npairs; pair += wave_stride) {
        const uint32_t blk = pair * 2 + half;
        const bool valid = blk < nblocks;
        const size_t pix_off = (size_t)blk * 1024;
        unsigned sad_acc = 0;
        int x[32];

        if (IN_U8) {
            // ---- load 2 x 1 KB, coalesced 16 B per lane ------------------------------
            const uint8_t* src = static_cast<const uint8_t*>(src_v);
            uint4 s0 = {0, 0, 0, 0}, s1 = s0, p0 = s0, p1 = s0;
            if (valid) {
                const uint4* s4 = reinterpret_cast<const uint4*>(src + pix_off);
                const uint4* p4 = reinterpret_cast<const uint4*>(pred + pix_off);
                s0 = s4[li]; s1 = s4[li + 32]; p0 = p4[li]; p1 = p4[li + 32];
            }
            // ---- SAD on the raw bytes (v_sad_u8: 4 pixels per instruction) -----------
            if (WITH_SAD) {
                sad_acc = __builtin_amdgcn_sad_u8(s0.x, p0.x, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s0.y, p0.y, sad_acc);
                sad_acc = __builtin_amdgcn_sad_u8(s0.z, p0.z, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s0.w, p0.w, sad_acc);
                sad_acc = __builtin_amdgcn_sad_u8(s1.x, p1.x, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s1.y, p1.y, sad_acc);
                sad_acc = __builtin_amdgcn_sad_u8(s1.z, p1.z, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s1.w, p1.w, sad_acc);
            }
            // ---- residual as packed int16 pairs -> LDS --------------------------------
            // lane li, chunk k holds row k*16 + li/2, columns (li&1)*16 .. +15.  Part p
            // (8 residuals = 16 B) goes to byte (2k+p)*544 + li*16: linear (conflict-free)
            // stores; the 544-B part stride keeps the column reads conflict-free too.
            const uint32_t sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const uint32_t pw[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int kp = 0; kp < 4; kp++) {   // kp = 2k + p
                uint32_t r[4];
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    const uint32_t a = sw[kp * 2 + h2], b = pw[kp * 2 + h2];
                    const int d0 = (int)(a & 0xff) - (int)(b & 0xff);
                    const int d1 = (int)((a >> 8) & 0xff) - (int)((b >> 8) & 0xff);
                    const int d2 = (int)((a >> 16) & 0xff) - (int)((b >> 16) & 0xff);
                    const int d3 = (int)(a >> 24) - (int)(b >> 24);
                    r[h2 * 2 + 0] = ((uint32_t)d0 & 0xffffu) | ((uint32_t)d1 << 16);
                    r[h2 * 2 + 1] = ((uint32_t)d2 & 0xffffu) | ((uint32_t)d3 << 16);
                }
                *reinterpret_cast<uint4*>(tile + kp * 544 + li * 16) = make_uint4(r[0], r[1], r[2], r[3]);
            }
            wave_lds_fence();
            // element (r, li): writer lane (r&15)*2 + (li>>4), part (li>>3)&1, item li&7
            const char* colbase = tile + ((li >> 3) & 1) * 544 + (li >> 4) * 16 + (li & 7) * 2;
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const short v = *reinterpret_cast<const short*>(colbase + (r >> 4) * 1088 + (r & 15) * 32);
                x[r] = (int)v * 4;                                   // shift[0] = 2 (fwd_shift_32x32)
            }
        } else {
            // ---- int16 residual, 2 KB per block: chunk k (0..3) of lane li = row k*8 + li/4,
            // columns (li&3)*8 .. +7, stored linearly at k*512 + li*16; column c of row r is
            // then at (r>>3)*512 + (r&7)*64 + c*2 : a 64-B contiguous run per row (no conflicts)
            const int16_t* res = static_cast<const int16_t*>(src_v);
            const uint4* r4 = reinterpret_cast<const uint4*>(res + pix_off);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = valid ? r4[k * 32 + li] : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(tile + k * 512 + li * 16) = v;
            }
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const short v = *reinterpret_cast<const short*>(tile + (r >> 3) * 512 + (r & 7) * 64 + li * 2);
                x[r] = (int)v * 4;
            }
        }
        // ---- column pass: lane li owns column li ------------------------------------
        if (is_idtx) svtgen::svt_fidentity32<F32_COS_BIT>(x); else svtgen::svt_fdct32<F32_COS_BIT>(x);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = (x[r] + 8) >> 4;      // shift[1] = -4
        wave_lds_fence();
        // ---- transpose 1: write column, read row (swizzle f = (row>>1)&7) ------------
#pragma unroll
        for (int r = 0; r < 32; r++)
            *reinterpret_cast<int*>(tile + tile_slot(r, li >> 2, (r >> 1) & 7) + (li & 3) * 4) = x[r];
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int4 v = *reinterpret_cast<const int4*>(tile + tile_slot(li, s, (li >> 1) & 7));
            x[s * 4 + 0] = v.x; x[s * 4 + 1] = v.y; x[s * 4 + 2] = v.z; x[s * 4 + 3] = v.w;
        }
        // ---- row pass: lane li owns row li (shift[2] = 0) ---------------------------
        if (is_idtx) svtgen::svt_fidentity32<F32_COS_BIT>(x); else svtgen::svt_fdct32<F32_COS_BIT>(x);
        wave_lds_fence();
        // ---- transpose 2: rows -> linear block order (swizzle f = row&7) --------------
#pragma unroll
        for (int s = 0; s < 8; s++)
            *reinterpret_cast<int4*>(tile + tile_slot(li, s, li & 7)) =
                make_int4(x[s * 4 + 0], x[s * 4 + 1], x[s * 4 + 2], x[s * 4 + 3]);
        wave_lds_fence();
        int4* co4 = reinterpret_cast<int4*>(coeff + pix_off);
        int4* qc4 = reinterpret_cast<int4*>(qcoeff + pix_off);
        int4* dq4 = reinterpret_cast<int4*>(dqcoeff + pix_off);
        int eob_acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int row = 4 * k + (li >> 3);
            const int4 c = *reinterpret_cast<const int4*>(tile + tile_slot(row, li & 7, row & 7));
            if (!QUANT) {
                if (valid) co4[k * 32 + li] = c;
                continue;
            }
            int4 q, d;
            // only linear position 0 (k == 0, li == 0, .x) uses the DC entries
            quant_one<QMODE>(c.x, (k == 0 && li == 0) ? 0 : 1, qp, q.x, d.x);
            quant_one<QMODE>(c.y, 1, qp, q.y, d.y);
            quant_one<QMODE>(c.z, 1, qp, q.z, d.z);
            quant_one<QMODE>(c.w, 1, qp, q.w, d.w);
            const int e0 = q.x ? (int)(isc[k].x & 0xffffu) : 0, e1 = q.y ? (int)(isc[k].x >> 16) : 0;
            const int e2 = q.z ? (int)(isc[k].y & 0xffffu) : 0, e3 = q.w ? (int)(isc[k].y >> 16) : 0;
            eob_acc = max(eob_acc, max(max(e0, e1), max(e2, e3)));
            if (valid) {
                if (NT) {   // streaming outputs are never re-read by this kernel
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(v4i{c.x, c.y, c.z, c.w}, reinterpret_cast<v4i*>(&co4[k * 32 + li]));
                    __builtin_nontemporal_store(v4i{q.x, q.y, q.z, q.w}, reinterpret_cast<v4i*>(&qc4[k * 32 + li]));
                    __builtin_nontemporal_store(v4i{d.x, d.y, d.z, d.w}, reinterpret_cast<v4i*>(&dq4[k * 32 + li]));
                } else {
                    co4[k * 32 + li] = c; qc4[k * 32 + li] = q; dq4[k * 32 + li] = d;
                }
            }
        }
        if (QUANT) {
            // eob = 1 + last scan position with a non-zero level (iscan max)
            eob_acc = half_wave_max(eob_acc);
            if (WITH_SAD) sad_acc = half_wave_sum(sad_acc);
            if (valid && li == 0) {
                eob[blk] = (uint16_t)eob_acc;
                if (WITH_SAD) sad[blk] = sad_acc;
            }
        }
        wave_lds_fence();   // tile is re-used by the next pair
    }
}

// ---------------------------------------------------------------------------
// inverse 32x32 + add (inv_txfm2d_add_c, EbTransforms.c:8180-8265): the mirror of
// the forward kernel.  Coefficients are read linearly (coalesced), a swizzled tile
// hands rows to lanes (row pass: clamp bd+8, idct32, round-shift 2), a second tile
// hands columns to lanes (column pass: clamp max(bd+6,16), idct32, round-shift 4),
// and a third, 16-bit tile puts the residual back in row order so that destination
// samples are read, updated and written 16 B per lane.
// dst block b at dst + (offsets ? offsets[b] : b*block_pitch), row stride dst_stride.
// ---------------------------------------------------------------------------
template <typename PixT, int BD>
__global__ __launch_bounds__(F32_WAVES * 64) void inv32_kernel(
    const int32_t* __restrict__ coeff, PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch,
    const uint32_t* __restrict__ dst_offsets, int is_idtx, uint32_t nblocks) {
    constexpr int bd = BD;    // compile-time ranges: min(max(x, lo), hi) becomes one v_med3_i32
    __shared__ __attribute__((aligned(16))) int32_t lds[F32_WAVES * 2 * F32_TILE_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, li = lane & 31;
    char* tile = reinterpret_cast<char*>(lds + (wave * 2 + half) * F32_TILE_WORDS);
    const uint32_t blk = (blockIdx.x * F32_WAVES + wave) * 2 + half;
    const bool valid = blk < nblocks;
    constexpr int row_bits = bd == 8 ? 16 : (bd == 10 ? 18 : 20);     // av1_gen_inv_stage_range (:5404-5456)
    constexpr int col_bits = bd == 12 ? 18 : 16;
    constexpr int in_lo = -(1 << (bd + 7)), in_hi = (1 << (bd + 7)) - 1;
    constexpr int cin_bits = bd + 6 > 16 ? bd + 6 : 16;
    int x[32];
    // destination samples are fetched up front (their latency hides under the two transform passes)
    constexpr int PPL = 16 / (int)sizeof(PixT);          // pixels per lane per step
    constexpr int STEPS = 1024 / (32 * PPL);
    const size_t dbase = valid ? (dst_offsets ? (size_t)dst_offsets[blk] : (size_t)blk * dst_block_pitch) : 0;
    const bool dst_aligned = (((reinterpret_cast<uintptr_t>(dst) + dbase * sizeof(PixT)) & 15) == 0) && ((dst_stride * (int)sizeof(PixT)) & 15) == 0;
    uint4 dpre[STEPS];
    if (valid && dst_aligned) {
#pragma unroll
        for (int k = 0; k < STEPS; k++) {
            const int p = (k * 32 + li) * PPL;
            dpre[k] = *reinterpret_cast<const uint4*>(dst + dbase + (size_t)(p >> 5) * dst_stride + (p & 31));
        }
    }
    // ---- linear coefficient load -> tile A (row reads: swizzle (row>>1)&7) -------------
    const int4* c4 = reinterpret_cast<const int4*>(coeff + (size_t)blk * 1024);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int4 v = valid ? c4[k * 32 + li] : make_int4(0, 0, 0, 0);
        const int row = 4 * k + (li >> 3);
        *reinterpret_cast<int4*>(tile + tile_slot(row, li & 7, (row >> 1) & 7)) = v;
    }
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int4 v = *reinterpret_cast<const int4*>(tile + tile_slot(li, s, (li >> 1) & 7));
        x[s * 4 + 0] = min(max(v.x, in_lo), in_hi); x[s * 4 + 1] = min(max(v.y, in_lo), in_hi);
        x[s * 4 + 2] = min(max(v.z, in_lo), in_hi); x[s * 4 + 3] = min(max(v.w, in_lo), in_hi);
    }
    // ---- row pass ----------------------------------------------------------------------
    if (is_idtx) svtgen::svt_iidentity32<12>(x, 0, 0);
    else svtgen::svt_idct32<12>(x, -(1 << (row_bits - 1)), (1 << (row_bits - 1)) - 1);
    wave_lds_fence();
    // ---- tile B: row writes (swizzle row&7), column reads --------------------------------
#pragma unroll
    for (int s = 0; s < 8; s++)
        *reinterpret_cast<int4*>(tile + tile_slot(li, s, li & 7)) =
            make_int4((x[s * 4 + 0] + 2) >> 2, (x[s * 4 + 1] + 2) >> 2, (x[s * 4 + 2] + 2) >> 2, (x[s * 4 + 3] + 2) >> 2);  // shift[0] = -2
    wave_lds_fence();
    constexpr int c_lo = -(1 << (cin_bits - 1)), c_hi = (1 << (cin_bits - 1)) - 1;
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const int v = *reinterpret_cast<const int*>(tile + tile_slot(r, li >> 2, r & 7) + (li & 3) * 4);
        x[r] = min(max(v, c_lo), c_hi);
    }
    // ---- column pass -------------------------------------------------------------------
    if (is_idtx) svtgen::svt_iidentity32<12>(x, 0, 0);
    else svtgen::svt_idct32<12>(x, -(1 << (col_bits - 1)), (1 << (col_bits - 1)) - 1);
    wave_lds_fence();
    // ---- tile C: residual as int32 words, row-major (conflict-free column writes) --------
#pragma unroll
    for (int r = 0; r < 32; r++)
        *reinterpret_cast<int*>(tile + r * 128 + li * 4) = (x[r] + 8) >> 4;                             // shift[1] = -4
    wave_lds_fence();
    if (valid) {
        constexpr int maxpix = (1 << bd) - 1;
#pragma unroll
        for (int k = 0; k < STEPS; k++) {
            const int p = (k * 32 + li) * PPL;               // linear pixel index in the block
            const int row = p >> 5, col = p & 31;
            PixT* d = dst + dbase + (size_t)row * dst_stride + col;
            const int* rs = reinterpret_cast<const int*>(tile + p * 4);
            int rv[PPL];
#pragma unroll
            for (int j = 0; j < PPL; j += 4) {
                const int4 t = *reinterpret_cast<const int4*>(rs + j);
                rv[j] = t.x; rv[j + 1] = t.y; rv[j + 2] = t.z; rv[j + 3] = t.w;
            }
            if (dst_aligned) {
                uint4 pv = dpre[k];                          // fetched before the transforms started
                PixT* px = reinterpret_cast<PixT*>(&pv);
#pragma unroll
                for (int j = 0; j < PPL; j++) px[j] = (PixT)min(max((int)px[j] + rv[j], 0), maxpix);
                *reinterpret_cast<uint4*>(d) = pv;
            } else {
#pragma unroll
                for (int j = 0; j < PPL; j++) d[j] = (PixT)min(max((int)d[j] + rv[j], 0), maxpix);
            }
        }
    }
}

}  // namespace svtdev
