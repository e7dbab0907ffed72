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

// IN: 0 = int16 residual (dense), 1 = uint8 src / pred, 2 = uint16 src / pred (10-bit).
// PLANES: blocks are addressed on picture planes: origin (x, y) = (xy[b] & 0xffff, xy[b] >> 16), row strides
// src_stride / pred_stride in samples; otherwise dense 32x32 blocks back to back (the strides fold to 32).
template <int IN, bool QUANT, bool WITH_SAD, int MIN_WAVES_PER_SIMD = 1, bool NT = false, int QMODE = 2, bool PLANES = false>
__global__ __launch_bounds__(F32_WAVES * 64, MIN_WAVES_PER_SIMD) void fwd32_kernel(
    const void* __restrict__ src_v, const void* __restrict__ pred_v, int32_t* __restrict__ coeff,
    int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob,
    uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, QParams qp, int is_idtx, uint32_t nblocks,
    uint32_t src_stride_rt = 32, uint32_t pred_stride_rt = 32, const uint32_t* __restrict__ xy = nullptr) {
    constexpr bool IN_U8 = IN == 1;
    const uint8_t* pred = static_cast<const uint8_t*>(pred_v);
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
        // sample offsets of the block in the source / prediction arrays and their row strides
        const uint32_t sstr = PLANES ? src_stride_rt : 32u, pstr = PLANES ? pred_stride_rt : 32u;
        size_t sbase = pix_off, pbase = pix_off;
        if (PLANES) {
            const uint32_t o = valid ? xy[blk] : 0u;
            sbase = (size_t)(o >> 16) * sstr + (o & 0xffffu);
            pbase = (size_t)(o >> 16) * pstr + (o & 0xffffu);
        }

        if (IN_U8) {
            // ---- load 2 x 1 KB, 16 B per lane: rows li/2 and 16 + li/2, columns (li&1)*16 .. +15 -------
            const uint8_t* src = static_cast<const uint8_t*>(src_v);
            uint4 s0 = {0, 0, 0, 0}, s1 = s0, p0 = s0, p1 = s0;
            if (valid) {
                if (PLANES) {
                    const uint8_t* sp = src + sbase + (size_t)(li >> 1) * sstr + (li & 1) * 16;
                    const uint8_t* pp = pred + pbase + (size_t)(li >> 1) * pstr + (li & 1) * 16;
                    __builtin_memcpy(&s0, sp, 16); __builtin_memcpy(&s1, sp + (size_t)16 * sstr, 16);
                    __builtin_memcpy(&p0, pp, 16); __builtin_memcpy(&p1, pp + (size_t)16 * pstr, 16);
                } else {
                    const uint4* s4 = reinterpret_cast<const uint4*>(src + pix_off);
                    const uint4* p4 = reinterpret_cast<const uint4*>(pred + pix_off);
                    s0 = s4[li]; s1 = s4[li + 32]; p0 = p4[li]; p1 = p4[li + 32];
                }
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
                    // bytes -> 16-bit lanes with v_perm_b32, then packed subtract and the transform's
                    // input up-shift (x4) on the packed words: 4 residuals in 8 instructions
                    const uint32_t a = sw[kp * 2 + h2], b = pw[kp * 2 + h2];
                    const uint32_t a01 = __builtin_amdgcn_perm(0u, a, 0x0c010c00u), a23 = __builtin_amdgcn_perm(0u, a, 0x0c030c02u);
                    const uint32_t b01 = __builtin_amdgcn_perm(0u, b, 0x0c010c00u), b23 = __builtin_amdgcn_perm(0u, b, 0x0c030c02u);
                    r[h2 * 2 + 0] = pk_shl2_i16(pk_sub_i16(a01, b01));
                    r[h2 * 2 + 1] = pk_shl2_i16(pk_sub_i16(a23, b23));
                }
                *reinterpret_cast<uint4*>(tile + kp * 544 + li * 16) = make_uint4(r[0], r[1], r[2], r[3]);
            }
            wave_lds_fence();
            // element (r, li): writer lane (r&15)*2 + (li>>4), part (li>>3)&1, item li&7
            const char* colbase = tile + ((li >> 3) & 1) * 544 + (li >> 4) * 16 + (li & 7) * 2;
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const short v = *reinterpret_cast<const short*>(colbase + (r >> 4) * 1088 + (r & 15) * 32);
                x[r] = (int)v;                                       // shift[0] = 2 (fwd_shift_32x32) already applied
            }
        } else {
            // ---- int16 residual, 2 KB per block: chunk k (0..3) of lane li = row k*8 + li/4,
            // columns (li&3)*8 .. +7, stored linearly at k*512 + li*16; column c of row r is
            // then at (r>>3)*512 + (r&7)*64 + c*2 : a 64-B contiguous run per row (no conflicts)
            const int16_t* res = static_cast<const int16_t*>(src_v);
            if (IN == 2) {
                // 10-bit: the same chunks of the uint16 source and prediction, residual by v_pk_sub_i16
                // (|s - p| <= 1023 fits int16), SAD by v_sad_u16
                const uint16_t* s16 = static_cast<const uint16_t*>(src_v);
                const uint16_t* p16 = static_cast<const uint16_t*>(pred_v);
                uint4 sv[4], pv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    sv[k] = make_uint4(0, 0, 0, 0); pv[k] = sv[k];
                    if (valid) {
                        __builtin_memcpy(&sv[k], s16 + sbase + (size_t)(k * 8 + (li >> 2)) * sstr + (li & 3) * 8, 16);
                        __builtin_memcpy(&pv[k], p16 + pbase + (size_t)(k * 8 + (li >> 2)) * pstr + (li & 3) * 8, 16);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t a[4] = {sv[k].x, sv[k].y, sv[k].z, sv[k].w}, b[4] = {pv[k].x, pv[k].y, pv[k].z, pv[k].w};
                    uint32_t d[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        d[j] = pk_sub_i16(a[j], b[j]);
                        if (WITH_SAD) sad_acc = __builtin_amdgcn_sad_u16(a[j], b[j], sad_acc);
                    }
                    *reinterpret_cast<uint4*>(tile + k * 512 + li * 16) = make_uint4(d[0], d[1], d[2], d[3]);
                }
            } else {
                const uint4* r4 = reinterpret_cast<const uint4*>(res + pix_off);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (valid) v = r4[k * 32 + li];
                    *reinterpret_cast<uint4*>(tile + k * 512 + li * 16) = v;
                }
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


// One 32-point inverse pass of a lane (row or column x[0..31], raw as loaded): input clamp + idct32 / identity.
// Clamp-free fast path: with L1 = sum |x[i]|, every stage clamp of idct32 is a no-op while gain * L1 + slack <= the stage bound
// (txfm_net.clamp_free_bound, constants in gen/txfm1d_gen.h) and, L1 bounding every element, so is the input clamp when
// L1 <= the input bound.  The test is wave-uniform (one ballot); a wave with a louder row / column runs the clamped form.
// 32 v_sad_u32 + xor (~86 issue units) buy 160 v_med3_i32 (~272).  g_tune "no_clamp_free" (FAST = false) keeps the old form.
template <int IN_BITS, int STAGE_BITS, bool FAST>
__device__ __forceinline__ void idct32_pass(int (&x)[32], int is_idtx, int in_lo, int in_hi, int st_lo, int st_hi) {
    constexpr int in_max = (1 << (IN_BITS - 1)) - 1, st_max = (1 << (STAGE_BITS - 1)) - 1;
    constexpr int lim_net = svtgen::svt_clamp_free_l1(svtgen::svt_idct32_gain_q10, svtgen::svt_idct32_slack, st_max);
    constexpr int lim = lim_net < in_max ? lim_net : in_max;
    if (FAST && !is_idtx) {
        const int l1 = svtgen::svt_l1<32>(x);
        if (__builtin_amdgcn_ballot_w64(l1 > lim) == 0) {
            svtgen::svt_idct32<12, false, false>(x, 0, 0);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = svtgen::svt_clamp(x[i], in_lo, in_hi);
    if (is_idtx) svtgen::svt_iidentity32<12>(x, 0, 0); else svtgen::svt_idct32<12>(x, st_lo, st_hi);
}

// ---------------------------------------------------------------------------
// inverse 32x32 + add (inv_txfm2d_add_c, EbTransforms.c:8180-8265): the mirror of
// the forward kernel.  Coefficients are read linearly (coalesced), a swizzled tile
// hands rows to lanes (row pass: clamp bd+8, idct32, round-shift 2), a second tile
// hands columns to lanes (column pass: clamp max(bd+6,16), idct32, round-shift 4),
// and a third tile puts the residual back in row order so that destination samples
// are read, updated and written 16 B per lane.
// dst block b at dst + (offsets ? offsets[b] : b*block_pitch), row stride dst_stride.
//
// This kernel is VALU-bound, not HBM-bound (tools/tune_inv32.py: with the global loads
// removed it runs in 73 % of the full time), so instruction selection follows the
// measured gfx950 issue costs (profiles/r01_valu_issue_cost_*.txt: v_add/v_sub/v_ashrrev/
// v_lshrrev/v_and/v_or/v_xor cost 1, nearly everything else - v_mul_i32_i24, v_mad_i32_i24,
// v_med3, v_lshlrev, v_bfe, SDWA, packed-16 ops - costs 1.7):
//   * every LDS address is  (per-lane base) ^ (compile-time constant)  + immediate offset,
//   * clamps are single v_med3_i32 with VGPR bounds, half_btf is two chained v_mad_i32_i24,
//   * the reconstruction works on packed 16-bit pairs: v_perm (pack two residuals),
//     v_pk_add_i16, v_sat_pk_u8_i16 (8-bit) or v_pk_max/min_i16 (16-bit samples).
// All LDS accesses are conflict-free under the lane-group rules of MI355X_MICROARCH.md
// (ds_read_b128: 4 x 16 lanes over 64 banks; ds_write_b128: 8 x 8 lanes over 32 banks).
// ---------------------------------------------------------------------------
// WAVES / VAR are tuning knobs (tools/tune_inv32.py): VAR bit0 = no destination prefetch,
// bit1 = skip the transforms (memory-only probe), bit2 = skip the global loads (compute-only probe), bit3 = always clamp
// (no clamp-free fast path, see idct32_pass; 5 waves / SIMD as before - the two-path form needs 117 VGPRs, so it runs at 4:
// measured 1.254 ms against 1.361 ms per 2^20 blocks, and 1.70 ms when squeezed into 96 VGPRs with 92 B of scratch).
// (A persistent, software-pipelined variant - next pair's coefficients fetched into registers during
// the transforms, 168 VGPRs, 3 waves/SIMD - measured 14 % SLOWER: the VALU needs the 5 waves/SIMD.)
template <typename PixT, int BD, int WAVES = F32_WAVES, int VAR = 0>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu((VAR & 8) ? 5 : 4))) void inv32_kernel(
    const int32_t* __restrict__ coeff, PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch,
    const uint32_t* __restrict__ dst_offsets, int is_idtx, uint32_t nblocks) {
    constexpr int bd = BD;
    constexpr bool PRE = !(VAR & 1);
    __shared__ __attribute__((aligned(16))) int32_t lds[WAVES * 2 * F32_TILE_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, li = lane & 31;
    char* tile = reinterpret_cast<char*>(lds + (wave * 2 + half) * F32_TILE_WORDS);
    constexpr int row_bits = bd == 8 ? 16 : (bd == 10 ? 18 : 20);     // av1_gen_inv_stage_range (:5404-5456)
    constexpr int col_bits = bd == 12 ? 18 : 16;
    constexpr int in_bits = bd + 8;                                    // clamp_buf(input, bd + 8) (:8226)
    constexpr int cin_bits = bd + 6 > 16 ? bd + 6 : 16;
    // clamp bounds in VGPRs; equal ranges share registers (bd = 8: all four are 16-bit)
    const int in_hi = svtgen::svt_vgpr((1 << (in_bits - 1)) - 1), in_lo = ~in_hi;
    const int row_hi = row_bits == in_bits ? in_hi : svtgen::svt_vgpr((1 << (row_bits - 1)) - 1), row_lo = ~row_hi;
    const int cin_hi = cin_bits == in_bits ? in_hi : svtgen::svt_vgpr((1 << (cin_bits - 1)) - 1), cin_lo = ~cin_hi;
    const int col_hi = col_bits == cin_bits ? cin_hi : svtgen::svt_vgpr((1 << (col_bits - 1)) - 1), col_lo = ~col_hi;
    constexpr int PPL = 16 / (int)sizeof(PixT);          // pixels per lane per step
    constexpr int STEPS = 1024 / (32 * PPL);
    constexpr int SPL = PPL / 4;                         // 16-B residual slots per lane per step (4 or 2)
    constexpr int maxpix = (1 << bd) - 1;
    // per-lane LDS offsets (see the access sites)
    const int a_w = (li >> 3) * 128 + (((li & 7) ^ (li >> 4)) << 4);
    const int a_r = li * 128 + (((li >> 1) & 7) << 4);
    const int b_w = li * 128 + ((li & 7) << 4);
    const int b_r = ((li >> 2) << 4) + (li & 3) * 4;
    const int c_w = b_r;

    const uint32_t blk = (blockIdx.x * WAVES + wave) * 2 + half;
    const bool valid = blk < nblocks;
    const size_t dbase = valid ? (dst_offsets ? (size_t)dst_offsets[blk] : (size_t)blk * dst_block_pitch) : 0;
    const bool dst_aligned = (((reinterpret_cast<uintptr_t>(dst) + dbase * sizeof(PixT)) & 15) == 0) && ((dst_stride * (int)sizeof(PixT)) & 15) == 0;
    // destination samples are fetched up front (their latency hides under the two transform passes)
    uint4 dcur[STEPS];
    const bool dcur_ok = PRE && valid && dst_aligned;
    if (dcur_ok) {
#pragma unroll
        for (int k = 0; k < STEPS; k++) {
            const int p = (k * 32 + li) * PPL;
            dcur[k] = (VAR & 4) ? make_uint4(k, li, k, li)
                                : *reinterpret_cast<const uint4*>(dst + dbase + (size_t)(p >> 5) * dst_stride + (p & 31));
        }
    }
    {
        int x[32];
        // ---- coefficients -> tile A.  Chunk k of lane li is row 4k + li/8, 16-B slot li%8, stored at
        // slot (li%8) ^ ((row>>1)&7):  address = a_w ^ ((2k & 7) << 4)  + k*512
        const int4* c4 = reinterpret_cast<const int4*>(coeff + (size_t)blk * 1024);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int4 v = (valid && !(VAR & 4)) ? c4[k * 32 + li] : make_int4(li, k, li, k);
            *reinterpret_cast<int4*>(tile + (a_w ^ (((2 * k) & 7) << 4)) + k * 512) = v;
        }
        wave_lds_fence();
        // row li, slot s sits at s ^ ((li>>1)&7)
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int4 v = *reinterpret_cast<const int4*>(tile + (a_r ^ (s << 4)));
            x[s * 4 + 0] = v.x; x[s * 4 + 1] = v.y; x[s * 4 + 2] = v.z; x[s * 4 + 3] = v.w;
        }
        // ---- row pass (input clamp inside) -------------------------------------------------
        if (VAR & 2) {}
        else idct32_pass<in_bits, row_bits, !(VAR & 8)>(x, is_idtx, in_lo, in_hi, row_lo, row_hi);
        wave_lds_fence();
        // ---- tile B: row li written with slot swizzle li&7, columns read back ----------------
#pragma unroll
        for (int s = 0; s < 8; s++)
            *reinterpret_cast<int4*>(tile + (b_w ^ (s << 4))) =
                make_int4((x[s * 4 + 0] + 2) >> 2, (x[s * 4 + 1] + 2) >> 2, (x[s * 4 + 2] + 2) >> 2, (x[s * 4 + 3] + 2) >> 2);  // shift[0] = -2
        wave_lds_fence();
        // element (r, li): slot (li>>2) ^ (r&7), word li&3
#pragma unroll
        for (int r = 0; r < 32; r++) {
            x[r] = *reinterpret_cast<const int*>(tile + (b_r ^ ((r & 7) << 4)) + r * 128);
        }
        // ---- column pass (input clamp inside) ----------------------------------------------
        if (VAR & 2) {}
        else idct32_pass<cin_bits, col_bits, !(VAR & 8)>(x, is_idtx, cin_lo, cin_hi, col_lo, col_hi);
        wave_lds_fence();
        // ---- tile C: residual words in row order.  Word (r, c) lives in 16-B slot sigma = r*8 + c/4;
        // the reader takes SPL consecutive slots per lane, so slots are swizzled by
        // sigma ^ ((sigma >> 4) & (SPL-1)) = (c/4) ^ ((r>>1) & (SPL-1)) to keep its b128 reads conflict-free.
#pragma unroll
        for (int r = 0; r < 32; r++)
            // shift[1] = -4.  The residual is added to the sample on 16-bit lanes below (the reference adds in int32): exact for
            // BD <= 10 (column outputs <= 16 bits, 12-14 after the shift); bd 12 takes the general kernel
            *reinterpret_cast<int*>(tile + (c_w ^ (((r >> 1) & (SPL - 1)) << 4)) + r * 128) = (x[r] + 8) >> 4;
        wave_lds_fence();
        if (valid) {
#pragma unroll
            for (int k = 0; k < STEPS; k++) {
                const int L = k * 32 + li;                        // 16-B store unit: pixels L*PPL .. +PPL-1
                const int p = L * PPL;
                const int row = p >> 5, col = p & 31;
                PixT* d = dst + dbase + (size_t)row * dst_stride + col;
                int rv[PPL];
                const int g = ((L * SPL) >> 4) & (SPL - 1);      // slot swizzle of tile C: sigma ^ ((sigma >> 4) & (SPL-1))
                const int c_r = L * (SPL * 16);
#pragma unroll
                for (int j = 0; j < SPL; j++) {
                    const int4 t = *reinterpret_cast<const int4*>(tile + c_r + ((j ^ g) << 4));
                    rv[4 * j] = t.x; rv[4 * j + 1] = t.y; rv[4 * j + 2] = t.z; rv[4 * j + 3] = t.w;
                }
                if (dst_aligned) {
                    const uint4 pv = dcur_ok ? dcur[k] : ((VAR & 4) ? make_uint4(k, li, k, li) : *reinterpret_cast<const uint4*>(d));
                    const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
                    uint32_t ow[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (sizeof(PixT) == 1) {
                            // pixels (0,2) / (1,3) of the dword as 16-bit lanes; same pairing for the residuals
                            const uint32_t pe = pw[q] & 0x00ff00ffu, po = (pw[q] >> 8) & 0x00ff00ffu;
                            const uint32_t re = __builtin_amdgcn_perm((uint32_t)rv[4 * q + 2], (uint32_t)rv[4 * q + 0], 0x05040100u);
                            const uint32_t ro = __builtin_amdgcn_perm((uint32_t)rv[4 * q + 3], (uint32_t)rv[4 * q + 1], 0x05040100u);
                            const uint32_t ue = sat_pk_u8_i16(pk_add_i16(pe, re)), uo = sat_pk_u8_i16(pk_add_i16(po, ro));
                            ow[q] = __builtin_amdgcn_perm(uo, ue, 0x05010400u);
                        } else {
                            const uint32_t rr = __builtin_amdgcn_perm((uint32_t)rv[2 * q + 1], (uint32_t)rv[2 * q + 0], 0x05040100u);
                            ow[q] = pk_clamp_i16(pk_add_i16(pw[q], rr), maxpix);
                        }
                    }
                    *reinterpret_cast<uint4*>(d) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < PPL; j++) d[j] = (PixT)min(max((int)d[j] + rv[j], 0), maxpix);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// enc32_kernel — the encode-pass chain of one 32x32 8-bit transform block in ONE kernel
// (reference call sequence in Av1EncodeLoop, EbCodingLoop.c:545-950):
//   ResidualKernel -> av1_estimate_transform (32x32 DCT_DCT / IDTX) -> av1_quantize_inv_quantize
//   (aom_highbd_quantize_b_32x32: qcoeff, dqcoeff, eob) -> av1_inv_transform_recon8bit (pred + inverse)
// plus the 32x32 SAD(src, pred).  The forward coefficients, the dequantised coefficients and the
// residual never leave the CU: HBM traffic per block is 2 x 1024 B in + 4096 B (qcoeff) + 1024 B (recon)
// + 6 B out = 7 174 B, against 14 342 + 6 144 B for the two separate kernels.  coeff / dqcoeff are
// written too when the caller passes buffers for them (KEEP).
// The body is fwd32_kernel<IN_U8, QUANT> up to the quantiser, whose dequantised int4 chunks are exactly
// the linear 16-B chunks inv32_kernel loads, and the prediction samples already sit in registers in
// the 16-B-per-lane layout of its reconstruction stage.
// ---------------------------------------------------------------------------
// PixT / BD: uint8_t / 8 or uint16_t / 10 (BASELINE configs[4]); the 16-bit variant differs only in how the
// residual is formed (v_pk_sub_i16 on the loaded words), in the inverse's clamp ranges and in the final clip.
// (body / kernel split: the body takes its workgroup index and LDS from the caller, so that enc_frame_kernel - one launch for
// every group of a picture, kernel_frame.h - can run it for the workgroups of a 32x32 group)
constexpr int ENC32_LDS_BYTES = F32_WAVES * 2 * F32_TILE_WORDS * 4;
template <typename PixT, int BD, bool KEEP, bool WITH_SAD>
__device__ __forceinline__ void enc32_body(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, const QParams& qp,
    int is_idtx, uint32_t nblocks, const uint32_t* __restrict__ xy, uint32_t src_stride,
    uint32_t pred_stride, uint32_t recon_stride, uint32_t bid, int32_t* lds) {
    // xy != NULL: blocks addressed on picture planes (origin (x, y) = (xy[b] & 0xffff, xy[b] >> 16), row strides
    // in samples; recon may be the prediction plane itself); NULL: dense 32x32 blocks.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, li = lane & 31;
    char* tile = reinterpret_cast<char*>(lds + (wave * 2 + half) * F32_TILE_WORDS);
    const uint32_t blk = (bid * F32_WAVES + wave) * 2 + half;
    const bool valid = blk < nblocks;
    const size_t pix_off = (size_t)blk * 1024;

    constexpr bool HBD = sizeof(PixT) == 2;
    constexpr int NPK = HBD ? 4 : 2;                          // 16-B prediction chunks per lane (= reconstruction steps)
    uint4 pk[NPK];                                            // chunk k of the lane: pixels (k*32 + li) * (16/sizeof(PixT)) ...
    size_t rbase = pix_off;                                   // recon sample offset of the block
    uint32_t rstr = 32, sstr = 32, pstr = 32;
    size_t sbase = pix_off, pbase = pix_off;
    if (valid && xy) {
        const uint32_t o = xy[blk];
        const size_t by = o >> 16, bx = o & 0xffffu;
        sstr = src_stride; pstr = pred_stride; rstr = recon_stride;
        sbase = by * sstr + bx; pbase = by * pstr + bx; rbase = by * rstr + bx;
    }
    unsigned sad_acc = 0;
    int x[32];
    if constexpr (!HBD) {
        // ---- 2 x 1 KB, 16 B per lane: rows li/2 and 16 + li/2, columns (li&1)*16 .. +15; SAD on the raw bytes
        uint4 s0 = {0, 0, 0, 0}, s1 = s0;
        pk[0] = s0; pk[1] = s0;
        if (valid) {
            const PixT* sp = src + sbase + (size_t)(li >> 1) * sstr + (li & 1) * 16;
            const PixT* pp = pred + pbase + (size_t)(li >> 1) * pstr + (li & 1) * 16;
            __builtin_memcpy(&s0, sp, 16); __builtin_memcpy(&s1, sp + (size_t)16 * sstr, 16);
            __builtin_memcpy(&pk[0], pp, 16); __builtin_memcpy(&pk[1], pp + (size_t)16 * pstr, 16);
        }
        const uint4 p0 = pk[0], p1 = pk[1];
        if (WITH_SAD) {
            sad_acc = __builtin_amdgcn_sad_u8(s0.x, p0.x, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s0.y, p0.y, sad_acc);
            sad_acc = __builtin_amdgcn_sad_u8(s0.z, p0.z, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s0.w, p0.w, sad_acc);
            sad_acc = __builtin_amdgcn_sad_u8(s1.x, p1.x, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s1.y, p1.y, sad_acc);
            sad_acc = __builtin_amdgcn_sad_u8(s1.z, p1.z, sad_acc); sad_acc = __builtin_amdgcn_sad_u8(s1.w, p1.w, sad_acc);
        }
        // residual as packed int16 pairs -> LDS -> columns (layout: see fwd32_kernel)
        const uint32_t sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const uint32_t pw[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
        for (int kp = 0; kp < 4; kp++) {
            uint32_t r[4];
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const uint32_t a = sw[kp * 2 + h2], b = pw[kp * 2 + h2];
                const uint32_t a01 = __builtin_amdgcn_perm(0u, a, 0x0c010c00u), a23 = __builtin_amdgcn_perm(0u, a, 0x0c030c02u);
                const uint32_t b01 = __builtin_amdgcn_perm(0u, b, 0x0c010c00u), b23 = __builtin_amdgcn_perm(0u, b, 0x0c030c02u);
                r[h2 * 2 + 0] = pk_shl2_i16(pk_sub_i16(a01, b01));      // residual x 4 (fwd_shift_32x32[0] = 2), packed
                r[h2 * 2 + 1] = pk_shl2_i16(pk_sub_i16(a23, b23));
            }
            *reinterpret_cast<uint4*>(tile + kp * 544 + li * 16) = make_uint4(r[0], r[1], r[2], r[3]);
        }
        wave_lds_fence();
        const char* colbase = tile + ((li >> 3) & 1) * 544 + (li >> 4) * 16 + (li & 7) * 2;
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const short v = *reinterpret_cast<const short*>(colbase + (r >> 4) * 1088 + (r & 15) * 32);
            x[r] = (int)v;                                       // shift[0] = 2 already applied
        }
    } else {
        // ---- 2 x 2 KB: chunk k (0..3) of lane li = row k*8 + li/4, columns (li&3)*8 .. +7 (fwd32_kernel, IN = 2)
        uint4 sv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sv[k] = make_uint4(0, 0, 0, 0); pk[k] = sv[k];
            if (valid) {
                __builtin_memcpy(&sv[k], src + sbase + (size_t)(k * 8 + (li >> 2)) * sstr + (li & 3) * 8, 16);
                __builtin_memcpy(&pk[k], pred + pbase + (size_t)(k * 8 + (li >> 2)) * pstr + (li & 3) * 8, 16);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t a[4] = {sv[k].x, sv[k].y, sv[k].z, sv[k].w}, b[4] = {pk[k].x, pk[k].y, pk[k].z, pk[k].w};
            uint32_t d[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                d[j] = pk_shl2_i16(pk_sub_i16(a[j], b[j]));      // |s - p| * 4 <= 4092 fits int16
                if (WITH_SAD) sad_acc = __builtin_amdgcn_sad_u16(a[j], b[j], sad_acc);
            }
            *reinterpret_cast<uint4*>(tile + k * 512 + li * 16) = make_uint4(d[0], d[1], d[2], d[3]);
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const short v = *reinterpret_cast<const short*>(tile + (r >> 3) * 512 + (r & 7) * 64 + li * 2);
            x[r] = (int)v;
        }
    }
    // ---- forward: column pass, transpose, row pass, re-order to linear ------------------------
    if (is_idtx) svtgen::svt_fidentity32<F32_COS_BIT>(x); else svtgen::svt_fdct32<F32_COS_BIT>(x);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = (x[r] + 8) >> 4;      // shift[1] = -4
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 32; r++)
        *reinterpret_cast<int*>(tile + tile_slot(r, li >> 2, (r >> 1) & 7) + (li & 3) * 4) = x[r];
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int4 v = *reinterpret_cast<const int4*>(tile + tile_slot(li, s, (li >> 1) & 7));
        x[s * 4 + 0] = v.x; x[s * 4 + 1] = v.y; x[s * 4 + 2] = v.z; x[s * 4 + 3] = v.w;
    }
    if (is_idtx) svtgen::svt_fidentity32<F32_COS_BIT>(x); else svtgen::svt_fdct32<F32_COS_BIT>(x);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 8; s++)
        *reinterpret_cast<int4*>(tile + tile_slot(li, s, li & 7)) =
            make_int4(x[s * 4 + 0], x[s * 4 + 1], x[s * 4 + 2], x[s * 4 + 3]);
    wave_lds_fence();
    // ---- quantise the lane's 8 linear 16-B chunks; dequantised chunks stay in registers ---------
    int4 dqv[8];
    {
        int4 cv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int row = 4 * k + (li >> 3);
            cv[k] = *reinterpret_cast<const int4*>(tile + tile_slot(row, li & 7, row & 7));
        }
        wave_lds_fence();                                        // the tile is free for the inverse now
        int4* co4 = reinterpret_cast<int4*>(coeff + pix_off);
        int4* qc4 = reinterpret_cast<int4*>(qcoeff + pix_off);
        int4* dq4 = reinterpret_cast<int4*>(dqcoeff + pix_off);
        int eob_acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint2 isc = *reinterpret_cast<const uint2*>(iscan + (k * 32 + li) * 4);
            int4 q;
            quant_one<2>(cv[k].x, (k == 0 && li == 0) ? 0 : 1, qp, q.x, dqv[k].x);
            quant_one<2>(cv[k].y, 1, qp, q.y, dqv[k].y);
            quant_one<2>(cv[k].z, 1, qp, q.z, dqv[k].z);
            quant_one<2>(cv[k].w, 1, qp, q.w, dqv[k].w);
            const int e0 = q.x ? (int)(isc.x & 0xffffu) + 1 : 0, e1 = q.y ? (int)(isc.x >> 16) + 1 : 0;
            const int e2 = q.z ? (int)(isc.y & 0xffffu) + 1 : 0, e3 = q.w ? (int)(isc.y >> 16) + 1 : 0;
            eob_acc = max(eob_acc, max(max(e0, e1), max(e2, e3)));
            if (valid) {
                qc4[k * 32 + li] = q;
                if (KEEP) { co4[k * 32 + li] = cv[k]; dq4[k * 32 + li] = dqv[k]; }
            }
        }
        eob_acc = half_wave_max(eob_acc);
        if (WITH_SAD) sad_acc = half_wave_sum(sad_acc);
        if (valid && li == 0) {
            eob[blk] = (uint16_t)eob_acc;
            if (WITH_SAD) sad[blk] = sad_acc;
        }
    }
    // ---- inverse (inv32_kernel<uint8_t, 8> from its tile-A stage on) ---------------------------
    // clamp ranges (av1_gen_inv_stage_range, :5404-5456): input bd+8, rows 16/18/20, column input max(bd+6,16), columns 16 (18 at bd 12)
    constexpr int in_bits = BD + 8, row_bits = BD == 8 ? 16 : (BD == 10 ? 18 : 20);
    constexpr int cin_bits = BD + 6 > 16 ? BD + 6 : 16, col_bits = BD == 12 ? 18 : 16;
    const int c_hi = svtgen::svt_vgpr((1 << (cin_bits - 1)) - 1), c_lo = ~c_hi;
    const int i_hi = in_bits == cin_bits ? c_hi : svtgen::svt_vgpr((1 << (in_bits - 1)) - 1), i_lo = ~i_hi;
    const int r_hi = row_bits == in_bits ? i_hi : svtgen::svt_vgpr((1 << (row_bits - 1)) - 1), r_lo = ~r_hi;
    const int o_hi = col_bits == cin_bits ? c_hi : svtgen::svt_vgpr((1 << (col_bits - 1)) - 1), o_lo = ~o_hi;
    const int a_w = (li >> 3) * 128 + (((li & 7) ^ (li >> 4)) << 4);
    const int a_r = li * 128 + (((li >> 1) & 7) << 4);
    const int b_w = li * 128 + ((li & 7) << 4);
    const int b_r = ((li >> 2) << 4) + (li & 3) * 4;
#pragma unroll
    for (int k = 0; k < 8; k++) *reinterpret_cast<int4*>(tile + (a_w ^ (((2 * k) & 7) << 4)) + k * 512) = dqv[k];
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int4 v = *reinterpret_cast<const int4*>(tile + (a_r ^ (s << 4)));
        x[s * 4 + 0] = v.x; x[s * 4 + 1] = v.y; x[s * 4 + 2] = v.z; x[s * 4 + 3] = v.w;
    }
    idct32_pass<in_bits, row_bits, true>(x, is_idtx, i_lo, i_hi, r_lo, r_hi);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 8; s++)
        *reinterpret_cast<int4*>(tile + (b_w ^ (s << 4))) =
            make_int4((x[s * 4 + 0] + 2) >> 2, (x[s * 4 + 1] + 2) >> 2, (x[s * 4 + 2] + 2) >> 2, (x[s * 4 + 3] + 2) >> 2);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 32; r++) {
        x[r] = *reinterpret_cast<const int*>(tile + (b_r ^ ((r & 7) << 4)) + r * 128);
    }
    idct32_pass<cin_bits, col_bits, true>(x, is_idtx, c_lo, c_hi, o_lo, o_hi);
    wave_lds_fence();
    constexpr int PPL = 16 / (int)sizeof(PixT), SPL = PPL / 4, maxpix = (1 << BD) - 1;
#pragma unroll
    for (int r = 0; r < 32; r++)
        *reinterpret_cast<int*>(tile + (b_r ^ (((r >> 1) & (SPL - 1)) << 4)) + r * 128) = (x[r] + 8) >> 4;
    wave_lds_fence();
    if (valid) {
#pragma unroll
        for (int k = 0; k < NPK; k++) {
            const int L = k * 32 + li;                            // 16-B unit: pixels L*PPL .. +PPL-1 = the lane's prediction chunk k
            const int g = ((L * SPL) >> 4) & (SPL - 1);
            int rv[PPL];
#pragma unroll
            for (int j = 0; j < SPL; j++) {
                const int4 t = *reinterpret_cast<const int4*>(tile + L * (SPL * 16) + ((j ^ g) << 4));
                rv[4 * j] = t.x; rv[4 * j + 1] = t.y; rv[4 * j + 2] = t.z; rv[4 * j + 3] = t.w;
            }
            const uint32_t pw[4] = {pk[k].x, pk[k].y, pk[k].z, pk[k].w};
            uint32_t ow[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if constexpr (!HBD) {
                    const uint32_t pe = pw[q] & 0x00ff00ffu, po = (pw[q] >> 8) & 0x00ff00ffu;
                    const uint32_t re = __builtin_amdgcn_perm((uint32_t)rv[4 * q + 2], (uint32_t)rv[4 * q + 0], 0x05040100u);
                    const uint32_t ro = __builtin_amdgcn_perm((uint32_t)rv[4 * q + 3], (uint32_t)rv[4 * q + 1], 0x05040100u);
                    const uint32_t ue = sat_pk_u8_i16(pk_add_i16(pe, re)), uo = sat_pk_u8_i16(pk_add_i16(po, ro));
                    ow[q] = __builtin_amdgcn_perm(uo, ue, 0x05010400u);
                } else {
                    const uint32_t rr = __builtin_amdgcn_perm((uint32_t)rv[2 * q + 1], (uint32_t)rv[2 * q + 0], 0x05040100u);
                    ow[q] = pk_clamp_i16(pk_add_i16(pw[q], rr), maxpix);
                }
            }
            const uint4 ov = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            const int p = L * PPL;                                // row p/32, column p%32
            if (xy) __builtin_memcpy(recon + rbase + (size_t)(p >> 5) * rstr + (p & 31), &ov, 16);
            else reinterpret_cast<uint4*>(recon + pix_off)[L] = ov;
        }
    }
}

template <typename PixT, int BD, bool KEEP, bool WITH_SAD>
__global__ __launch_bounds__(F32_WAVES * 64) __attribute__((amdgpu_waves_per_eu(3))) void enc32_kernel(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, QParams qp,
    int is_idtx, uint32_t nblocks, const uint32_t* __restrict__ xy = nullptr, uint32_t src_stride = 32,
    uint32_t pred_stride = 32, uint32_t recon_stride = 32) {
    __shared__ __attribute__((aligned(16))) int32_t lds[F32_WAVES * 2 * F32_TILE_WORDS];
    enc32_body<PixT, BD, KEEP, WITH_SAD>(src, pred, recon, coeff, qcoeff, dqcoeff, eob, sad, iscan, qp, is_idtx, nblocks, xy, src_stride,
                                         pred_stride, recon_stride, blockIdx.x, lds);
}

}  // namespace svtdev
