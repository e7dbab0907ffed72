// kernel_txfm_staged.h — forward transforms / fused chain for DENSE block batches of
// every size: same math as kernel_txfm.h, but all HBM traffic is 16 B per lane and
// fully coalesced.  A wave's 64/max(W,H) blocks are contiguous in memory, so
//   in : linear 16-B chunks -> LDS staging (padded per block: conflict-free column reads)
//   out: rows -> XOR-swizzled LDS tile -> linear 16-B chunks (quantised in that order)
// MODE 0: int16 residual -> coeff (av1_fwd_txfm2d_WxH, full W*H output)
// MODE 1: u8 src + pred -> coeff, qcoeff, dqcoeff (packed min(W,32)*min(H,32)), eob, sad,
//         three_quad_energy  (the headline chain for sizes other than 32x32)
#pragma once
#include "kernel_txfm.h"

namespace svtdev {

constexpr int cmax(int a, int b) { return a > b ? a : b; }
template <int PELS> struct StagedWaves { static constexpr int N = PELS >= 4096 ? 2 : 4; };

template <int W, int H>
struct StagedGeom {
    using G = TxGeom<W, H>;
    static constexpr int WAVES = StagedWaves<W * H>::N;
    static constexpr int KW = W > 32 ? 32 : W, KH = H > 32 ? 32 : H, NC = KW * KH;
    // out tile: row r, 16-B slot s -> slot s ^ ((r / RDIV) & SMASK)   (conflict-free b128 row writes)
    static constexpr int NS = W / 4;
    static constexpr int RDIV = (8 / NS) > 1 ? (8 / NS) : 1;
    static constexpr int SMASK = (NS < 8 ? NS : 8) - 1;
    __device__ static __forceinline__ int out_addr(int sub, int r, int s) {
        return sub * (W * H * 4) + r * (W * 4) + ((s ^ ((r / RDIV) & SMASK)) << 4);
    }
};

// LDS layouts shared by the inverse / encode kernels below, chosen against the PMC (profiles/r02_pmc_c_enc8.json: 26 % of the LDS
// cycles of the 8x8 encode kernel were bank conflicts, 14 % at 16x16):
//  * coefficient tile (linear 16-byte stores -> one-row-per-lane b128 reads): rows of KW * 4 bytes whose 16-byte slots are
//    XOR-swizzled by the row (the out tile's rule): stores AND reads are conflict-free.  The first version padded the row pitch
//    by one slot instead, which kept the reads clean and made every linear store a 2-way conflict;
//  * residual tile (column pass -> reconstruction): int16, W * H + ResPad<W> per block: with the dense pitch the column stores
//    of all the wave's blocks fell on the same W / 2 banks (8-way at 8x8, 4-way at 16x16).
template <int KW, int KH>
__device__ __forceinline__ int coef_tile_addr(int b, int r, int sl) {
    constexpr int NS = KW / 4, RDIV = (8 / NS) > 1 ? (8 / NS) : 1, SMASK = (NS < 8 ? NS : 8) - 1;
    return (b * KH + r) * (KW * 4) + ((sl ^ ((r / RDIV) & SMASK)) << 4);
}
template <int W> struct ResPad { static constexpr int N = W > 8 ? W : 8; };

// PixT: sample type of src / pred in MODE 1 (uint8_t, or uint16_t for 10-bit).  xy != NULL: the blocks are
// addressed on picture planes (origin (x, y) = (xy[b] & 0xffff, xy[b] >> 16), row strides in samples) and are
// fetched row segment by row segment into the same linear staging image; NULL: dense batches.
template <int W, int H, int MODE, typename PixT = uint8_t>
__global__ __launch_bounds__(StagedWaves<W * H>::N * 64) void fwd_staged_kernel(
    const void* __restrict__ in0, const void* __restrict__ pred, int32_t* __restrict__ coeff,
    int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob,
    uint32_t* __restrict__ sad, unsigned long long* __restrict__ energy, const int16_t* __restrict__ iscan,
    QParams qp, int tx_type, uint32_t nblocks, const uint32_t* __restrict__ xy = nullptr, uint32_t src_stride = 0,
    uint32_t pred_stride = 0) {
    using S = StagedGeom<W, H>;
    using G = TxGeom<W, H>;
    constexpr bool FUSED = MODE == 1;
    constexpr int ES = FUSED ? (int)sizeof(PixT) : 2;
    constexpr int BB = W * H * ES;                       // input bytes per block and array
    constexpr int PADI = (W * ES >= 32) ? 32 : 16;       // staging pad per block
    constexpr int IN_ONE = G::BPW * (BB + PADI);
    constexpr int IN_BYTES = IN_ONE * (FUSED ? 2 : 1);
    constexpr int TILE_BYTES = G::BPW * G::TILE * 4;
    constexpr int OUT_BYTES = G::BPW * W * H * 4;
    constexpr int WAVE_LDS = (cmax(cmax(IN_BYTES, TILE_BYTES), OUT_BYTES) + 15) & ~15;
    __shared__ __attribute__((aligned(16))) char lds[S::WAVES * WAVE_LDS];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wl = lds + wave * WAVE_LDS;
    const uint32_t first = (blockIdx.x * S::WAVES + wave) * G::BPW;
    if (first >= nblocks) return;                         // wave-uniform
    const int sub = lane / G::LPB, l = lane % G::LPB;
    const uint32_t blk = first + sub;
    const bool valid = blk < nblocks;
    const int vk = kVKind[tx_type], hk = kHKind[tx_type];
    const bool ud = vk == K1D_FLIPADST, lr = hk == K1D_FLIPADST;
    constexpr int CBC = fwd_cos_col(W, H), CBR = fwd_cos_row(W, H);
    constexpr int S0 = fwd_shift(W, H, 0), S1 = fwd_shift(W, H, 1), S2 = fwd_shift(W, H, 2);

    // ---- stage the wave's input -------------------------------------------------------------
    if (FUSED && xy) {
        // planes: chunks of CS bytes that never cross a block row; every offset-table and sample load of a
        // lane is issued before the first LDS write
        constexpr int ROWB = W * ES, CS = ROWB >= 16 ? 16 : ROWB;
        constexpr int CPR = ROWB / CS, CPBP = BB / CS, NCHP = G::BPW * CPBP, NIT = (NCHP + 63) / 64;
        uint32_t org[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, b = q / CPBP;
            org[it] = (q < NCHP && first + b < nblocks) ? xy[first + b] : 0xffffffffu;
        }
        uint4 v0[NIT], v1[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, w = q % CPBP;
            const int row = w / CPR, cb = (w % CPR) * CS;
            v0[it] = make_uint4(0, 0, 0, 0); v1[it] = v0[it];
            if (org[it] != 0xffffffffu) {
                const size_t y = (org[it] >> 16) + row, x = org[it] & 0xffffu;
                __builtin_memcpy(&v0[it], static_cast<const char*>(in0) + (y * src_stride + x) * ES + cb, CS);
                __builtin_memcpy(&v1[it], static_cast<const char*>(pred) + (y * pred_stride + x) * ES + cb, CS);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, b = q / CPBP;
            if (NCHP % 64 == 0 || q < NCHP) {
                __builtin_memcpy(wl + q * CS + b * PADI, &v0[it], CS);
                __builtin_memcpy(wl + IN_ONE + q * CS + b * PADI, &v1[it], CS);
            }
        }
    } else {
        constexpr int NCH = G::BPW * BB / 16;
        const char* g0 = static_cast<const char*>(in0) + (size_t)first * BB;
        const char* g1 = FUSED ? reinterpret_cast<const char*>(pred) + (size_t)first * BB : nullptr;
#pragma unroll
        for (int q0 = 0; q0 < NCH; q0 += 64) {
            const int q = q0 + lane;
            if (NCH % 64 == 0 || q < NCH) {
                const int b = (q * 16) / BB;
                const bool ok = first + b < nblocks;
                // (predicated loads into zeroed registers: `ok ? *p : zero` makes the compiler park the zero in scratch
                // and select the POINTER - a flat load through private memory)
                uint4 va = make_uint4(0, 0, 0, 0), vb = va;
                if (ok) { va = *reinterpret_cast<const uint4*>(g0 + (size_t)q * 16); if (FUSED) vb = *reinterpret_cast<const uint4*>(g1 + (size_t)q * 16); }
                *reinterpret_cast<uint4*>(wl + q * 16 + b * PADI) = va;
                if (FUSED) *reinterpret_cast<uint4*>(wl + IN_ONE + q * 16 + b * PADI) = vb;
            }
        }
    }
    wave_lds_fence();
    // ---- column pass ----------------------------------------------------------------------
    unsigned sad_acc = 0;
    int x[H];
    if (l < W) {
        const char* bs = wl + sub * (BB + PADI);
#pragma unroll
        for (int r = 0; r < H; r++) {
            const int idx = (ud ? H - 1 - r : r) * W + l;
            int d;
            if (FUSED) {
                d = (int)*reinterpret_cast<const PixT*>(bs + idx * ES) - (int)*reinterpret_cast<const PixT*>(bs + IN_ONE + idx * ES);
                sad_acc += (unsigned)(d < 0 ? -d : d);
            } else {
                d = *reinterpret_cast<const short*>(bs + idx * 2);
            }
            x[r] = round_shift_c<-S0>(d);
        }
        fwd1d<H, CBC>(vk, x);
    }
    wave_lds_fence();                                     // staging is dead: the tile may overwrite it
    int32_t* tile = reinterpret_cast<int32_t*>(wl) + sub * G::TILE;
    if (l < W) {
        const int cdst = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++) tile[r * G::PITCH + cdst] = round_shift_c<-S1>(x[r]);
    }
    wave_lds_fence();
    // ---- row pass -------------------------------------------------------------------------
    unsigned long long en = 0;
    int y[W];
    if (l < H) {
#pragma unroll
        for (int c = 0; c < W; c++) y[c] = tile[l * G::PITCH + c];
        fwd1d<W, CBR>(hk, y);
#pragma unroll
        for (int c = 0; c < W; c++) {
            int t = round_shift_c<-S2>(y[c]);
            if (G::RECT2) t = mul_q12(t, 5793);
            y[c] = t;
        }
        if (FUSED && (W > 32 || H > 32)) {
#pragma unroll
            for (int c = 0; c < W; c++)
                if (l >= S::KH || c >= S::KW) { const long long v = y[c]; en += (unsigned long long)(v * v); }
        }
    }
    wave_lds_fence();                                     // tile is dead: the out tile may overwrite it
    if (l < H) {
#pragma unroll
        for (int s = 0; s < W / 4; s++)
            *reinterpret_cast<int4*>(wl + S::out_addr(sub, l, s)) = make_int4(y[4 * s], y[4 * s + 1], y[4 * s + 2], y[4 * s + 3]);
    }
    wave_lds_fence();
    if (FUSED) {
        sad_acc = group_sum<G::LPB>(sad_acc);
        if (W > 32 || H > 32) en = group_sum64<G::LPB>(en);
        if (valid && l == 0) {
            if (sad) sad[blk] = sad_acc;
            if (energy) energy[blk] = en;
        }
    }
    // ---- linear phase: coalesced stores (and quantisation) ----------------------------------
    if (!FUSED) {
        constexpr int CPB = W * H / 4;                    // chunks per block
        constexpr int NOUT = G::BPW * CPB;
        int4* o4 = reinterpret_cast<int4*>(coeff + (size_t)first * (W * H));
#pragma unroll
        for (int q0 = 0; q0 < NOUT; q0 += 64) {
            const int q = q0 + lane;
            if (NOUT % 64 == 0 || q < NOUT) {
                const int b = q / CPB, w4 = q % CPB;
                const int4 v = *reinterpret_cast<const int4*>(wl + S::out_addr(b, w4 / (W / 4), w4 % (W / 4)));
                if (first + b < nblocks) o4[q] = v;
            }
        }
    } else {
        constexpr int CPB = S::NC / 4;
        constexpr int NOUT = G::BPW * CPB;
        int eob_acc = 0;
#pragma unroll
        for (int q0 = 0; q0 < NOUT; q0 += 64) {
            const int q = q0 + lane;
            const bool act = (NOUT % 64 == 0) || q < NOUT;
            const int b = act ? q / CPB : 0, w4 = act ? q % CPB : 0;
            const bool ok = act && (first + b < nblocks);
            const int4 c = *reinterpret_cast<const int4*>(wl + S::out_addr(b, w4 / (S::KW / 4), w4 % (S::KW / 4)));
            int4 qv, dv;
            // QMODE 2: the host only takes this kernel when quant_shift is a power of two
            quant_one<2>(c.x, w4 == 0 ? 0 : 1, qp, qv.x, dv.x);
            quant_one<2>(c.y, 1, qp, qv.y, dv.y);
            quant_one<2>(c.z, 1, qp, qv.z, dv.z);
            quant_one<2>(c.w, 1, qp, qv.w, dv.w);
            const uint2 is = *reinterpret_cast<const uint2*>(iscan + w4 * 4);
            int e = max(max(qv.x ? (int)(is.x & 0xffffu) + 1 : 0, qv.y ? (int)(is.x >> 16) + 1 : 0),
                        max(qv.z ? (int)(is.y & 0xffffu) + 1 : 0, qv.w ? (int)(is.y >> 16) + 1 : 0));
            if (!act) e = 0;
            if (ok) {
                const size_t o = (size_t)(first + b) * S::NC + (size_t)w4 * 4;
                *reinterpret_cast<int4*>(coeff + o) = c;
                *reinterpret_cast<int4*>(qcoeff + o) = qv;
                *reinterpret_cast<int4*>(dqcoeff + o) = dv;
            }
            if constexpr (CPB >= 64) {                    // one block spans CPB/64 iterations of the whole wave
                eob_acc = max(eob_acc, e);
                if (((q0 / 64) + 1) % (CPB / 64) == 0) {
                    const int m = group_max<64>(eob_acc);
                    if (lane == 0 && ok) eob[first + b] = (uint16_t)m;
                    eob_acc = 0;
                }
            } else {                                      // 64/CPB blocks per iteration
                const int m = group_max<(CPB < 64 ? CPB : 64)>(e);
                if (ok && w4 == 0) eob[first + b] = (uint16_t)m;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// inverse + add for DENSE batches (coefficients packed KW*KH per block, destination
// blocks W*H samples back to back): linear coefficient chunks -> padded LDS rows ->
// row pass -> transpose tile -> column pass -> int16 residual tile in row order ->
// destination read / add / clip / write in linear 16-B chunks.
// Math == inv_txfm2d_add_kernel (kernel_txfm.h) == inv_txfm2d_add_c (EbTransforms.c:8180).
// ---------------------------------------------------------------------------
// dst_offsets != NULL: destination block b starts at dst + dst_offsets[b] (samples) with row stride dst_stride
// (reconstruction written in place into a picture plane); NULL: dense W*H blocks back to back.
// TILE16 (bd <= 10: the column pass clamps its input to 16 bits anyway): the transpose tile holds the row-pass output
// already clamped, as int16 - half the LDS, which is what limits the 64-point sizes to 2 waves per SIMD.
template <int W, int H, typename PixT, bool TILE16 = false>
__global__ __launch_bounds__(StagedWaves<W * H>::N * 64) void inv_staged_kernel(
    const int32_t* __restrict__ in, PixT* __restrict__ dst, int tx_type, int bd, uint32_t nblocks,
    const uint32_t* __restrict__ dst_offsets = nullptr, int32_t dst_stride = 0) {
    using S = StagedGeom<W, H>;
    using G = TxGeom<W, H>;
    constexpr int KW = S::KW, KH = S::KH, NC = S::NC;
    constexpr int IN_BYTES = G::BPW * KH * KW * 4;        // coefficient tile, slots swizzled by the row (coef_tile_addr)
    constexpr int RPAD = ResPad<W>::N;
    constexpr int P16 = W + 2;                           // int16 tile row pitch: (W+2)/2 is odd -> conflict-free row writes
    constexpr int TILE_BYTES = TILE16 ? G::BPW * H * P16 * 2 : G::BPW * G::TILE * 4;
    constexpr int RES_BYTES = G::BPW * (W * H + RPAD) * 2;
    constexpr int WAVE_LDS = (cmax(cmax(IN_BYTES, TILE_BYTES), RES_BYTES) + 15) & ~15;
    __shared__ __attribute__((aligned(16))) char lds[S::WAVES * WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wl = lds + wave * WAVE_LDS;
    const uint32_t first = (blockIdx.x * S::WAVES + wave) * G::BPW;
    if (first >= nblocks) return;
    const int sub = lane / G::LPB, l = lane % G::LPB;
    const int vk = kVKind[tx_type], hk = kHKind[tx_type];
    const bool ud = vk == K1D_FLIPADST, lr = hk == K1D_FLIPADST;
    const int row_bits = bd == 8 ? 16 : (bd == 10 ? 18 : 20);
    const int col_bits = bd == 12 ? 18 : 16;
    const int in_bits = bd + 8;
    const int colin_bits = bd + 6 > 16 ? bd + 6 : 16;
    constexpr int S0 = inv_shift0(W, H);
    // ---- stage coefficients ----------------------------------------------------------------
    {
        constexpr int CPB = NC / 4, NCH = G::BPW * CPB;
        const int4* g = reinterpret_cast<const int4*>(in + (size_t)first * NC);
#pragma unroll
        for (int q0 = 0; q0 < NCH; q0 += 64) {
            const int q = q0 + lane;
            if (NCH % 64 == 0 || q < NCH) {
                const int b = q / CPB, w4 = q % CPB;
                int4 v = make_int4(0, 0, 0, 0);
                if (first + b < nblocks) v = g[q];
                *reinterpret_cast<int4*>(wl + coef_tile_addr<KW, KH>(b, w4 / (KW / 4), w4 % (KW / 4))) = v;
            }
        }
    }
    wave_lds_fence();
    // ---- row pass ----------------------------------------------------------------------------
    int x[W];
    if (l < H) {
        if (l < KH) {
#pragma unroll
            for (int s = 0; s < KW / 4; s++) {
                const int4 v = *reinterpret_cast<const int4*>(wl + coef_tile_addr<KW, KH>(sub, l, s));
                x[4 * s] = v.x; x[4 * s + 1] = v.y; x[4 * s + 2] = v.z; x[4 * s + 3] = v.w;
            }
#pragma unroll
            for (int c = 0; c < W; c++) {
                int v = c < KW ? x[c] : 0;
                if (G::RECT2) v = mul_q12(v, 2896);
                x[c] = svtgen::svt_clamp(v, -(1 << (in_bits - 1)), (1 << (in_bits - 1)) - 1);
            }
            inv1d<W>(hk, x, -(1 << (row_bits - 1)), (1 << (row_bits - 1)) - 1);
        } else {
#pragma unroll
            for (int c = 0; c < W; c++) x[c] = 0;
        }
    }
    wave_lds_fence();
    int32_t* tile = reinterpret_cast<int32_t*>(wl) + sub * G::TILE;
    short* tile16 = reinterpret_cast<short*>(wl) + sub * (H * P16);
    if (l < H) {
        if (TILE16) {
            const int lo16 = svtgen::svt_vgpr(-32768), hi16 = svtgen::svt_vgpr(32767);
#pragma unroll
            for (int c = 0; c < W; c++) tile16[l * P16 + c] = (short)svtgen::svt_clamp(round_shift_c<-S0>(x[c]), lo16, hi16);
        } else {
#pragma unroll
            for (int c = 0; c < W; c++) tile[l * G::PITCH + c] = round_shift_c<-S0>(x[c]);
        }
    }
    wave_lds_fence();
    // ---- column pass -------------------------------------------------------------------------
    int y[H];
    if (l < W) {
        const int csrc = lr ? W - 1 - l : l;
        if (TILE16) {
#pragma unroll
            for (int r = 0; r < H; r++) y[r] = tile16[r * P16 + csrc];          // already clamped to 16 bits (== colin range)
        } else {
#pragma unroll
            for (int r = 0; r < H; r++)
                y[r] = svtgen::svt_clamp(tile[r * G::PITCH + csrc], -(1 << (colin_bits - 1)), (1 << (colin_bits - 1)) - 1);
        }
        inv1d<H>(vk, y, -(1 << (col_bits - 1)), (1 << (col_bits - 1)) - 1);
    }
    wave_lds_fence();
    if (l < W) {
        short* res = reinterpret_cast<short*>(wl) + sub * (W * H + RPAD);
#pragma unroll
        // the add to the sample below runs on 16-bit lanes (the reference adds in int32, highbd_clip_pixel_add): exact for
        // bd <= 10, where every column kernel's output is at most 16 bits (12 after this shift, identities 14); bd 12 takes
        // the general kernel (svt_hip_inv_txfm2d_add_batch), so no wrap can occur here
        for (int r = 0; r < H; r++) res[r * W + l] = (short)round_shift_c<4>(y[ud ? H - 1 - r : r]);
    }
    wave_lds_fence();
    // ---- destination on a plane: chunks of CS bytes that never cross a block row ---------------------
    if (dst_offsets) {
        constexpr int ES = (int)sizeof(PixT);
        constexpr int ROWB = W * ES, CS = ROWB >= 16 ? 16 : ROWB, PPC = CS / ES;
        constexpr int CPR = ROWB / CS, CPBP = W * H * ES / CS, NCHP = G::BPW * CPBP, NIT = (NCHP + 63) / 64;
        const int maxpix = (1 << bd) - 1;
        uint32_t org[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, b = q / CPBP;
            org[it] = (q < NCHP && first + b < nblocks) ? dst_offsets[first + b] : 0xffffffffu;
        }
        uint4 pvv[NIT];
        PixT* dp[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, w = q % CPBP;
            pvv[it] = make_uint4(0, 0, 0, 0);
            dp[it] = dst + (size_t)(org[it] == 0xffffffffu ? 0u : org[it]) + (size_t)(w / CPR) * dst_stride + (w % CPR) * PPC;
            if (org[it] != 0xffffffffu) __builtin_memcpy(&pvv[it], dp[it], CS);
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane;
            if (org[it] != 0xffffffffu) {
                // q-th chunk of the wave = PPC consecutive residuals of the row-major int16 tile
                const uint32_t* rs = reinterpret_cast<const uint32_t*>(reinterpret_cast<const short*>(wl) + (size_t)q * PPC + (q / CPBP) * RPAD);
                const uint32_t pw[4] = {pvv[it].x, pvv[it].y, pvv[it].z, pvv[it].w};
                uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < CS / 4; k++) {
                    if (ES == 1) {
                        const uint32_t p01 = __builtin_amdgcn_perm(0u, pw[k], 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, pw[k], 0x0c030c02u);
                        const uint32_t u01 = sat_pk_u8_i16(pk_add_i16(p01, rs[2 * k])), u23 = sat_pk_u8_i16(pk_add_i16(p23, rs[2 * k + 1]));
                        ow[k] = (u23 << 16) | (u01 & 0xffffu);
                    } else {
                        ow[k] = pk_clamp_i16(pk_add_i16(pw[k], rs[k]), maxpix);
                    }
                }
                __builtin_memcpy(dp[it], ow, CS);
            }
        }
    } else
    // ---- destination: linear 16-B chunks --------------------------------------------------------
    {
        constexpr int PPL = 16 / (int)sizeof(PixT);
        constexpr int CPB = W * H / PPL, NCH = G::BPW * CPB;
        static_assert(W * H % PPL == 0, "block must be a whole number of 16-B chunks");
        const int maxpix = (1 << bd) - 1;
        uint4* d4 = reinterpret_cast<uint4*>(dst + (size_t)first * (W * H));
#pragma unroll
        for (int q0 = 0; q0 < NCH; q0 += 64) {
            const int q = q0 + lane;
            if ((NCH % 64 == 0 || q < NCH) && (first + q / CPB < nblocks)) {
                // residuals arrive as packed int16 pairs; the add / clip runs on 16-bit lanes (v_pk_add_i16,
                // v_sat_pk_u8_i16 or v_pk_max/min_i16) — SDWA / bfe byte arithmetic costs ~4x as much
                const uint4* rs4 = reinterpret_cast<const uint4*>(reinterpret_cast<const short*>(wl) + (size_t)q * PPL + (q / CPB) * RPAD);
                const uint4 pv = d4[q];
                const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
                uint32_t ow[4];
                if (sizeof(PixT) == 1) {
                    const uint4 ra = rs4[0], rb = rs4[1];
                    const uint32_t rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};     // (r0,r1) (r2,r3) ...
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t p01 = __builtin_amdgcn_perm(0u, pw[k], 0x0c010c00u);       // bytes 0,1 -> 16-bit lanes
                        const uint32_t p23 = __builtin_amdgcn_perm(0u, pw[k], 0x0c030c02u);       // bytes 2,3
                        const uint32_t u01 = sat_pk_u8_i16(pk_add_i16(p01, rw[2 * k])), u23 = sat_pk_u8_i16(pk_add_i16(p23, rw[2 * k + 1]));
                        ow[k] = (u23 << 16) | (u01 & 0xffffu);
                    }
                } else {
                    const uint4 ra = rs4[0];
                    const uint32_t rw[4] = {ra.x, ra.y, ra.z, ra.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) ow[k] = pk_clamp_i16(pk_add_i16(pw[k], rw[k]), maxpix);
                }
                d4[q] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// enc_staged_kernel<W, H, KEEP> — the encode-pass chain (see enc32_kernel in kernel_fused32.h) for every
// other size, 8-bit dense batches: fwd_staged_kernel<W, H, 1> up to the quantiser, whose dequantised 16-B
// chunks are written straight into the coefficient rows inv_staged_kernel starts from; the prediction
// chunks loaded for the residual stay in registers and are the destination of the reconstruction.
// HBM traffic: 2*W*H in + 4*KW*KH (qcoeff) + W*H (recon) + 6 B out (+ coeff, dqcoeff when KEEP).
// ---------------------------------------------------------------------------
template <int W, int H, typename PixT>
struct EncStagedLds {                                    // LDS bytes per wave / per workgroup of enc_staged_body<W, H, ., PixT>
    using S = StagedGeom<W, H>;
    using G = TxGeom<W, H>;
    static constexpr int ES = (int)sizeof(PixT), BB = W * H * ES;
    // input staging: blocks padded by PADI bytes so that the column pass's per-sample reads spread over the banks.  (For 64-byte
    // blocks - 8x8 8-bit - the pad makes the linear 16-byte staging stores 2-way conflicts; the unpadded form with the chunks of
    // block b XOR-swizzled by b >> 1 is conflict-free on both sides and was measured 2 % SLOWER, A/B on one box: the kernel is
    // bound by VALU issue, and the swizzle puts address arithmetic into the column pass.  SWZ_IN keeps that form selectable.)
    static constexpr bool SWZ_IN = false;
    static constexpr int PADI = SWZ_IN ? 0 : ((W * ES >= 32) ? 32 : 16), IN_ONE = G::BPW * (BB + PADI);
    static constexpr int RPAD = ResPad<W>::N;              // residual tile: int16 per block = W * H + RPAD
    static constexpr int WAVE = (cmax(cmax(IN_ONE * 2, G::BPW * G::TILE * 4), G::BPW * W * H * 4) + 15) & ~15;
    static constexpr int BYTES = S::WAVES * WAVE;
};
template <int W, int H, bool KEEP, typename PixT, int BD>
__device__ __forceinline__ void enc_staged_body(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, const QParams& qp,
    int tx_type, uint32_t nblocks, const uint32_t* __restrict__ xy, uint32_t src_stride,
    uint32_t pred_stride, uint32_t recon_stride, uint32_t bid, char* lds) {
    // xy != NULL: blocks addressed on picture planes (origin (x, y) = (xy[b] & 0xffff, xy[b] >> 16), row strides
    // in samples; recon may be the prediction plane itself); NULL: dense batches.
    using S = StagedGeom<W, H>;
    using G = TxGeom<W, H>;
    constexpr int KW = S::KW, KH = S::KH, NC = S::NC;
    constexpr int ES = (int)sizeof(PixT);                // 1, or 2 for 10-bit samples (BD = 10)
    constexpr int BB = W * H * ES;                       // input bytes per block and array
    using L = EncStagedLds<W, H, PixT>;
    constexpr bool SWZ_IN = L::SWZ_IN;
    constexpr int PADI = L::PADI, IN_ONE = L::IN_ONE, RPAD = L::RPAD, WAVE_LDS = L::WAVE;
    auto dq_addr = [](int b, int r, int sl) { return coef_tile_addr<KW, KH>(b, r, sl); };      // dequantised rows -> inverse row pass
    static_assert(W * H % 16 == 0, "block must be a whole number of 16-B chunks");
    constexpr int in_bits = BD + 8, row_bits = BD == 8 ? 16 : (BD == 10 ? 18 : 20);      // av1_gen_inv_stage_range (:5404-5456)
    constexpr int cin_bits = BD + 6 > 16 ? BD + 6 : 16, col_bits = BD == 12 ? 18 : 16, maxpix = (1 << BD) - 1;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wl = lds + wave * WAVE_LDS;
    if (wave >= S::WAVES) return;                           // (run inside a larger workgroup: the spare waves have nothing to do)
    const uint32_t first = (bid * S::WAVES + wave) * G::BPW;
    if (first >= nblocks) return;                         // wave-uniform
    const int sub = lane / G::LPB, l = lane % G::LPB;
    const uint32_t blk = first + sub;
    const bool valid = blk < nblocks;
    const int vk = kVKind[tx_type], hk = kHKind[tx_type];
    const bool ud = vk == K1D_FLIPADST, lr = hk == K1D_FLIPADST;
    constexpr int CBC = fwd_cos_col(W, H), CBR = fwd_cos_row(W, H);
    constexpr int S0 = fwd_shift(W, H, 0), S1 = fwd_shift(W, H, 1), S2 = fwd_shift(W, H, 2);

    // ---- stage the wave's input: linear 16-B chunks; the prediction chunks stay in registers ----
    constexpr int NCH = G::BPW * BB / 16, NCHI = (NCH + 63) / 64;
    // plane mode: chunks of CS bytes that never cross a block row (same linear LDS image)
    constexpr int ROWB = W * ES, CS = ROWB >= 16 ? 16 : ROWB, PPC = CS / ES;
    constexpr int CPR = ROWB / CS, CPBP = BB / CS, NCHP = G::BPW * CPBP, NIT = (NCHP + 63) / 64;
    uint4 pk[NIT > NCHI ? NIT : NCHI];
    uint32_t org[NIT];
    if (xy) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, b = q / CPBP;
            org[it] = (q < NCHP && first + b < nblocks) ? xy[first + b] : 0xffffffffu;
        }
        uint4 v0[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, w = q % CPBP;
            v0[it] = make_uint4(0, 0, 0, 0); pk[it] = v0[it];
            if (org[it] != 0xffffffffu) {
                const size_t y = (org[it] >> 16) + w / CPR, x = (org[it] & 0xffffu) + (w % CPR) * PPC;
                __builtin_memcpy(&v0[it], src + y * src_stride + x, CS);
                __builtin_memcpy(&pk[it], pred + y * pred_stride + x, CS);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, b = q / CPBP;
            if (NCHP % 64 == 0 || q < NCHP) {
                // (SWZ_IN: 8-byte rows - two of them form a 16-byte chunk, chunk index (q % CPBP) >> 1)
                const int o = SWZ_IN ? b * BB + (((((q % CPBP) * CS) >> 4) ^ ((b >> 1) & 3)) << 4) + (((q % CPBP) * CS) & 15) : q * CS + b * PADI;
                __builtin_memcpy(wl + o, &v0[it], CS);
                __builtin_memcpy(wl + IN_ONE + o, &pk[it], CS);
            }
        }
    } else {
        const char* g0 = reinterpret_cast<const char*>(src) + (size_t)first * BB;
        const char* g1 = reinterpret_cast<const char*>(pred) + (size_t)first * BB;
#pragma unroll
        for (int it = 0; it < NCHI; it++) {
            const int q = it * 64 + lane;
            pk[it] = make_uint4(0, 0, 0, 0);
            if (NCH % 64 == 0 || q < NCH) {
                const int b = (q * 16) / BB;
                const bool ok = first + b < nblocks;
                uint4 va = make_uint4(0, 0, 0, 0);
                if (ok) { pk[it] = *reinterpret_cast<const uint4*>(g1 + (size_t)q * 16); va = *reinterpret_cast<const uint4*>(g0 + (size_t)q * 16); }
                const int o = SWZ_IN ? b * BB + (((q & 3) ^ ((b >> 1) & 3)) << 4) : q * 16 + b * PADI;
                *reinterpret_cast<uint4*>(wl + o) = va;
                *reinterpret_cast<uint4*>(wl + IN_ONE + o) = pk[it];
            }
        }
    }
    wave_lds_fence();
    // ---- forward: column pass ------------------------------------------------------------------
    unsigned sad_acc = 0;
    {
        int x[H];
        if (l < W) {
            const char* bs = wl + sub * (BB + PADI);
#pragma unroll
            for (int r = 0; r < H; r++) {
                const int idx = (ud ? H - 1 - r : r) * W + l;
                const int io = SWZ_IN ? ((((idx >> 4) ^ ((sub >> 1) & 3)) << 4) | (idx & 15)) : idx * ES;
                const int d = (int)*reinterpret_cast<const PixT*>(bs + io) - (int)*reinterpret_cast<const PixT*>(bs + IN_ONE + io);
                sad_acc += (unsigned)(d < 0 ? -d : d);
                x[r] = round_shift_c<-S0>(d);
            }
            fwd1d<H, CBC>(vk, x);
        }
        wave_lds_fence();                                 // staging is dead: the tile may overwrite it
        int32_t* tile = reinterpret_cast<int32_t*>(wl) + sub * G::TILE;
        if (l < W) {
            const int cdst = lr ? W - 1 - l : l;
#pragma unroll
            for (int r = 0; r < H; r++) tile[r * G::PITCH + cdst] = round_shift_c<-S1>(x[r]);
        }
    }
    wave_lds_fence();
    // ---- forward: row pass ----------------------------------------------------------------------
    {
        int32_t* tile = reinterpret_cast<int32_t*>(wl) + sub * G::TILE;
        int y[W];
        if (l < H) {
#pragma unroll
            for (int c = 0; c < W; c++) y[c] = tile[l * G::PITCH + c];
            fwd1d<W, CBR>(hk, y);
#pragma unroll
            for (int c = 0; c < W; c++) {
                int t = round_shift_c<-S2>(y[c]);
                if (G::RECT2) t = mul_q12(t, 5793);
                y[c] = t;
            }
        }
        wave_lds_fence();                                 // tile is dead: the out tile may overwrite it
        if (l < H) {
#pragma unroll
            for (int s = 0; s < W / 4; s++)
                *reinterpret_cast<int4*>(wl + S::out_addr(sub, l, s)) = make_int4(y[4 * s], y[4 * s + 1], y[4 * s + 2], y[4 * s + 3]);
        }
    }
    wave_lds_fence();
    sad_acc = group_sum<G::LPB>(sad_acc);
    if (valid && l == 0 && sad) sad[blk] = sad_acc;
    // ---- quantise in linear chunk order; dequantised chunks stay in registers ----------------------
    constexpr int CPB = NC / 4, NOUT = G::BPW * CPB, NOUTI = (NOUT + 63) / 64;
    int4 dvs[NOUTI];
    {
        int eob_acc = 0;
#pragma unroll
        for (int it = 0; it < NOUTI; it++) {
            const int q = it * 64 + lane;
            const bool act = (NOUT % 64 == 0) || q < NOUT;
            const int b = act ? q / CPB : 0, w4 = act ? q % CPB : 0;
            const bool ok = act && (first + b < nblocks);
            const int4 c = *reinterpret_cast<const int4*>(wl + S::out_addr(b, w4 / (KW / 4), w4 % (KW / 4)));
            int4 qv, dv;
            quant_one<2>(c.x, w4 == 0 ? 0 : 1, qp, qv.x, dv.x);     // the host only takes this kernel for power-of-two quant_shift
            quant_one<2>(c.y, 1, qp, qv.y, dv.y);
            quant_one<2>(c.z, 1, qp, qv.z, dv.z);
            quant_one<2>(c.w, 1, qp, qv.w, dv.w);
            dvs[it] = dv;
            const uint2 is = *reinterpret_cast<const uint2*>(iscan + w4 * 4);
            int e = max(max(qv.x ? (int)(is.x & 0xffffu) + 1 : 0, qv.y ? (int)(is.x >> 16) + 1 : 0),
                        max(qv.z ? (int)(is.y & 0xffffu) + 1 : 0, qv.w ? (int)(is.y >> 16) + 1 : 0));
            if (!act) e = 0;
            if (ok) {
                const size_t o = (size_t)(first + b) * NC + (size_t)w4 * 4;
                *reinterpret_cast<int4*>(qcoeff + o) = qv;
                if (KEEP) { *reinterpret_cast<int4*>(coeff + o) = c; *reinterpret_cast<int4*>(dqcoeff + o) = dv; }
            }
            if constexpr (CPB >= 64) {
                eob_acc = max(eob_acc, e);
                if ((it + 1) % (CPB / 64) == 0) {
                    const int m = group_max<64>(eob_acc);
                    if (lane == 0 && ok) eob[first + b] = (uint16_t)m;
                    eob_acc = 0;
                }
            } else {
                const int m = group_max<(CPB < 64 ? CPB : 64)>(e);
                if (ok && w4 == 0) eob[first + b] = (uint16_t)m;
            }
        }
    }
    wave_lds_fence();                                     // the out tile is dead: coefficient rows may overwrite it
#pragma unroll
    for (int it = 0; it < NOUTI; it++) {
        const int q = it * 64 + lane;
        if (NOUT % 64 == 0 || q < NOUT) {
            const int b = q / CPB, w4 = q % CPB;
            *reinterpret_cast<int4*>(wl + dq_addr(b, w4 / (KW / 4), w4 % (KW / 4))) = dvs[it];
        }
    }
    wave_lds_fence();
    // ---- inverse (inv_staged_kernel<W, H, uint8_t>, bd = 8) -------------------------------------
    constexpr int IS0 = inv_shift0(W, H);
    {
        int x[W];
        if (l < H) {
            if (l < KH) {
#pragma unroll
                for (int s = 0; s < KW / 4; s++) {
                    const int4 v = *reinterpret_cast<const int4*>(wl + dq_addr(sub, l, s));
                    x[4 * s] = v.x; x[4 * s + 1] = v.y; x[4 * s + 2] = v.z; x[4 * s + 3] = v.w;
                }
#pragma unroll
                for (int c = 0; c < W; c++) {
                    int v = c < KW ? x[c] : 0;
                    if (G::RECT2) v = mul_q12(v, 2896);
                    x[c] = svtgen::svt_clamp(v, -(1 << (in_bits - 1)), (1 << (in_bits - 1)) - 1);
                }
                inv1d<W>(hk, x, -(1 << (row_bits - 1)), (1 << (row_bits - 1)) - 1);
            } else {
#pragma unroll
                for (int c = 0; c < W; c++) x[c] = 0;
            }
        }
        wave_lds_fence();
        int32_t* tile = reinterpret_cast<int32_t*>(wl) + sub * G::TILE;
        if (l < H) {
#pragma unroll
            for (int c = 0; c < W; c++) tile[l * G::PITCH + c] = round_shift_c<-IS0>(x[c]);
        }
    }
    wave_lds_fence();
    {
        int32_t* tile = reinterpret_cast<int32_t*>(wl) + sub * G::TILE;
        int y[H];
        if (l < W) {
            const int csrc = lr ? W - 1 - l : l;
#pragma unroll
            for (int r = 0; r < H; r++) y[r] = svtgen::svt_clamp(tile[r * G::PITCH + csrc], -(1 << (cin_bits - 1)), (1 << (cin_bits - 1)) - 1);
            inv1d<H>(vk, y, -(1 << (col_bits - 1)), (1 << (col_bits - 1)) - 1);
        }
        wave_lds_fence();
        if (l < W) {
            short* res = reinterpret_cast<short*>(wl) + sub * (W * H + RPAD);
#pragma unroll
            for (int r = 0; r < H; r++) res[r * W + l] = (short)round_shift_c<4>(y[ud ? H - 1 - r : r]);
        }
    }
    wave_lds_fence();
    // ---- reconstruction = prediction (still in registers) + residual ---------------------------------
    if (xy) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * 64 + lane, w = q % CPBP;
            if (org[it] != 0xffffffffu) {
                const uint32_t* rs = reinterpret_cast<const uint32_t*>(reinterpret_cast<const short*>(wl) + (size_t)q * PPC + (q / CPBP) * RPAD);
                const uint32_t pw[4] = {pk[it].x, pk[it].y, pk[it].z, pk[it].w};
                uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < CS / 4; k++) {
                    if constexpr (ES == 1) {
                        const uint32_t p01 = __builtin_amdgcn_perm(0u, pw[k], 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, pw[k], 0x0c030c02u);
                        const uint32_t u01 = sat_pk_u8_i16(pk_add_i16(p01, rs[2 * k])), u23 = sat_pk_u8_i16(pk_add_i16(p23, rs[2 * k + 1]));
                        ow[k] = (u23 << 16) | (u01 & 0xffffu);
                    } else {
                        ow[k] = pk_clamp_i16(pk_add_i16(pw[k], rs[k]), maxpix);
                    }
                }
                const size_t y = (org[it] >> 16) + w / CPR, x = (org[it] & 0xffffu) + (w % CPR) * PPC;
                __builtin_memcpy(recon + y * recon_stride + x, ow, CS);
            }
        }
    } else {
        uint4* d4 = reinterpret_cast<uint4*>(reinterpret_cast<char*>(recon) + (size_t)first * BB);
        constexpr int PP16 = 16 / ES;                     // pixels per 16-B chunk
#pragma unroll
        for (int it = 0; it < NCHI; it++) {
            const int q = it * 64 + lane;
            if ((NCH % 64 == 0 || q < NCH) && (first + (q * 16) / BB < nblocks)) {
                const uint4* rs4 = reinterpret_cast<const uint4*>(reinterpret_cast<const short*>(wl) + (size_t)q * PP16 + ((q * 16) / BB) * RPAD);
                const uint32_t pw[4] = {pk[it].x, pk[it].y, pk[it].z, pk[it].w};
                uint32_t ow[4];
                if constexpr (ES == 1) {
                    const uint4 ra = rs4[0], rb = rs4[1];
                    const uint32_t rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t p01 = __builtin_amdgcn_perm(0u, pw[k], 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, pw[k], 0x0c030c02u);
                        const uint32_t u01 = sat_pk_u8_i16(pk_add_i16(p01, rw[2 * k])), u23 = sat_pk_u8_i16(pk_add_i16(p23, rw[2 * k + 1]));
                        ow[k] = (u23 << 16) | (u01 & 0xffffu);
                    }
                } else {
                    const uint4 ra = rs4[0];
                    const uint32_t rw[4] = {ra.x, ra.y, ra.z, ra.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) ow[k] = pk_clamp_i16(pk_add_i16(pw[k], rw[k]), maxpix);
                }
                d4[q] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        }
    }
}

template <int W, int H, bool KEEP, typename PixT = uint8_t, int BD = 8>
__global__ __launch_bounds__(StagedWaves<W * H>::N * 64) void enc_staged_kernel(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, QParams qp,
    int tx_type, uint32_t nblocks, const uint32_t* __restrict__ xy = nullptr, uint32_t src_stride = 0,
    uint32_t pred_stride = 0, uint32_t recon_stride = 0) {
    __shared__ __attribute__((aligned(16))) char lds[EncStagedLds<W, H, PixT>::BYTES];
    enc_staged_body<W, H, KEEP, PixT, BD>(src, pred, recon, coeff, qcoeff, dqcoeff, eob, sad, iscan, qp, tx_type, nblocks, xy, src_stride,
                                          pred_stride, recon_stride, blockIdx.x, lds);
}

// ---------------------------------------------------------------------------
// enc4_kernel<PixT, BD, KEEP> — the encode-pass chain for TX_4X4, all 16 transform types: ONE LANE PER BLOCK, the whole
// block in registers (16 residuals, two passes of four 4-point transforms each way), no LDS.  Adjacent lanes take adjacent
// blocks, so on planes a row of 64 blocks is fetched as 64 adjacent 4-sample loads per block row.  Replaces the two-stage
// path (forward + quantise kernel, device copy, inverse kernel) the 4x4 blocks took in round 1: 2 x 16 B in, 64 B qcoeff +
// 16 B reconstruction out per block instead of two launches and the coeff / dqcoeff round trip through HBM.
// Reference chain: Av1EncodeLoop (EbCodingLoop.c:617-753) with av1_fwd_txfm2d_4x4 (shift {2, 0, 0}, cos_bit 13 / 13,
// EbTransforms.h:120-156), aom_highbd_quantize_b (log_scale 0) and av1_inv_txfm2d_add_4x4 (shift {0, -4}).
// qfast: the quantiser table has power-of-two quant_shift (host-checked): one-product form; else the exact 64-bit form.
// ---------------------------------------------------------------------------
template <typename PixT, int BD, bool KEEP>
__device__ __forceinline__ void enc4_body(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, const QParams& qp, int qfast,
    int tx_type, uint32_t nblocks, const uint32_t* __restrict__ xy, uint32_t src_stride, uint32_t pred_stride,
    uint32_t recon_stride, uint32_t bid) {
    const uint32_t blk = bid * 256u + threadIdx.x;
    if (blk >= nblocks) return;
    constexpr int in_bits = BD + 8, row_bits = BD == 8 ? 16 : (BD == 10 ? 18 : 20);      // av1_gen_inv_stage_range (:5404-5456)
    constexpr int cin_bits = BD + 6 > 16 ? BD + 6 : 16, col_bits = BD == 12 ? 18 : 16, maxpix = (1 << BD) - 1;
    const int vk = kVKind[tx_type], hk = kHKind[tx_type];
    const bool ud = vk == K1D_FLIPADST, lr = hk == K1D_FLIPADST;
    size_t so, po, ro;
    uint32_t ss = 4, ps = 4, rs = 4;
    if (xy) {
        const uint32_t o = xy[blk];
        const size_t x = o & 0xffffu, y = o >> 16;
        ss = src_stride; ps = pred_stride; rs = recon_stride;
        so = y * ss + x; po = y * ps + x; ro = y * rs + x;
    } else {
        so = po = ro = (size_t)blk * 16;
    }
    PixT sv[4][4], pv[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        __builtin_memcpy(sv[r], src + so + (size_t)r * ss, 4 * sizeof(PixT));
        __builtin_memcpy(pv[r], pred + po + (size_t)r * ps, 4 * sizeof(PixT));
    }
    int d[4][4];
    unsigned sad_acc = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            d[r][c] = (int)sv[r][c] - (int)pv[r][c];
            sad_acc += (unsigned)(d[r][c] < 0 ? -d[r][c] : d[r][c]);
        }
    if (sad) sad[blk] = sad_acc;
    // ---- forward: columns (up-shift 2, ud flip on the way in, lr flip on the way out), then rows ----
    int t[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int x[4];
#pragma unroll
        for (int r = 0; r < 4; r++) x[r] = (ud ? d[3 - r][c] : d[r][c]) * 4;
        fwd1d<4, 13>(vk, x);
#pragma unroll
        for (int r = 0; r < 4; r++) t[r][c] = x[r];
    }
    int co[16];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int y[4];
#pragma unroll
        for (int c = 0; c < 4; c++) y[c] = lr ? t[r][3 - c] : t[r][c];      // column c of the pass-1 output went to 3 - c
        fwd1d<4, 13>(hk, y);
#pragma unroll
        for (int c = 0; c < 4; c++) co[r * 4 + c] = y[c];
    }
    // ---- quantise / dequantise / eob ----
    int q[16], dq[16];
    int e = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (qfast) quant_one<2>(co[i], i == 0 ? 0 : 1, qp, q[i], dq[i]);
        else quant_one<0>(co[i], i == 0 ? 0 : 1, qp, q[i], dq[i]);
        e = max(e, q[i] ? (int)iscan[i] + 1 : 0);
    }
    eob[blk] = (uint16_t)e;
    {
        int4* qo = reinterpret_cast<int4*>(qcoeff + (size_t)blk * 16);
#pragma unroll
        for (int r = 0; r < 4; r++) __builtin_memcpy(qo + r, &q[4 * r], 16);
        if (KEEP) {
            int4* c4 = reinterpret_cast<int4*>(coeff + (size_t)blk * 16);
            int4* d4 = reinterpret_cast<int4*>(dqcoeff + (size_t)blk * 16);
#pragma unroll
            for (int r = 0; r < 4; r++) { __builtin_memcpy(c4 + r, &co[4 * r], 16); __builtin_memcpy(d4 + r, &dq[4 * r], 16); }
        }
    }
    // ---- inverse: rows (clamp bd + 8, shift 0), columns (clamp, shift 4), flips, add, clip ----
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int x[4];
#pragma unroll
        for (int c = 0; c < 4; c++) x[c] = svtgen::svt_clamp(dq[r * 4 + c], -(1 << (in_bits - 1)), (1 << (in_bits - 1)) - 1);
        inv1d<4>(hk, x, -(1 << (row_bits - 1)), (1 << (row_bits - 1)) - 1);
#pragma unroll
        for (int c = 0; c < 4; c++) t[r][c] = x[c];
    }
    PixT ov[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int y[4];
#pragma unroll
        for (int r = 0; r < 4; r++) y[r] = svtgen::svt_clamp(lr ? t[r][3 - c] : t[r][c], -(1 << (cin_bits - 1)), (1 << (cin_bits - 1)) - 1);
        inv1d<4>(vk, y, -(1 << (col_bits - 1)), (1 << (col_bits - 1)) - 1);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int res = round_shift_c<4>(ud ? y[3 - r] : y[r]);
            ov[r][c] = (PixT)min(max((int)pv[r][c] + res, 0), maxpix);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) __builtin_memcpy(recon + ro + (size_t)r * rs, ov[r], 4 * sizeof(PixT));
}

template <typename PixT, int BD, bool KEEP>
__global__ __launch_bounds__(256) void enc4_kernel(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, QParams qp, int qfast,
    int tx_type, uint32_t nblocks, const uint32_t* __restrict__ xy, uint32_t src_stride, uint32_t pred_stride,
    uint32_t recon_stride) {
    enc4_body<PixT, BD, KEEP>(src, pred, recon, coeff, qcoeff, dqcoeff, eob, sad, iscan, qp, qfast, tx_type, nblocks, xy, src_stride, pred_stride,
                              recon_stride, blockIdx.x);
}

}  // namespace svtdev
