// TN GEMM for gfx950 (weight gradients):  C[M,N] (f32) = sum_k A[k][m] * B[k][n], both operands stored
// with the contraction index k as the SLOW dimension (rows = pixels / pairs, columns = channels), which is
// how every activation and activation-gradient of the relation head already lies in HBM (channels-last).
//
//  * No transposed copies: tiles are staged with global_load_lds_dwordx4 into LDS as 4(k) x 16(m)
//    sub-blocks of 128 contiguous bytes, and the k-contiguous MFMA fragments are produced by the CDNA4
//    LDS transpose read ds_read_b64_tr_b16 (each 16-lane group reads one 4x16 block and receives its
//    columns).  A and B use the same k->slot assignment, so the dot products are exact w.r.t. ordering.
//  * BMODE_CONV: B rows are gathered from a zero-padded channels-last image at the pixel of row k shifted
//    by the tap of the N tile (N = 9 taps x Cin), i.e. the 3x3 conv weight gradient with no im2col.
//  * Split-K over blockIdx.y into f32 slabs C[split][M][ldc] (deterministic; reduced by a later kernel).
#pragma once
#include "common.h"
#include "gemm_nt.h"

enum { BMODE_PLAIN = 0, BMODE_CONV = 1, BMODE_GATHER = 2, BMODE_PATCH = 3 };   // GATHER / PATCH (ping-pong block only): conv rows from a window list / from per-window 4x4 patches, see TnParams

struct TnParams {
    const u16* A; const u16* B; float* C;
    int M, N, K;                 // K multiple of 64 (zero rows pad it), M,N multiples of 128
    long lda, ldb, ldc;
    long slab_stride;            // elements between split-K slabs
    int lgS, Cin;                // BMODE_CONV
    int CinA;                    // ACONV: A rows are the centre pixels of a zero-padded image with CinA channels
    int tiles_m, tiles_n, ktiles_per_split, splits;
    int xcd_map;                 // conv3 wgrad only: XCD-aware tile assignment (see kernel)
    int xcd_patch;               // ping-pong block: per-XCD 4x8 tile patches (tile count per split divisible by 8)
    const int* gather;           // BMODE_GATHER: contraction row r = pixel r&3 of window gather[r>>2] = image*64 + window of 16x16 maps
                                 // [img][18][18][Cin] (csrc/kernels_shared.hip): the conv weight gradient over LISTED windows with no
                                 // im2col buffer.  A k block of 4 rows is one window, so the row base is wave-uniform (scalar load
                                 // of the list entry) and only the pixel / tap / column part is per lane.
                                 // BMODE_PATCH: B = [windows][16 pixels of the window's 4 x 4 input patch][Cin] (sgc_windows_im2patch);
                                 // contraction row r = own pixel r&3 of window r>>2, its value for tap (ky, kx) is patch pixel
                                 // (qy + ky, qx + kx): 16 instead of the 36 rows per window of the im2col form, fixed per-lane offsets.
    const int* goff;             // ping-pong block, grouped form: block -> (group g, tile); the contraction runs over rows goff[g] .. goff[g+1]
                                 // (multiples of 64) and the result goes to columns g*N .. of C (no split-K, splits = number of groups)
};

template <int ELEM, int BMODE, int ACONV, int WR, int WC, int TM, int TN>
__global__ __launch_bounds__(WR * WC * 64, 2) void gemm_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WR * WC;
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int MB16 = BM / 16, NB16 = BN / 16;             // 4x16 sub-blocks (128 B) per 4-row k block
    constexpr int A_BYTES = 64 * BM * 2, B_BYTES = 64 * BN * 2, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int AI = A_BYTES / 1024 / NW, BI = B_BYTES / 1024 / NW;   // global_load_lds instructions per wave
    constexpr int APK = MB16 / 8, BPK = NB16 / 8;             // instructions per k block row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Work mapping: K split slowest; inside a split the tile grid is walked in 16 x 16 super-tiles (supertile_map), so the
    // blocks in flight sit at the same K position of a square patch of tiles and share their A/B rows through L2 / MALL.
    // (An XCD-grouped mapping - all N tiles of one (M tile, split) on one XCD - was measured 9 % slower on the conv3
    // weight gradient: the tail of each group starts out of phase with its head and re-fetches the slices.)
    const int tiles = p.tiles_n * p.tiles_m;
    int split, tm, tn;
    if (p.xcd_map) {
        // conv3 weight gradient (4 M tiles x 18 N tiles = 9 taps x 2 channel halves): XCD x (= block id mod 8) owns
        // M tile x&3 and channel half x>>2, i.e. a quarter of dy3 and half of z instead of all of dy3 and half of z;
        // the 72 tiles of one K split still start together (9 per XCD).
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        split = j / 9;
        tm = xcd & 3;
        tn = (j - split * 9) * 2 + (xcd >> 2);
    } else {
        split = blockIdx.x / tiles;
        supertile_map(blockIdx.x - split * tiles, p.tiles_m, p.tiles_n, tm, tn);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    const int nk_total = p.K >> 6;
    if (kt_end > nk_total) kt_end = nk_total;

    // loader: one instruction fills 8 sub-blocks = 4 k rows x 128 columns
    const int kr = (lane >> 1) & 3;
    const int mo = ((lane >> 3) * 2 + (lane & 1)) * 8;
    long boff_tap = 0;
    int bcol0 = n0;
    if constexpr (BMODE == BMODE_CONV) {
        const int tap = n0 / p.Cin;
        bcol0 = n0 - tap * p.Cin;
        const int ky = tap / 3, kx = tap - 3 * ky;
        boff_tap = (long)(ky * ((1 << p.lgS) + 2) + kx) * p.Cin;
    }

    // Row addresses split into a per-lane constant (row inside a 64-row K tile, computed once) and a wave-uniform
    // per-tile offset (scalar ALU): a K tile of 64 window-major rows starts at window column 0, so the padded-pixel
    // offset of row kt*64 + l is conv_row_base(kt*64) + conv_row_base(l).
    long a_loc[AI], b_loc[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int q = wid * AI + i;
        const int kl = (q / APK) * 4 + kr;
        const int mcol = (q % APK) * 128 + mo;
        if constexpr (ACONV) a_loc[i] = conv_row_base(kl, p.lgS, p.CinA) + (long)((1 << p.lgS) + 3) * p.CinA + m0 + mcol;
        else a_loc[i] = (long)kl * p.lda + m0 + mcol;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int q = wid * BI + i;
        const int kl = (q / BPK) * 4 + kr;
        const int ncol = (q % BPK) * 128 + mo;
        if constexpr (BMODE == BMODE_CONV) b_loc[i] = conv_row_base(kl, p.lgS, p.Cin) + boff_tap + bcol0 + ncol;
        else b_loc[i] = (long)kl * p.ldb + n0 + ncol;
    }
    auto stage = [&](int buf, int kt) {
        char* abase = smem + buf * BUF_BYTES + wid * (AI * 1024);
        char* bbase = smem + buf * BUF_BYTES + A_BYTES + wid * (BI * 1024);
        long ta, tb;
        if constexpr (ACONV) ta = conv_row_base(kt * 64, p.lgS, p.CinA); else ta = (long)kt * 64 * p.lda;
        if constexpr (BMODE == BMODE_CONV) tb = conv_row_base(kt * 64, p.lgS, p.Cin); else tb = (long)kt * 64 * p.ldb;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(p.A + ta + a_loc[i]), LDS_PTR(abase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BI; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(p.B + tb + b_loc[i]), LDS_PTR(bbase + i * 1024), 16, 0, 0);
    };

    const int wr = wid / WC, wc = wid % WC;
    const int g = (lane >> 4) & 1, kh = lane >> 5, t = lane & 15;
    const int lane_off = (t >> 2) * 32 + (t & 3) * 8;
    int a_blk[TM], b_blk[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_blk[i] = (wr * TM * 2 + i * 2 + g) * 128 + lane_off;
#pragma unroll
    for (int i = 0; i < TN; ++i) b_blk[i] = (wc * TN * 2 + i * 2 + g) * 128 + lane_off;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) stage(0, kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int it = kt - kt_begin;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < kt_end) stage((it + 1) & 1, kt + 1);
        const char* ab = smem + (it & 1) * BUF_BYTES;
        const char* bb = ab + A_BYTES;
        // The compiler groups the 12 transpose reads of a k-step ahead of its 8 MFMAs.  Forcing a finer
        // read/MFMA interleave with sched_group_barrier was measured 8 % SLOWER on the conv3 weight gradient
        // (84.3 vs 77.8 ms); a register double buffer that pins the reads of k-step ks+1 ahead of the MFMAs of k-step ks
        // was neutral (74.4 vs 74.3 ms): the kernel is bound by the L2->LDS stream, not by LDS-read latency.  The schedule
        // is left to hipcc.
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kb = ks * 4 + kh * 2;          // first of two 4-row k blocks
            s16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(ab + kb * (MB16 * 128) + a_blk[i]));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(ab + (kb + 1) * (MB16 * 128) + a_blk[i]));
                af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(bb + kb * (NB16 * 128) + b_blk[i]));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(bb + (kb + 1) * (NB16 * 128) + b_blk[i]));
                bf[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<ELEM>(af[i], bf[j], acc[i][j]);
        }
    }

    float* C = p.C + (long)split * p.slab_stride;
    const int h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * TN * 32 + j * 32 + cl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[(long)row * p.ldc + col] = acc[i][j][r];
            }
        }
}

template <int ELEM, int BMODE, int ACONV, int WR, int WC, int TM, int TN>
static int launch_gemm_tn_cfg(TnParams p, int splits, int* slabs_out, hipStream_t stream) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int LDS = 2 * 64 * (BM + BN) * 2;
    p.tiles_m = p.M / BM;
    p.tiles_n = p.N / BN;
    const int nk = p.K >> 6;
    if (splits < 1) splits = 1;
    if (splits > nk) splits = nk;
    p.ktiles_per_split = (nk + splits - 1) / splits;
    splits = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
    auto kern = gemm_tn_kernel<ELEM, BMODE, ACONV, WR, WC, TM, TN>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    p.splits = splits;
    {
        const int xm = sgc_tuning().tn_xcd;      // XCD-aware assignment of the 4 x 18 tile grid (measured -0.7 % time, 2.5x less fabric traffic)
        p.xcd_map = (xm && BMODE == BMODE_CONV && p.tiles_m == 4 && p.tiles_n == 18) ? 1 : 0;
    }
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n * splits)), dim3(WR * WC * 64), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    if (slabs_out) *slabs_out = splits;
    return SGC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong scheduled variant of the 256x256 block (schedule, ring and hazard rules: see gemm_nt_pp.h).  Half tiles are
// 16(k blocks of 4 rows) x 8(sub-blocks of 16 columns) x 128 B = 16 KiB; A0/A1 hold the column halves {0-63,128-191} /
// {64-127,192-255} of the M tile (the a-half of both wave rows), B0/B1 the 32-column halves of the four wave columns.
// Phase X reads A0 and both B halves (32 ds_read_b64_tr_b16), phase Y reads A1 (16).
template <int ELEM, int BMODE, int ACONV>
__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HT = 16384, BM = 256, BN = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tiles = p.tiles_n * p.tiles_m;
    int split, tm, tn;
    if (p.xcd_map == 3) {
        // grouped form, 64 groups (window positions w = 8 row + col) of 64 tiles (round 6): ALL tiles of a group on ONE XCD - its 32 CUs share
        // the group's rows of both operands through its L2 and every operand byte leaves the fabric once (tiles of a group dealt over the
        // eight XCDs: every XCD fetched all of B and an eighth of A, 9.5 GB per step against 4.2 algorithmic).  Groups differ in length
        // (centre windows are pair-specific more often than border windows): an XCD takes one window of every row and of every column.
        // (second form, round 6: HALF a group - 8 M tiles x all 4 N tiles = 32 tiles, one round of the XCD's CUs - per unit, sixteen units
        //  per XCD: the two halves of a group go to XCDs four apart, B leaves the fabric twice, A once.)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;          // j = 0 .. 511: the XCD's sixteen units of 32 tiles
        const int unit = j >> 5, t = j & 31;
        const int half = unit & 1, row = unit >> 1;
        split = row * 8 + ((xcd - row - 4 * half) & 7);
        tm = half * 8 + (t >> 2);
        tn = t & 3;
    } else if (p.xcd_map) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        split = j / 9;
        tm = xcd & 3;
        tn = (j - split * 9) * 2 + (xcd >> 2);
    } else {
        split = blockIdx.x / tiles;
        if (p.xcd_patch) xcd_patch_map(blockIdx.x - split * tiles, p.tiles_m, p.tiles_n, tm, tn);
        else supertile_map(blockIdx.x - split * tiles, p.tiles_m, p.tiles_n, tm, tn);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    const int nk_total = p.K >> 6;
    if (kt_end > nk_total) kt_end = nk_total;
    if (p.goff) { kt_begin = p.goff[split] >> 6; kt_end = p.goff[split + 1] >> 6; }      // grouped: "split" is the group
    const int nk = kt_end - kt_begin;

    // ---- staging: one instruction = one k block (4 rows) x 8 sub-blocks of a half tile; wave w owns k blocks 2w, 2w+1
    const int kr = (lane >> 1) & 3, sb = lane >> 3, c8 = (lane & 1) * 8;
    int voff[4][2];                                  // kind 0 A0, 1 B0, 2 B1, 3 A1: byte offsets inside a K tile
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int kl = (wid * 2 + q) * 4 + kr;
        long arow;
        if constexpr (ACONV) arow = conv_row_base(kl, p.lgS, p.CinA) + (long)((1 << p.lgS) + 3) * p.CinA;
        else arow = (long)kl * p.lda;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            voff[h ? 3 : 0][q] = (int)((arow + m0 + ((sb >> 2) * 8 + h * 4 + (sb & 3)) * 16 + c8) * 2);
            const int col = n0 + ((sb >> 1) * 4 + h * 2 + (sb & 1)) * 16;      // first column of this lane's 16-column sub-block
            long b;
            if constexpr (BMODE == BMODE_CONV) {
                // the tap is a per-lane quantity: with Cin = 128 a 256-column tile spans two taps, and the last tile of
                // N = 9*Cin may reach past tap 8 (those columns read tap 8 again and are never stored)
                const int tap_raw = col / p.Cin;
                const int tap = tap_raw > 8 ? 8 : tap_raw;
                const int ky = tap / 3, kx = tap - 3 * ky;
                b = conv_row_base(kl, p.lgS, p.Cin) + (long)(ky * ((1 << p.lgS) + 2) + kx) * p.Cin + (col - tap_raw * p.Cin);
            } else if constexpr (BMODE == BMODE_GATHER) {
                const int tap_raw = col / p.Cin;
                const int tap = tap_raw > 8 ? 8 : tap_raw;
                const int ky = tap / 3, kx = tap - 3 * ky;
                // pixel kr of the window + tap + column; the window's own offset is added per K tile (stage)
                b = (long)(((kr >> 1) + ky) * 18 + (kr & 1) + kx) * p.Cin + (col - tap_raw * p.Cin);
            } else if constexpr (BMODE == BMODE_PATCH) {
                const int tap_raw = col / p.Cin;
                const int tap = tap_raw > 8 ? 8 : tap_raw;
                const int ky = tap / 3, kx = tap - 3 * ky;
                b = (long)((kl >> 2) * 16 + ((kr >> 1) + ky) * 4 + (kr & 1) + kx) * p.Cin + (col - tap_raw * p.Cin);
            } else {
                b = (long)kl * p.ldb + col;
            }
            voff[1 + h][q] = (int)((b + c8) * 2);
        }
    }
    auto stage = [&](int kind, int it) __attribute__((always_inline)) {      // it = K tile index inside this split
        char* base = smem + (((it & 1) << 2) + kind) * HT + wid * 2048;
        const int kt = kt_begin + it;
        long toff;
        const u16* g;
        if (kind == 0 || kind == 3) {
            if constexpr (ACONV) toff = conv_row_base(kt * 64, p.lgS, p.CinA); else toff = (long)kt * 64 * p.lda;
            g = p.A + toff;                          // wave-uniform: goes into the buffer descriptor
        } else {
            if constexpr (BMODE == BMODE_CONV) toff = conv_row_base(kt * 64, p.lgS, p.Cin);
            else if constexpr (BMODE == BMODE_PATCH) toff = (long)kt * 16 * 16 * p.Cin;        // 16 windows per K tile, 16 patch pixels each
            else toff = (long)kt * 64 * p.ldb;
            g = p.B + toff;                          // (BMODE_GATHER: replaced below by the window's own base)
        }
        if constexpr (BMODE == BMODE_GATHER) {
            if (kind == 1 || kind == 2) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int code = __builtin_amdgcn_readfirstlane(p.gather[__builtin_amdgcn_readfirstlane(kt * 16 + wid * 2 + q)]);
                    const int W = code & 63;
                    const u16* gq = p.B + ((long)((code >> 6) * 18 + 2 * (W >> 3)) * 18 + 2 * (W & 7)) * p.Cin;
                    buf_load_lds16(gq, voff[kind][q], 0, base + q * 1024);
                }
                return;
            }
        }
        buf_load_lds16(g, voff[kind][0], 0, base);
        buf_load_lds16(g, voff[kind][1], 0, base + 1024);
    };

    // ---- fragment reads: sub-block (wr*4 + i*2 + g) of an A half, (wc*2 + g) of a B half; k blocks ks*4 + kh*2 (+1)
    const int g = (lane >> 4) & 1, kh = lane >> 5, t16 = lane & 15;
    const int lane_off = (t16 >> 2) * 32 + (t16 & 3) * 8 + kh * 2048;
    const int a_rd = (wr * 4 + g) * 128 + lane_off, b_rd = (wc * 2 + g) * 128 + lane_off;
    s16x8 af[2][4], bf[2][4];
    auto tr2 = [&](const char* ptr) __attribute__((always_inline)) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr + 1024));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_a = [&](int h, int par) __attribute__((always_inline)) {
        const char* base = smem + ((par << 2) + (h ? 3 : 0)) * HT + a_rd;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[i][ks] = tr2(base + i * 256 + ks * 4096);
    };
    auto read_b = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const char* base = smem + ((par << 2) + 1 + h) * HT + b_rd;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bf[h][ks] = tr2(base + ks * 4096);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto half = [&](int a) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[2 * a + i][j] = mfma32<ELEM>(af[i][ks], bf[j][ks], acc[2 * a + i][j]);
        SGC_PP_BARRIER();
    };

    if (nk > 0) {
        stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
        if (nk > 1) { stage(0, 1); stage(1, 1); stage(2, 1); SGC_WAIT_VM(8); } else SGC_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    auto tile = [&](int it, int par, auto steady_c) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        read_a(0, par); read_b(par);
        if (it + 1 < nk) stage(3, it + 1);
        SGC_PP_CLOSE(STEADY);
        half(0);
        read_a(1, par);
        if (STEADY) { stage(0, it + 2); stage(1, it + 2); stage(2, it + 2); }
        SGC_PP_CLOSE(STEADY);
        half(1);
    };
    int it = 0;
#pragma unroll 1
    for (; it + 2 < nk; ++it) tile(it, it & 1, std::true_type{});
#pragma unroll 1
    for (; it < nk; ++it) tile(it, it & 1, std::false_type{});
    if (wr == 0) __builtin_amdgcn_s_barrier();

    float* C = p.C + (p.goff ? (long)split * p.N : (long)split * p.slab_stride);
    const int hh = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + cl;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                C[(long)row * p.ldc + col] = acc[i][j][r];
            }
        }
}

template <int ELEM, int BMODE, int ACONV>
static int launch_gemm_tn_pp(TnParams p, int splits, int* slabs_out, hipStream_t stream) {
    constexpr int LDS = 8 * 16384;
    p.tiles_m = p.M / 256;
    p.tiles_n = (p.N + 255) / 256;
    const int nk = p.K >> 6;
    if (splits < 1) splits = 1;
    if (splits > nk) splits = nk;
    p.ktiles_per_split = (nk + splits - 1) / splits;
    splits = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
    auto kern = gemm_tn_pp_kernel<ELEM, BMODE, ACONV>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    p.splits = splits;
    {
        const int xm = sgc_tuning().tn_xcd;
        p.xcd_map = (xm && BMODE == BMODE_CONV && p.tiles_m == 4 && p.tiles_n == 18) ? 1 : 0;
        const int xp = sgc_tuning().tn_patch;    // 0: 16x16 super-tiles instead of per-XCD 4x8 patches
        // per-XCD 4(M) x 8(N) patches need M tiles to share: with 4 M tiles (weight gradient over the pair-specific windows, 4 x 18
        // tiles x 7 splits) a patch is a whole tile column block and the 9 tiles an XCD gets per split straddle two of them - the
        // 16 x 16 super-tile walk measured 7.61 vs 8.48 ms there (alternated twice in one box); grids with >= 8 M tiles keep the patches
        p.xcd_patch = (xp && !p.xcd_map && ((p.tiles_m * p.tiles_n) & 7) == 0 && p.tiles_m >= 8) ? 1 : 0;
    }
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n * splits)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    if (slabs_out) *slabs_out = splits;
    return SGC_OK;
}

// Split-K count for a tile grid when the caller passes splits <= 0: the smallest count that fills the 256 CUs in whole
// waves of blocks to >= 95 % while leaving >= 8 K tiles per block.  Few, long splits matter: every split writes a full f32
// slab with 4-byte stores and adds a slab to the reduction (conv3 weight gradient: 32 splits 62.1 ms, 7 splits 57.5 ms).
static inline int tn_auto_splits(int tiles, int nk) {
    int best = 1;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && nk / s < 8) break;
        const long blocks = (long)tiles * s;
        const long waves = (blocks + 255) / 256;
        best = s;
        if (blocks >= 256 && blocks * 100 >= waves * 256 * 95) break;
    }
    return best;
}

template <int ELEM, int BMODE, int ACONV = 0>
static int launch_gemm_tn(TnParams p, int splits, int* slabs_out, hipStream_t stream) {
    if ((p.K & 63) || (p.N & 127) || (p.M & 127) || p.K <= 0) return SGC_ERR_ARG;
    if (BMODE == BMODE_CONV && ((p.Cin & 127) || p.N != 9 * p.Cin)) return SGC_ERR_ARG;
    const int cfg = sgc_gemm_cfg();
    const bool big_ok = (p.N % 256) == 0 && (p.M % 256) == 0 && (BMODE != BMODE_CONV || (p.Cin % 256) == 0);
    const bool big = big_ok && (cfg == 2 || (cfg == 0 && (long)p.M * p.N >= 512L * 512));
    // the ping-pong block takes per-lane taps: conv weight gradients with Cin = 128 (N = 1152 -> 5 column tiles, the last half empty)
    const bool pp_ok = (p.M % 256) == 0 && (BMODE == BMODE_CONV ? (p.Cin % 128) == 0 : (p.N % 256) == 0);
    if (pp_ok && (cfg == 5 || cfg == 7 || (cfg == 0 && sgc_gemm_pp() && (long)p.M * p.N >= 256L * 1024))) {
        if (splits <= 0) splits = tn_auto_splits((p.M / 256) * ((p.N + 255) / 256), p.K >> 6);
        return launch_gemm_tn_pp<ELEM, BMODE, ACONV>(p, splits, slabs_out, stream);
    }
    if (splits <= 0) splits = tn_auto_splits(big ? (p.M / 256) * (p.N / 256) : (p.M / 128) * (p.N / 128), p.K >> 6);
    if (big) return launch_gemm_tn_cfg<ELEM, BMODE, ACONV, 2, 4, 4, 2>(p, splits, slabs_out, stream);
    return launch_gemm_tn_cfg<ELEM, BMODE, ACONV, 2, 2, 2, 2>(p, splits, slabs_out, stream);
}
