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

enum { BMODE_PLAIN = 0, BMODE_CONV = 1 };

struct TnParams {
    const u16* A; const u16* B; float* C;
    int M, N, K;                 // K multiple of 64 (zero rows pad it), M,N multiples of 128
    long lda, ldb, ldc;
    long slab_stride;            // elements between split-K slabs
    int lgS, Cin;                // BMODE_CONV
    int CinA;                    // ACONV: A rows are the centre pixels of a zero-padded image with CinA channels
    int tiles_m, tiles_n, ktiles_per_split;
};

template <int ELEM, int BMODE, int ACONV>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_BYTES = 64 * 128 * 2;   // 16 KiB per operand per buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = blockIdx.x % p.tiles_n, tm = blockIdx.x / p.tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int kt_begin = blockIdx.y * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    const int nk_total = p.K >> 6;
    if (kt_end > nk_total) kt_end = nk_total;

    // loader: instruction i of wave w fills k-block kb = w*4+i (4 k rows) x 128 columns
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

    auto stage = [&](int buf, int kt) {
        char* abase = smem + buf * 2 * TILE_BYTES + wid * 4096;
        char* bbase = abase + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kt * 64 + (wid * 4 + i) * 4 + kr;
            const u16* ap;
            if constexpr (ACONV) ap = p.A + conv_row_base(k, p.lgS, p.CinA) + (long)((1 << p.lgS) + 3) * p.CinA + m0 + mo;
            else ap = p.A + (long)k * p.lda + m0 + mo;
            const u16* bp;
            if constexpr (BMODE == BMODE_CONV) bp = p.B + conv_row_base(k, p.lgS, p.Cin) + boff_tap + bcol0 + mo;
            else bp = p.B + (long)k * p.ldb + n0 + mo;
            __builtin_amdgcn_global_load_lds(GLB_PTR(ap), LDS_PTR(abase + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(bp), LDS_PTR(bbase + i * 1024), 16, 0, 0);
        }
    };

    const int wr = wid >> 1, wc = wid & 1;
    const int g = (lane >> 4) & 1, kh = lane >> 5, t = lane & 15;
    const int lane_off = (t >> 2) * 32 + (t & 3) * 8;
    int a_blk[2], b_blk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_blk[i] = (wr * 4 + i * 2 + g) * 128 + lane_off;
        b_blk[i] = (wc * 4 + i * 2 + g) * 128 + lane_off;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) stage(0, kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int it = kt - kt_begin;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < kt_end) stage((it + 1) & 1, kt + 1);
        const char* ab = smem + (it & 1) * 2 * TILE_BYTES;
        const char* bb = ab + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kb = ks * 4 + kh * 2;          // first of two 4-row k blocks
            s16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(ab + kb * 1024 + a_blk[i]));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(ab + (kb + 1) * 1024 + a_blk[i]));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(bb + kb * 1024 + b_blk[i]));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(bb + (kb + 1) * 1024 + b_blk[i]));
                af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32<ELEM>(af[i], bf[j], acc[i][j]);
        }
    }

    float* C = p.C + (long)blockIdx.y * p.slab_stride;
    const int h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + cl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[(long)row * p.ldc + col] = acc[i][j][r];
            }
        }
}

template <int ELEM, int BMODE, int ACONV = 0>
static int launch_gemm_tn(TnParams p, int splits, int* slabs_out, hipStream_t stream) {
    if ((p.K & 63) || (p.N & 127) || (p.M & 127) || p.K <= 0) return SGC_ERR_ARG;
    if (BMODE == BMODE_CONV && ((p.Cin & 127) || p.N != 9 * p.Cin)) return SGC_ERR_ARG;
    p.tiles_m = p.M / 128;
    p.tiles_n = p.N / 128;
    const int nk = p.K >> 6;
    if (splits < 1) splits = 1;
    if (splits > nk) splits = nk;
    p.ktiles_per_split = (nk + splits - 1) / splits;
    splits = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
    static bool attr_set = false;
    auto kern = gemm_tn_kernel<ELEM, BMODE, ACONV>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n), (unsigned)splits), dim3(256), 65536, stream, p);
    SGC_CHECK_LAUNCH();
    if (slabs_out) *slabs_out = splits;
    return SGC_OK;
}
