// NT GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T, 16-bit inputs (f16 or bf16), f32 accumulate on
// v_mfma_f32_32x32x16_{f16,bf16}.  128x128x64 block tile, 4 wavefronts (2x2), each 64x64 = 2x2 MFMA tiles.
//
//  * Operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered, one
//    barrier per K tile.  The LDS image is lane-linear, so the bank swizzle (16-byte chunk index XOR
//    ((row>>1)&7), conflict-free for the 4x16-lane groups of ds_read_b128 at a 128-byte row pitch) is
//    applied to the per-lane SOURCE address and again on the fragment read.
//  * AMODE_CONV turns the A operand into an implicit 3x3 convolution over a zero-padded channels-last
//    image [img][S+2][S+2][Cin]: K index = (c/64, tap, c%64); every A row is a pixel whose address is shifted by
//    the tap, so no im2col buffer exists.  Rows are enumerated window-major (m = 4*window + q,
//    q = dy*2+dx inside a 2x2 pooling window) so that a 2x2 max-pool is a pure in-register max over the
//    four accumulator registers a lane holds for one window (MFMA C layout: row = (reg&3) + 8*(reg>>2)
//    + 4*(lane>>5)).
//  * Epilogues fuse what the reference does after each contraction (bias, tanh, ReLU, dropout, 2x2
//    max-pool + argmax, one-hot label columns of fc2 as a per-object row gather, ReLU-mask for dgrad).
#pragma once
#include "common.h"

enum { AMODE_PLAIN = 0, AMODE_CONV = 1 };
enum { EPI_STORE = 0, EPI_BIAS_TANH = 1, EPI_BIAS_RELU = 2, EPI_POOL = 3, EPI_FC2 = 4, EPI_RELUMASK = 5 };

struct NtParams {
    const u16* A; const u16* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int lgS, Cin;               // AMODE_CONV: image side S = 1<<lgS, channels per tap
    const float* bias;          // [N] or nullptr
    const u16* mask_src;        // EPI_RELUMASK: forward activation [M][ldc], gradient passes where it is > 0
    const float* lsub; const float* lobj; const int* sub_idx; const int* obj_idx;   // EPI_FC2
    unsigned char* argmax;      // EPI_POOL (optional)
    float scale;                // EPI_RELUMASK / dropout scale
    unsigned drop_seed; int drop_enable;
    int tiles_m, tiles_n;
};

template <int ELEM>
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
    if constexpr (ELEM == ELEM_F16) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

// window-major row -> element offset of padded pixel (tap 0,0) in [img][S+2][S+2][Cin]
__device__ __forceinline__ long conv_row_base(int m, int lgS, int Cin) {
    const int S = 1 << lgS;
    const int img = m >> (2 * lgS);
    const int ml = m & (S * S - 1);
    const int W = ml >> 2, q = ml & 3;
    const int py = W >> (lgS - 1), px = W & ((S >> 1) - 1);
    const int y = 2 * py + (q >> 1), x = 2 * px + (q & 1);
    return ((long)(img * (S + 2) + y) * (S + 2) + x) * Cin;
}

// Block configuration: WR x WC wavefronts, each owning a (TM*32) x (TN*32) output tile.
//   small: 2x2 waves of 64x64   -> 128x128 block, 64 KiB LDS, 2 blocks/CU  (small problems, N % 256 != 0)
//   big  : 2x4 waves of 128x64  -> 256x256 block, 128 KiB LDS, 1 block/CU  (half the LDS bytes per MFMA)
template <int ELEM, int AMODE, int EPI, int WR, int WC, int TM, int TN>
__global__ __launch_bounds__(WR * WC * 64, 2) void gemm_nt_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WR * WC;
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int AI = BM / (8 * NW), BI = BN / (8 * NW);      // global_load_lds instructions per wave per K tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = blockIdx.x % p.tiles_n, tm = blockIdx.x / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- loader addresses: wave w stages rows w*AI*8 .. of A and w*BI*8 .. of B, 8 rows (1 KiB) per instruction
    const int lrow = lane >> 3, cpos = lane & 7;
    const u16* a_ptr[AI];
    const u16* b_ptr[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = wid * AI * 8 + i * 8 + lrow;
        const int chunk = cpos ^ ((row >> 1) & 7);
        int m = m0 + row; if (m > p.M - 1) m = p.M - 1;
        if constexpr (AMODE == AMODE_CONV) a_ptr[i] = p.A + conv_row_base(m, p.lgS, p.Cin) + chunk * 8;
        else a_ptr[i] = p.A + (long)m * p.lda + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = wid * BI * 8 + i * 8 + lrow;
        const int chunk = cpos ^ ((row >> 1) & 7);
        b_ptr[i] = p.B + (long)(n0 + row) * p.ldb + chunk * 8;
    }
    const int Wp = (1 << p.lgS) + 2;

    auto stage = [&](int buf, int kt) {
        long aoff;
        if constexpr (AMODE == AMODE_CONV) {
            // K order = (64-channel chunk, tap, channel): the nine taps of one chunk are consecutive K tiles, so
            // the shifted re-reads of a pixel row hit L2 (working set 18x18x64 ch per block) instead of the fabric.
            const int cc = kt / 9, tap = kt - cc * 9;
            const int ky = tap / 3, kx = tap - 3 * ky;
            aoff = (long)(ky * Wp + kx) * p.Cin + (cc << 6);
        } else {
            aoff = (long)kt << 6;
        }
        const long boff = (long)kt << 6;
        char* abase = smem + buf * BUF_BYTES + wid * (AI * 1024);
        char* bbase = smem + buf * BUF_BYTES + A_BYTES + wid * (BI * 1024);
#pragma unroll
        for (int i = 0; i < AI; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_ptr[i] + aoff), LDS_PTR(abase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BI; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_ptr[i] + boff), LDS_PTR(bbase + i * 1024), 16, 0, 0);
    };

    // ---- fragment read addresses
    const int wr = wid / WC, wc = wid % WC;
    const int kh = lane >> 5;
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ra = wr * TM * 32 + i * 32 + (lane & 31);
        a_off[i] = ra * 128; a_sw[i] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int rb = wc * TN * 32 + i * 32 + (lane & 31);
        b_off[i] = rb * 128; b_sw[i] = (rb >> 1) & 7;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K >> 6;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* ab = smem + (kt & 1) * BUF_BYTES;
        const char* bb = ab + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks * 2 + kh;
            s16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const s16x8*>(ab + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int i = 0; i < TN; ++i) bf[i] = *reinterpret_cast<const s16x8*>(bb + b_off[i] + ((c ^ b_sw[i]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<ELEM>(af[i], bf[j], acc[i][j]);
        }
    }

    // ---- epilogue
    const int h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * TN * 32 + j * 32 + cl;
            const int rbase = m0 + wr * TM * 32 + i * 32;
            const float bias = p.bias ? p.bias[col] : 0.f;
            if constexpr (EPI == EPI_POOL) {
                u16* out = reinterpret_cast<u16*>(p.C);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    float v = acc[i][j][4 * w];
                    int am = 0;
#pragma unroll
                    for (int q = 1; q < 4; ++q) {
                        const float t = acc[i][j][4 * w + q];
                        if (t > v) { v = t; am = q; }
                    }
                    v += bias;
                    const int prow = (rbase >> 2) + 2 * w + h;
                    if (prow * 4 < p.M) {
                        if (!(v > 0.f)) { v = 0.f; am = 4; }        // ReLU killed: no gradient path
                        out[(long)prow * p.ldc + col] = to_elem<ELEM>(v);
                        if (p.argmax) p.argmax[(long)prow * p.ldc + col] = (unsigned char)am;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row >= p.M) continue;
                    float v = acc[i][j][r];
                    const long o = (long)row * p.ldc + col;
                    if constexpr (EPI == EPI_STORE) {
                        reinterpret_cast<u16*>(p.C)[o] = to_elem<ELEM>(v + bias);
                    } else if constexpr (EPI == EPI_BIAS_TANH) {
                        reinterpret_cast<u16*>(p.C)[o] = to_elem<ELEM>(tanhf(v + bias));
                    } else if constexpr (EPI == EPI_BIAS_RELU) {
                        v = fmaxf(v + bias, 0.f);
                        if (p.drop_enable) v = dropout_keep(p.drop_seed, (uint32_t)(row * p.N + col)) ? v * p.scale : 0.f;
                        reinterpret_cast<u16*>(p.C)[o] = to_elem<ELEM>(v);
                    } else if constexpr (EPI == EPI_FC2) {
                        v += bias + p.lsub[(long)p.sub_idx[row] * p.N + col] + p.lobj[(long)p.obj_idx[row] * p.N + col];
                        v = fmaxf(v, 0.f);
                        if (p.drop_enable) v = dropout_keep(p.drop_seed, (uint32_t)(row * p.N + col)) ? v * p.scale : 0.f;
                        reinterpret_cast<float*>(p.C)[o] = v;
                    } else if constexpr (EPI == EPI_RELUMASK) {
                        const float f = from_elem<ELEM_F16>(p.mask_src[o]);   // forward activations are f16
                        reinterpret_cast<u16*>(p.C)[o] = to_elem<ELEM>(f > 0.f ? v * p.scale : 0.f);
                    }
                }
            }
        }
    }
}

inline int sgc_gemm_cfg() {       // test hook: SGC_GEMM_CFG=1 forces the 128x128 block, =2 the 256x256 block
    static int cfg = -1;
    if (cfg < 0) { const char* e = getenv("SGC_GEMM_CFG"); cfg = e ? atoi(e) : 0; }
    return cfg;
}

template <int ELEM, int AMODE, int EPI, int WR, int WC, int TM, int TN>
static int launch_gemm_nt_cfg(NtParams p, hipStream_t stream) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int LDS = 2 * (BM + BN) * 128;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.N / BN;
    static bool attr_set = false;
    auto kern = gemm_nt_kernel<ELEM, AMODE, EPI, WR, WC, TM, TN>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(WR * WC * 64), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

template <int ELEM, int AMODE, int EPI>
static int launch_gemm_nt(NtParams p, hipStream_t stream) {
    if (p.M <= 0) return SGC_OK;
    if ((p.K & 63) || (p.N & 127) || p.K <= 0) return SGC_ERR_ARG;
    if (AMODE == AMODE_CONV && ((p.Cin & 63) || p.K != 9 * p.Cin)) return SGC_ERR_ARG;
    const int cfg = sgc_gemm_cfg();
    const bool big_ok = (p.N % 256) == 0;
    const bool big = big_ok && (cfg == 2 || (cfg == 0 && (long)p.M * p.N >= 256L * 256 * 256));
    if (big) return launch_gemm_nt_cfg<ELEM, AMODE, EPI, 2, 4, 4, 2>(p, stream);
    return launch_gemm_nt_cfg<ELEM, AMODE, EPI, 2, 2, 2, 2>(p, stream);
}
