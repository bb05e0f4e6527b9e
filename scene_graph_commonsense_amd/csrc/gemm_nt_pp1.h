// EXPERIMENT (tools/gemm_microbench.py only): ping-pong block with ONE phase per K tile - 32 MFMAs per barrier slot instead
// of 16 (two slots per K tile instead of four; a slot costs ~100 cycles beyond its MFMAs).  Both wave rows read all their
// fragments of a tile in one load section (24 ds_read_b128, 96 fragment registers); two whole-tile LDS buffers.  Tile T's
// loads may start once wave row 1 has read tile T-2 (slot 2T-3) and must have landed before wave row 0 reads tile T (slot 2T):
// wave row 0 issues its share at the start of its load section L(T-1) and waits at the end of M(T-1); wave row 1 issues its
// share interleaved with the first MFMAs of M(T-2) and waits at the end of L(T-1).
// RESULT (32768x4096x8192 and x16384 bf16, random operands, one box, same 2.2e-3 error): 1.83 / 3.93 ms (1199 / 1118 TFLOP/s)
// against 1.59 / 3.15 ms (1383 / 1397) for the shipped two-phase block in the same run: with only two whole-tile buffers every
// load has to be waited for with vmcnt(0) one slot after it was issued, and that costs more than the two barrier slots save.
// Not used by the product path.
#pragma once

template <int ELEM, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp1_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 65536;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    int tm, tn;
    xcd_patch_map(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    // ---- staging: instruction (w + 8q), q = 0..3, covers rows 8(w+8q) .. +7 of A and of B
    const int lrow = lane >> 3, cpos = lane & 7;
    const int r0 = wid * 8 + lrow;
    const int chunk = (cpos ^ ((r0 >> 1) & 7)) << 3;
    const u16* const a_blk = p.A + (long)m0 * p.lda;
    const u16* const b_blk = p.B + (long)n0 * p.ldb;
    int va[4], vb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int m = r0 + 64 * q;
        if (m0 + m > p.M - 1) m = p.M - 1 - m0;
        va[q] = (int)((m * p.lda + chunk) * 2);
        vb[q] = (int)(((r0 + 64 * q) * p.ldb + chunk) * 2);
    }
    auto stage = [&](int t) __attribute__((always_inline)) {
        char* base = smem + (t & 1) * TILE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            buf_load_lds16(a_blk, va[q], t << 7, base + q * 8192);
            buf_load_lds16(b_blk, vb[q], t << 7, base + 32768 + q * 8192);
        }
    };
    const int l31 = lane & 31, kh = lane >> 5, sw = (l31 >> 1) & 7;
    int ko[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ko[ks] = ((ks * 2 + kh) ^ sw) << 4;
    const int a_rd = (wr * 128 + l31) * 128, b_rd = 32768 + (wc * 64 + l31) * 128;
    s16x8 af[4][4], bf[2][4];
    auto read_all = [&](int t) __attribute__((always_inline)) {
        const char* base = smem + (t & 1) * TILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][ks] = *reinterpret_cast<const s16x8*>(base + a_rd + i * 4096 + ko[ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j][ks] = *reinterpret_cast<const s16x8*>(base + b_rd + j * 4096 + ko[ks]);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32<ELEM>(af[i][ks], bf[j][ks], acc[i][j]);
    };

    const int nk = p.K >> 6;
    stage(0);
    if (nk > 1) { stage(1); SGC_WAIT_VM(8); } else SGC_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    if (wr == 0) {
#pragma unroll 1
        for (int t = 0; t < nk; ++t) {
            if (t >= 1 && t + 1 < nk) stage(t + 1);             // L(t): tile t+1 first, then every fragment of tile t
            read_all(t);
            SGC_WAIT_LGKM0();
            SGC_PP_BARRIER();
            mfmas();                                            // M(t)
            __builtin_amdgcn_sched_barrier(0);
            SGC_WAIT_VM(0);
            SGC_PP_BARRIER();
        }
        __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll 1
        for (int t = 0; t < nk; ++t) {
            read_all(t);                                        // L(t)
            SGC_WAIT_VM(0);
            SGC_WAIT_LGKM0();
            SGC_PP_BARRIER();
            if (t + 2 < nk) stage(t + 2);                       // M(t): tile t+2 threaded between the first MFMAs
            mfmas();
            if (t + 2 < nk) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
            SGC_PP_BARRIER();
        }
    }
    if constexpr (EPI == EPI_STORE) {
        if (p.epi_lds) { nt_epilogue_store16<ELEM>(p, acc, m0, n0, wr, wc, lane, wid, smem); return; }
    }
    nt_epilogue<ELEM, EPI, 4, 2>(p, acc, m0, n0, wr, wc, lane);
}

template <int ELEM, int EPI>
static int launch_gemm_nt_pp1(NtParams p, hipStream_t stream) {
    constexpr int LDS = (EPI == EPI_STORE) ? EPI_LDS_BYTES : 2 * 65536;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 256;
    auto kern = gemm_nt_pp1_kernel<ELEM, EPI>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
