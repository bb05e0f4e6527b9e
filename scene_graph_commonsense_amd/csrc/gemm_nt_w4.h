// EXPERIMENT (tools/gemm_microbench.py only): 256x256x64 NT block with FOUR wavefronts, one per SIMD, each owning a 128x128
// output tile (4x4 MFMA tiles, 256 accumulator registers of the 512-entry file).  LDS fragment reads per MFMA fall by a third
// against the 8-wave block (8 reads per 16 MFMAs instead of 6 per 8) and there is one barrier per K tile; the price is that a
// single wave has to keep its own MFMA pipe fed, i.e. the reads and loads of the next step must be interleaved between the MFMAs
// of the current one (sched_group_barrier), with register double buffering of the fragments.
// RESULT (32768x4096x8192 bf16, random operands, one box, correct to the same 3.1e-3 as the other blocks): compiler schedule
// 1.92 ms (1143 TFLOP/s), 2 MFMA : 1 read interleave 1.73 ms (1274), against 1.48 ms (1486) for the shipped 8-wave ping-pong
// block in the same run - a single wave per SIMD written in HIP does not keep the MFMA pipe as busy as two alternating waves
// do.  Not used by the product path.
#pragma once

template <int ELEM, int EPI, int SCHED>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 65536;                  // per K tile: A [256][128 B] then B [256][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    int tm, tn;
    xcd_patch_map(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    // ---- staging: instruction j = wid + 4q of the block covers rows 8j..8j+7 of A (q < 8) or B (q >= 8)
    const int lrow = lane >> 3, cpos = lane & 7;
    const int r0 = wid * 8 + lrow;                                   // row for q = 0; row(q) = r0 + 32 q
    const int chunk = (cpos ^ ((r0 >> 1) & 7)) << 3;
    const u16* const a_blk = p.A + (long)m0 * p.lda;
    const u16* const b_blk = p.B + (long)n0 * p.ldb;
    int va[8], vb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int m = r0 + 32 * q;
        if (m0 + m > p.M - 1) m = p.M - 1 - m0;
        va[q] = (int)((m * p.lda + chunk) * 2);
        vb[q] = (int)(((r0 + 32 * q) * p.ldb + chunk) * 2);
    }
    auto stage = [&](int buf_tile, int k_tile, int q) __attribute__((always_inline)) {          // q in 0..15
        char* base = smem + (buf_tile & 1) * TILE + (q >= 8 ? 32768 : 0) + (wid + 4 * (q & 7)) * 1024;
        if (q < 8) buf_load_lds16(a_blk, va[q], k_tile << 7, base);
        else buf_load_lds16(b_blk, vb[q - 8], k_tile << 7, base);
    };

    const int l31 = lane & 31, kh = lane >> 5, sw = (l31 >> 1) & 7;
    int ko[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ko[ks] = ((ks * 2 + kh) ^ sw) << 4;
    const int a_rd = (wr * 128 + l31) * 128, b_rd = 32768 + (wc * 128 + l31) * 128;
    s16x8 af[2][4], bf[2][4];
    auto read_frags = [&](int buf, int t, int ks) __attribute__((always_inline)) {
        const char* base = smem + (t & 1) * TILE + ko[ks];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[buf][i] = *reinterpret_cast<const s16x8*>(base + a_rd + i * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[buf][i] = *reinterpret_cast<const s16x8*>(base + b_rd + i * 4096);
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfmas = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma32<ELEM>(af[buf][i], bf[buf][j], acc[i][j]);
    };
    auto interleave = [&](int n_vmem) __attribute__((always_inline)) {
        if (SCHED == 1) {
            // 16 MFMAs with the 8 LDS reads (and, in the loading steps, the 8 global->LDS loads AFTER them: a load may not move
            // above an LDS read of the same address space) threaded between them
            if (n_vmem == 0) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        }
    };

    const int nk = p.K >> 6;
    // prologue: tile 0 and the first half (A) of tile 1.  Inside the loop every load and read is UNCONDITIONAL (tile indices
    // are clamped to nk-1: the tail re-loads the last tile into a buffer nobody reads) so that each step is one basic block
    // and the interleave below applies.
    const int last = nk - 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) stage(0, 0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) stage(1, min(1, last), q);
    SGC_WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);
    for (int t = 0; t < nk; ++t) {
        const int k1 = min(t + 1, last), k2 = min(t + 2, last);
        // ks = 0: reads for ks 1; second half (B) of the loads of tile t+1
        read_frags(1, t, 1);
#pragma unroll
        for (int q = 8; q < 16; ++q) stage(t + 1, k1, q);
        mfmas(0);
        interleave(8);
        __builtin_amdgcn_sched_barrier(0);
        // ks = 1
        read_frags(0, t, 2);
        mfmas(1);
        interleave(0);
        __builtin_amdgcn_sched_barrier(0);
        // ks = 2: reads for ks 3, then the tile barrier: tile t+1 has landed, tile t is read completely
        read_frags(1, t, 3);
        mfmas(0);
        interleave(0);
        __builtin_amdgcn_sched_barrier(0);
        SGC_WAIT_VM(0);
        SGC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ks = 3: reads for (t+1, 0); first half (A) of the loads of tile t+2 into the buffer tile t just left
        read_frags(0, t + 1, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) stage(t + 2, k2, q);
        mfmas(1);
        interleave(8);
        __builtin_amdgcn_sched_barrier(0);
    }
    SGC_WAIT_VM(0);
    __syncthreads();
    nt_epilogue<ELEM, EPI, 4, 4>(p, acc, m0, n0, wr, wc, lane);
}

template <int ELEM, int EPI, int SCHED>
static int launch_gemm_nt_w4(NtParams p, hipStream_t stream) {
    constexpr int LDS = 2 * 65536;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 256;
    auto kern = gemm_nt_w4_kernel<ELEM, EPI, SCHED>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
