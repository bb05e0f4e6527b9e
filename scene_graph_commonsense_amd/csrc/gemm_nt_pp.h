// Ping-pong scheduled 256x256x64 NT GEMM blocks for gfx950 (included by gemm_nt.h): plain GEMM and the halo-staged 3x3
// convolution on 16x16 maps.  Same tile geometry as the 8-wave block of gemm_nt_kernel (2x4 waves, 128x64 per wave) but a
// different schedule:
//  * The two waves that share a SIMD (w and w+4, i.e. the two wave ROWS wr = 0/1) run half a phase apart: while one
//    issues its LDS fragment reads and its share of the global->LDS prefetch, the other issues MFMAs, and they swap at
//    every s_barrier.  The 2-stage loop of gemm_nt_kernel leaves both waves of a SIMD reading LDS at the same time after
//    each barrier (LDS-read + MFMA alone: 1450 TFLOP/s-equivalent there, 1555 here; tools/gemm_microbench.py).
//  * A K tile is consumed in two phases = the two 64x64 halves of the wave's 128x64 output, 16 MFMAs per barrier slot:
//    phase X (rows a0) reads the A0 half and both B halves (16 ds_read_b128), phase Y (rows a1) reads A1 (8).
//  * Operands are staged in HALF tiles (128 rows x 64 k = 16 KiB = 2 global_load_lds per wave): A0/A1 = rows
//    {0-63,128-191}/{64-127,192-255} of the block (the a-half of both wave rows), B0/B1 = the b-half of the four wave
//    columns.  Eight 16 KiB slots (two K tiles) form a ring; a half tile's successor (two K tiles ahead) is issued in
//    the phase after its last read, and every phase ends with a COUNTED s_waitcnt vmcnt(8): up to 64 KiB per CU stay
//    in flight across the barriers, every half tile has two phases (~1.2 us) to land; vmcnt(0) only in the last two tiles.
//  Hazards (phase p of wave row 0 occupies barrier slots 2p [loads] and 2p+1 [MFMA]; wave row 1 is one slot later):
//   RAW  a half tile read in phase p+1 is waited for (vmcnt) in the load section of phase p by BOTH wave rows, i.e.
//        before the barriers ending slots 2p and 2p+1; the first read is in slot 2p+2.
//   WAR  the reads of phase p are retired (lgkmcnt(0)) before the barrier that ends their slot (2p or 2p+1); the slot's
//        overwrite is issued in phase p+1 (slot >= 2p+2).
//  Measured on 32768x4096x8192 bf16, random operands, one box: 2-stage loop 1.91 ms (1150 TFLOP/s), this schedule
//  1.77 ms (1243), with the XCD-aware tile walk 1.74 ms (1264).  Variants measured and dropped: four phases per K tile
//  (8 MFMAs per slot, vmcnt(12)) 2.0 ms - the ~85-cycle barrier slot overhead is paid twice as often; s_setprio(1)
//  around the MFMA section -1 %; issuing the last 1/2/4 MFMAs of a slot AFTER its closing barrier (to keep the pipe fed
//  across the barrier) -7/-9/-11 %.  Row strides: padding the 128 KiB rows of the fc1 operands (K = 65536) by 128 B ... 1152 B
//  moved the launch by <= 1 % (13.71 -> 13.55 ms): the power-of-two stride is not what costs the L2 its hits.  Counters for
//  that launch: 56 GB of L2 misses against 25 GB for perfectly shared 4x8 patches (blocks of a patch drift apart in K) and an
//  effective clock of 1.40 GHz (conv3 data gradient: 1.90) - the chip's power budget, not the schedule, sets these rates.
#pragma once

#define SGC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define SGC_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SGC_PP_BARRIER()                                       \
    do {                                                       \
        __builtin_amdgcn_sched_barrier(0);                     \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

// global -> LDS copy of 16 B per lane through a buffer resource: wave-uniform base (SGPR descriptor) + 32-bit per-lane byte
// offset (+ SGPR byte offset).  With global_load_lds every lane sends a 64-bit address; the buffer form was measured
// +10 % on the whole GEMM (1272 -> 1400 TFLOP/s, 32768x4096x8192 bf16): the VMEM issue cost was the larger part of the
// interference between the load stream and the MFMA loop.  Offsets stay far below the 2 GiB record limit set here.
__device__ __forceinline__ void buf_load_lds16(const void* base, int voff, int soff, void* lds) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds), 16, voff, soff, 0, 0);
}

// XCD-aware tile walk: block id -> XCD id%8 (hardware round-robin); each XCD walks a contiguous range of the tile
// sequence in 4(M) x 8(N) patches, so the 32 blocks resident on one XCD share 4 A panels and 8 B panels through its L2.
__device__ __forceinline__ void xcd_patch_map(int id, int tiles_m, int tiles_n, int& tm, int& tn, int gm = 4, int gn = 8) {
    const int nb = tiles_m * tiles_n, q = nb >> 3, r = nb & 7, x = id & 7;
    const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    supertile_map_g(lin, tiles_m, tiles_n, gm, gn, tm, tn);
}

// Patch-aligned form used by the NT block: the tile grid is cut into GM x GN = 32-tile patches (4 x 8; narrower grids 8x4, 16x2,
// 32x1), patch k goes to XCD k % 8 and the grid is padded to whole patches, so the 32 blocks resident on an XCD are always ONE
// patch (with the unpadded walk above an XCD's range starts at a multiple of nb/8, not of 32: fc1 forward, 126 x 16 tiles, ran
// 2.6 % slower than the 128 x 16 grid that does MORE work).  Edge patches come last on every XCD; blocks beyond the grid exit.
// Used for grids of >= 1024 tiles whose padding is < 10 % and whose per-XCD share is not already a multiple of 32 tiles; smaller
// grids keep the walk above, which spreads them over all XCDs (in-box A/B, SGC_NT_ALIGNED=0/1: fc1 forward 14.0 -> 13.6 ms).
__host__ __device__ __forceinline__ int xcd_patch_gn(int tiles_n) { return tiles_n >= 8 ? 8 : (tiles_n >= 4 ? 4 : (tiles_n >= 2 ? 2 : 1)); }
static inline int xcd_patch_grid(int tiles_m, int tiles_n) {
    const int gn = xcd_patch_gn(tiles_n), gm = 32 / gn;
    const int npatch = ((tiles_m + gm - 1) / gm) * ((tiles_n + gn - 1) / gn);
    return ((npatch + 7) / 8) * 8 * 32;
}
__device__ __forceinline__ bool xcd_patch_map_aligned(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int gn = xcd_patch_gn(tiles_n), gm = 32 / gn;
    const int sn = (tiles_n + gn - 1) / gn, npatch = ((tiles_m + gm - 1) / gm) * sn;
    const int x = id & 7, j = id >> 3;
    const int patch = (j >> 5) * 8 + x, w = j & 31;
    if (patch >= npatch) return false;
    const int a = patch / sn, b = patch - a * sn;
    const int wn = min(gn, tiles_n - b * gn);
    tm = a * gm + w / wn;
    tn = b * gn + w % wn;
    return tm < tiles_m && w < gm * wn;
}

// ABL (tools/gemm_microbench.py only): 0 normal, 1 no global loads inside the K loop.
// ACG = 1: the A operand is the implicit 3x3 convolution over a LIST of 2x2 windows of S x S maps, S = 16 or 32 (AMODE_CONV_GATHER of
// gemm_nt.h: row m = pixel m&3 of window gather[m>>2] = image*(S/2)^2 + window, *gather_n entries, p.M only bounds the launch).  The rows of one
// tile come from a few consecutive images (the list is sorted), so their byte offsets from the tile's first image fit the 32-bit
// per-lane offset of the buffer load; the K step (64-channel chunk, tap) is a wave-uniform byte offset.
// SEG = 1: the data gradient of conv3 over listed windows in PATCH form (csrc/kernels_shared.hip, sgc_windows_dgrad_patches).  Row m is
// a listed window; the virtual N tile tn = (patch pixel pp = 4*py + px of the window's 4x4 input patch, N half): the gradient of that
// patch pixel is the sum over the (own pixel q, tap t) combinations with q + t = pp - 1, 2 or 4 of them - of dy[4m + q][:1024] x
// W_t, i.e. a product with K = 1024 x combinations whose A row is made of SEGMENTS (rows 4m + q_c of dy) and whose B is the matching
// stack of tap matrices (prepared per pp, contiguous).  Against the column form (rows 4m + q, N = 9 x 512, K = 1024) the same
// multiply-adds leave 16 instead of 36 rows of 512 values per window, and the 9-tap sum of col2im happens in the accumulators.
template <int ELEM, bool GATHER>
__device__ __forceinline__ void nt_epilogue_pool16(const NtParams& p, f32x16 (&acc)[4][2], int m0, int n0, int wr, int wc, int lane,
                                                   int wid, char* smem, int m_limit);
template <int ELEM, int EPI, int ABL = 0, int ACG = 0, int SEG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HT = 16384;                        // half-tile bytes; slot = parity*4 + kind, kind 0 A0, 1 B0, 2 B1, 3 A1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
#ifdef SGC_EXPERIMENTS      // tools/fc1_windows_microbench.py: staggered first generation of blocks, per-block wall clocks
    if (p.stagger > 0 && blockIdx.x < 256) {          // equal tiles started together keep every CU's epilogue at the same moment
        const int late = ((blockIdx.x >> 3) & (p.stagger_phases - 1)) * p.stagger;
        for (int s = 0; s < late; ++s) __builtin_amdgcn_s_sleep(127);
    }
    const unsigned long long clk0 = p.clk ? wall_clock64() : 0ULL;
#endif
    int tm, tn;
    if (p.patch_aligned) {
        if (!xcd_patch_map_aligned(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn)) return;  // padding block (uniform exit)
    } else if (SEG && p.patch_gn > 0) {
        xcd_patch_map(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn, 32 / p.patch_gn, p.patch_gn);      // tools/dgrad_patch_microbench.py: patch shape
    } else {
        xcd_patch_map(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
    }
    const int m0 = tm * 256;
    int n0 = tn * 256, n_tile = tn, Kloc = p.K, seg_py = 0, seg_px = 0, seg_ny = 1, seg_nx = 1, seg_c0 = 0;
    long ldb = p.ldb, b_off = 0;
    if constexpr (SEG) {
        // virtual N tile = (slot, channel half).  Slots are the 16 patch pixels pp in natural order; with p.seg_split the four centre pixels
        // (4 combinations, K = 4096) take TWO slots of 2 combinations each, summed by the reader (windows_patch_sum): the blocks of an XCD
        // patch share their A and B tiles through the L2 only while they stay at the same K step, and blocks of 64 K steps drift apart
        // (measured: slots ordered by class, i.e. whole patches of K = 4096 blocks, 10.9 ms per launch against 8.6 in natural order).
        const int slot = tn >> 1;
        int pp = 0, part = 0;
        if (p.seg_split) {
            int acc = 0;
            for (pp = 0; pp < 16; ++pp) {
                const int wdt = (((pp >> 2) == 1 || (pp >> 2) == 2) && ((pp & 3) == 1 || (pp & 3) == 2)) ? 2 : 1;
                if (slot < acc + wdt) { part = slot - acc; break; }
                acc += wdt;
            }
        } else {
            pp = slot;
        }
        n_tile = tn & 1;
        n0 = slot * 512 + n_tile * 256;                // output columns: slot, channel half
        seg_py = pp >> 2; seg_px = pp & 3;
        auto cnt1 = [](int c) { return (c == 0 || c == 3) ? 1 : 2; };
        seg_ny = cnt1(seg_py); seg_nx = cnt1(seg_px);
        const int ncomb = seg_ny * seg_nx;
        seg_c0 = (p.seg_split && ncomb == 4) ? 2 * part : 0;
        Kloc = ((p.seg_split && ncomb == 4) ? 2 : ncomb) * 1024;
        ldb = ncomb * 1024 + p.seg_bpad;               // row of B_pp: all combinations of pp (+ padding)
        for (int j = 0; j < pp; ++j) b_off += 512L * (1024 * (cnt1(j >> 2) * cnt1(j & 3)) + p.seg_bpad);     // B_pp follows B_0 .. B_pp-1
        b_off += seg_c0 * 1024;
    }
    int Mlim = p.M;
    long img0 = 0;
    if constexpr (ACG) {
        Mlim = min(p.M, 4 * *p.gather_n);
        if (m0 >= Mlim) return;                      // the grid is sized for the bound (uniform exit, before any barrier)
        img0 = p.gather[m0 >> 2] >> (2 * p.lgS - 2);       // windows per map: (S/2)^2
    }
    const int SP = (1 << p.lgS) + 2;                  // padded map side (ACG only)
    const long img_elems = (long)SP * SP * p.Cin;

    // ---- staging sources: wave w writes LDS rows (2w+q)*8 .. +7 of every half tile (q = 0,1), 8 lanes per 128-B row
    const int lrow = lane >> 3, cpos = lane & 7;
    const u16* const a_blk = ACG ? p.A + img0 * img_elems : p.A + (long)m0 * p.lda;
    const u16* const b_blk = p.B + b_off + (long)(n_tile * 256) * ldb + (p.tile_group ? (long)p.tile_group[tm] * p.group_stride : 0L);
    int voff[4][2];                                  // byte offsets from a_blk / b_blk
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wid * 2 + q) * 8 + lrow;
        const int chunk = (cpos ^ ((r >> 1) & 7)) << 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = (r >> 6) * 128 + h * 64 + (r & 63);
            if (m0 + m > Mlim - 1) m = Mlim - 1 - m0;
            if constexpr (ACG) {
                const int mg = m0 + m;
                const int mv = p.gather[mg >> 2] * 4 + (mg & 3);
                voff[h ? 3 : 0][q] = (int)((conv_row_base(mv, p.lgS, p.Cin) - img0 * img_elems + chunk) * 2);
            } else {
                voff[h ? 3 : 0][q] = (int)((m * p.lda + chunk) * 2);
            }
            const int n = (r >> 5) * 64 + h * 32 + (r & 31);
            voff[1 + h][q] = (int)((n * ldb + chunk) * 2);
        }
    }
    auto stage = [&](int kind, int t) __attribute__((always_inline)) {
        if (ABL == 1 && t > 1) return;
        char* base = smem + (((t & 1) << 2) + kind) * HT + wid * 2048;
        const u16* g = (kind == 0 || kind == 3) ? a_blk : b_blk;
        int soff = t << 7;
        if constexpr (ACG) {
            if (kind == 0 || kind == 3) {            // K tile t = (64-channel chunk t/9, tap t%9) of the padded 18x18 map
                const int cc = t / 9, tap = t - cc * 9;
                const int ky = tap / 3, kx = tap - 3 * ky;
                soff = ((ky * SP + kx) * p.Cin + (cc << 6)) * 2;
            }
        }
        if constexpr (SEG) {
            if (kind == 0 || kind == 3) {            // K tile t: combination c = t / 16 -> own pixel (qy, qx) of the window, channels (t % 16) * 64
                const int c = seg_c0 + (t >> 4);
                const int cy = seg_nx == 2 ? (c >> 1) : c, cx = seg_nx == 2 ? (c & 1) : 0;
                const int qy = seg_ny == 1 ? (seg_py == 3) : cy, qx = seg_nx == 1 ? (seg_px == 3) : cx;
                g = a_blk + (long)(qy * 2 + qx) * p.seg_stride;       // wave-uniform: goes into the buffer descriptor
                soff = (t & 15) << 7;
            }
        }
        buf_load_lds16(g, voff[kind][0], soff, base);
        buf_load_lds16(g, voff[kind][1], soff, base + 1024);
    };

    // ---- fragment reads (row swizzle (row>>1)&7 only depends on lane&31: half tiles start at multiples of 32 rows)
    const int l31 = lane & 31, kh = lane >> 5, sw = (l31 >> 1) & 7;
    int ko[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ko[ks] = ((ks * 2 + kh) ^ sw) << 4;
    const int a_rd = (wr * 64 + l31) * 128, b_rd = (wc * 32 + l31) * 128;
    s16x8 af[2][4], bf[2][4];
    auto read_a = [&](int h, int par) __attribute__((always_inline)) {
        const char* base = smem + ((par << 2) + (h ? 3 : 0)) * HT + a_rd;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const s16x8*>(base + i * 4096 + ko[ks]);
    };
    auto read_b = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const char* base = smem + ((par << 2) + 1 + h) * HT + b_rd;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bf[h][ks] = *reinterpret_cast<const s16x8*>(base + ko[ks]);
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
                for (int j = 0; j < 2; ++j) {
                    if constexpr (EPI == EPI_STORE_F32T) acc[2 * a + i][j] = mfma32<ELEM>(bf[j][ks], af[i][ks], acc[2 * a + i][j]);
                    else acc[2 * a + i][j] = mfma32<ELEM>(af[i][ks], bf[j][ks], acc[2 * a + i][j]);
                }
        SGC_PP_BARRIER();
    };
    // end of a load section: counted wait for the half tiles the NEXT phase reads, retire this phase's reads, barrier
#define SGC_PP_CLOSE(STEADY)                                   \
    do {                                                       \
        if (STEADY) SGC_WAIT_VM(8); else SGC_WAIT_VM(0);       \
        SGC_WAIT_LGKM0();                                      \
        SGC_PP_BARRIER();                                      \
    } while (0)

    const int nk = Kloc >> 6;
    // ---- prologue: ring order is [A0 B0 B1](t) in phase Y(t-2), A1(t) in phase X(t-1)
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
    if (nk > 1) { stage(0, 1); stage(1, 1); stage(2, 1); SGC_WAIT_VM(8); } else SGC_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier slot behind wave row 0

    auto tile = [&](int t, int par, auto steady_c) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;       // tile t+2 exists
        read_a(0, par); read_b(par);                             // phase X
        if (t + 1 < nk) stage(3, t + 1);                         // A1(t+1) replaces A1(t-1), read in phase Y(t-1)
        SGC_PP_CLOSE(STEADY);
        half(0);
        read_a(1, par);                                          // phase Y
        if (STEADY) { stage(0, t + 2); stage(1, t + 2); stage(2, t + 2); }     // replace what phase X(t) read
        SGC_PP_CLOSE(STEADY);
        half(1);
    };
    int t = 0;
#pragma unroll 1
    for (; t + 2 < nk; ++t) tile(t, t & 1, std::true_type{});
#pragma unroll 1
    for (; t < nk; ++t) tile(t, t & 1, std::false_type{});
    if (wr == 0) __builtin_amdgcn_s_barrier();        // re-align the two wave rows

    bool stored = false;
    if constexpr (EPI == EPI_STORE && !ACG) {         // (gathered rows are scattered: the generic epilogue)
        if (p.epi_lds) { nt_epilogue_store16<ELEM>(p, acc, m0, n0, wr, wc, lane, wid, smem); stored = true; }
    }
    if constexpr (EPI == EPI_RELUMASK && !ACG) {
        if (p.epi_lds && !p.bias && (p.ldc & 7) == 0) { nt_epilogue_store16<ELEM, true>(p, acc, m0, n0, wr, wc, lane, wid, smem); stored = true; }
    }
    if (stored) {
    } else if constexpr (EPI == EPI_STORE_F32T) {
        static_assert(!ACG, "transposed f32 tile: plain rows only");
#ifdef SGC_EXPERIMENTS
        const unsigned long long clk1 = p.clk ? wall_clock64() : 0ULL;
#endif
        nt_epilogue_f32t(p, acc, m0, n0, wr, wc, lane);
#ifdef SGC_EXPERIMENTS
        if (p.clk && tid == 0) {                      // (wall clock: 100 MHz) time to ISSUE the stores, not to complete them
            const unsigned long long clk2 = wall_clock64();
            atomicAdd(p.clk, clk1 - clk0); atomicAdd(p.clk + 1, clk2 - clk1); atomicAdd(p.clk + 2, 1ULL);
        }
#endif
    }
    else if constexpr (ACG && EPI == EPI_POOL) {
        if (p.epi_lds) nt_epilogue_pool16<ELEM, true>(p, acc, m0, n0, wr, wc, lane, wid, smem, Mlim);
        else nt_epilogue<ELEM, EPI, 4, 2, true>(p, acc, m0, n0, wr, wc, lane, Mlim);
    }
    else if constexpr (ACG) nt_epilogue<ELEM, EPI, 4, 2, true>(p, acc, m0, n0, wr, wc, lane, Mlim);
    else nt_epilogue<ELEM, EPI, 4, 2>(p, acc, m0, n0, wr, wc, lane);
}

// conv3 over a window list with the ping-pong block (csrc/kernels_shared.hip): rows = 4 x max_entries, N = 1024, K = 9 x 512
template <int ELEM, int EPI>
static int launch_gemm_nt_pp_conv_gather(NtParams p, hipStream_t stream) {
    constexpr int LDS = 8 * 16384;
    if ((p.lgS != 4 && p.lgS != 5) || (p.Cin & 63) || p.K != 9 * p.Cin || (p.N & 255)) return SGC_ERR_ARG;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 256;
    const int al = sgc_tuning().acg_aligned;      // 1: grid padded to whole per-XCD patches
    p.patch_aligned = al;
    if (EPI == EPI_POOL) p.epi_lds = sgc_tuning().epi_lds;      // pooled rows through LDS to 16-byte row stores (nt_epilogue_pool16<.., GATHER>)
    auto kern = gemm_nt_pp_kernel<ELEM, EPI, 0, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SGC_LAUNCH(kern, dim3((unsigned)(al ? xcd_patch_grid(p.tiles_m, p.tiles_n) : p.tiles_m * p.tiles_n)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// SEG form (see gemm_nt_pp_kernel): M = listed windows, virtual N = 16 patch pixels x 512 channels, bf16 output through the LDS epilogue
template <int ELEM>
static int launch_gemm_nt_pp_seg(NtParams p, hipStream_t stream) {
    constexpr int LDS = EPI_LDS_BYTES;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 256;                          // 32 or, with the centre pixels split, 40
    p.epi_lds = 1;
    auto kern = gemm_nt_pp_kernel<ELEM, EPI_STORE, 0, 0, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int grid_aligned = xcd_patch_grid(p.tiles_m, p.tiles_n), nb = p.tiles_m * p.tiles_n;
    p.patch_aligned = (p.patch_gn == 0 && nb >= 1024 && (nb & 255) != 0 && grid_aligned * 10 <= nb * 11) ? 1 : 0;
    SGC_LAUNCH(kern, dim3((unsigned)(p.patch_aligned ? grid_aligned : nb)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

template <int ELEM, int EPI, int ABL = 0>
static int launch_gemm_nt_pp(NtParams p, hipStream_t stream) {
    constexpr int LDS = (EPI == EPI_STORE) ? EPI_LDS_BYTES : 8 * 16384;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 256;
    auto kern = gemm_nt_pp_kernel<ELEM, EPI, ABL>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    // whole patches per XCD only pay once every XCD has several of them; small grids keep the even spread of the contiguous walk
    const int grid_aligned = xcd_patch_grid(p.tiles_m, p.tiles_n);
    const int al = sgc_tuning().nt_aligned;       // 0 / 1 forces, -1 = by grid size
    const int nb = p.tiles_m * p.tiles_n;
    // the contiguous walk is itself patch-aligned when every XCD's share is a multiple of 32 tiles (fc1 data gradient: 126 x 256
    // tiles, measured 1 % faster than the round-robin patches); otherwise pad, unless the grid is small or the padding > 10 %
    p.patch_aligned = al >= 0 ? al : ((nb >= 1024 && (nb & 255) != 0 && grid_aligned * 10 <= nb * 11) ? 1 : 0);
    SGC_LAUNCH(kern, dim3((unsigned)(p.patch_aligned ? grid_aligned : p.tiles_m * p.tiles_n)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}


// LDS-staged bias + ReLU + 2x2 max-pool epilogue of the 8-wave block (conv3 forward: 4.2 GB of f16 + 2.1 GB of routing bytes
// per launch, + 4.2 GB when the bf16 copy is requested).  In the MFMA C layout a lane owns ONE column, so the direct form
// issues 2-byte and 1-byte stores (measured 2.0 ms of the 59 ms launch).  Here every wave pools in registers (the four
// accumulator registers of a window), writes its 32 windows x 64 channels into a private LDS tile (pitch 128 B / 64 B) and
// streams complete 128-byte / 64-byte rows out with 16-byte stores.
// Row pitches 128 B (16-bit tiles) and 64 B (arg-max tile), no padding: the 16-byte read-back covers four rows per 16-lane group and
// 32 / 16 dwords per row put them on disjoint bank quarters (see EPI_LDS_PITCH in gemm_nt.h; rounds 2-4: 144 / 80 B, two-way conflicts
// on every read); the narrow writes of a row fall several lanes to a dword either way.
constexpr int POOL_EPI_PY = 128, POOL_EPI_PA = 64;
constexpr int POOL_EPI_WAVE_BYTES = 32 * POOL_EPI_PY * 2 + 32 * POOL_EPI_PA;      // f16 tile + bf16 tile + argmax tile
// GATHER (conv3 over a window LIST, gemm_nt_pp_kernel<.., ACG = 1>): pooled row prow of the tile is list entry prow - y / its bf16 copy go
// to row dest[prow] (window-major row space) or gather[prow], the routing bytes to gather[prow]; entries >= *gather_n are not written;
// the raw accumulators of the entries >= raw_first (linear pairs: per-object pre-activations) leave directly from the registers.
// Rounds 2-4 sent this launch - the step's longest - through the generic epilogue: 96 two- and one-byte stores per lane.
template <int ELEM, bool GATHER>
__device__ __forceinline__ void nt_epilogue_pool16(const NtParams& p, f32x16 (&acc)[4][2], int m0, int n0, int wr, int wc, int lane,
                                                   int wid, char* smem, int m_limit) {
    __syncthreads();                                   // every wave is done reading operand tiles
    char* ty = smem + wid * POOL_EPI_WAVE_BYTES;
    char* tb = ty + 32 * POOL_EPI_PY;
    char* ta = tb + 32 * POOL_EPI_PY;
    const int h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int colw = j * 32 + cl;
            const float bias = p.bias ? p.bias[n0 + wc * 64 + colw] : 0.f;
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
                if (!(v > 0.f)) { v = 0.f; am = 4; }        // ReLU killed: no gradient path
                const int row = i * 8 + 2 * w + h;          // window inside the wave's 32
                if constexpr (GATHER) {
                    const int prow = (m0 >> 2) + wr * 32 + row;
                    if (p.raw && prow >= p.raw_first && prow * 4 < m_limit) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) p.raw[((long)(prow - p.raw_first) * 4 + q) * p.ldc + n0 + wc * 64 + colw] = acc[i][j][4 * w + q];
                    }
                }
                *reinterpret_cast<u16*>(ty + row * POOL_EPI_PY + colw * 2) = to_elem<ELEM>(v);
                if (p.C2) *reinterpret_cast<u16*>(tb + row * POOL_EPI_PY + colw * 2) = f32_to_bf16_bits(v);
                if (p.argmax) *reinterpret_cast<unsigned char*>(ta + row * POOL_EPI_PA + colw) = (unsigned char)am;
            }
        }
    __builtin_amdgcn_wave_barrier();
    const long prow0 = ((long)m0 >> 2) + wr * 32;
    const int c8 = lane & 7, r8 = lane >> 3;
    u16* out = reinterpret_cast<u16*>(p.C);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + r8;
        long orow = prow0 + row;
        if constexpr (GATHER) {
            if (orow * 4 >= m_limit) continue;
            orow = p.dest ? p.dest[orow] : p.gather[orow];
        } else if (p.wm_goff) orow = p.wm_goff[orow & 63] + (orow >> 6);      // window-major row space (shared fc1); argmax stays pair-major
        const long o = orow * p.ldc + n0 + wc * 64 + c8 * 8;
        *reinterpret_cast<uint4*>(out + o) = *reinterpret_cast<const uint4*>(ty + row * POOL_EPI_PY + c8 * 16);
        if (p.C2) *reinterpret_cast<uint4*>(p.C2 + o) = *reinterpret_cast<const uint4*>(tb + row * POOL_EPI_PY + c8 * 16);
    }
    if (p.argmax) {
        const int c4 = lane & 3, r16 = lane >> 2;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 16 + r16;
            long arow = prow0 + row;
            if constexpr (GATHER) {
                if (arow * 4 >= m_limit) continue;
                arow = p.gather[arow];
            }
            *reinterpret_cast<uint4*>(p.argmax + arow * p.ldc + n0 + wc * 64 + c4 * 16) =
                *reinterpret_cast<const uint4*>(ta + row * POOL_EPI_PA + c4 * 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Halo-staged implicit 3x3 convolution on 16x16 maps with the ping-pong schedule (conv3 forward / data gradient).
// One workgroup = one image (256 pixels, window-major rows) x 256 output channels; K step = (64-channel chunk, tap).
// The zero-padded 18x18x64 patch of a chunk is staged ONCE (41 KiB, double buffered) and the nine taps read their A
// fragments from it at shifted pixel rows (see conv16_halo_kernel in gemm_nt.h for the swizzle); the weight tile is
// staged per step as two half tiles B0/B1 in a four-slot ring (two steps).  LDS: 2 x 41 KiB + 4 x 16 KiB = 146 KiB.
// Per wave and step: phase Y issues the four weight loads of step s+2 (replacing what phase X(s) read) and, in the first
// taps of a chunk, one piece of the NEXT chunk's patch; it ends with vmcnt(4 or 5): only its own loads may still be in
// flight, so the weights of step s+1 and every older patch piece have landed.  Phase X issues nothing and has no vmcnt wait.
// ASRC = 1 (conv3 data gradient in the training step): the A operand is the gradient BEFORE the 2x2 max-pool, which has one
// non-zero per pooling window and channel.  Instead of DMA-ing a materialised un-pooled tensor (21 GB written by the un-pool pass
// and read back here), the block reads the POOLED gradient (p.Apool, 8x8 windows x Cin) and the routing byte of the forward
// arg-max (p.Acode) and scatters them into the LDS patch itself: thread = (window, 8-channel group) issues one 16-byte and one
// 8-byte global load per chunk in tap 0, and four ds_write_b128 (the four pixels of its window, value or zero) in tap 2, into the
// patch buffer of the NEXT chunk.  The zero border of both patch buffers is written once at kernel start.
template <int ELEM, int EPI, int ASRC = 0>
__global__ __launch_bounds__(512, 2) void conv16_halo_pp_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = 328 * 128;            // 324 patch rows, padded to 41 x 8 rows
    constexpr int HT = 16384;
    char* const abuf0 = smem;
    char* const bbuf0 = smem + 2 * A_BYTES;       // slot = (step & 1) * 2 + b-half
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    int img, tn;
    if (p.halo_walk) {          // the tiles_n blocks of one image are consecutive in the XCD's own sequence
        const int nb = p.tiles_m * p.tiles_n, q = nb >> 3, r = nb & 7, x = blockIdx.x & 7;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
        img = lin / p.tiles_n; tn = lin - img * p.tiles_n;
    } else {                    // N tile = block id mod tiles_n: with tiles_n = 2 or 4 every XCD keeps ONE weight tile in its L2
        img = blockIdx.x / p.tiles_n; tn = blockIdx.x - img * p.tiles_n;
    }
    const int m0 = img * 256, n0 = tn * 256;
    const int Cin = p.Cin;

    // ---- A patch loader: 41 pieces of 8 rows; wave w issues pieces w, w+8, ... (<= 6), one per phase Y
    const int lrow = lane >> 3, cpos = lane & 7;
    int a_off[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int r = (wid + 8 * i) * 8 + lrow;
        if (r > 323) r = 323;
        const int py = r / 18, px = r - py * 18;
        const int f = ((px >> 1) + 4 * (py & 1)) & 7;
        a_off[i] = r * Cin + ((cpos ^ f) << 3);
    }
    const u16* const a_img = ASRC ? p.A : p.A + (long)img * (324L * Cin);
    const int npiece = ASRC ? 0 : ((wid == 0) ? 6 : 5);
    auto stage_a_piece = [&](int i, int cc) __attribute__((always_inline)) {
        buf_load_lds16(a_img, a_off[i] * 2, cc << 7, abuf0 + (cc & 1) * A_BYTES + (wid + 8 * i) * 1024);
    };
    // ---- ASRC = 1: pooled source.  thread -> window pw (8x8) and channel group pg (8 channels of the 64-channel chunk).
    // The pooled values (64 windows x 128 B) and routing bytes (64 x 64 B) of a chunk go through a 12 KiB LDS scratch by DMA
    // (no registers are live across taps: a register-staged version made the compiler drain vmcnt right after the loads);
    // every wave DMAs its own 8 windows of values, waves 0-3 also 16 windows of routing bytes each.
    char* const pscr = smem + 2 * A_BYTES + 4 * HT;               // [64][128] values, then [64][64] routing bytes
    const int pw = tid >> 3, pg = tid & 7;
    // descriptor bases per image (64-bit), lane offsets inside the image's 64 windows (32-bit: < 128 KiB)
    const u16* const apool_img = ASRC ? p.Apool + (long)img * 64 * Cin : nullptr;
    const unsigned char* const acode_img = ASRC ? p.Acode + (long)img * 64 * Cin : nullptr;
    const int pool_voff = (pw * Cin + pg * 8) * 2;                                         // byte offset of the thread's 8 values
    const int pool_coff = (wid * 16 + (lane >> 2)) * Cin + (lane & 3) * 16;                // byte offset of the lane's 16 routing bytes
    const int npool = (wid < 4) ? 2 : 1;                           // DMA instructions this wave issues per chunk
    auto pool_dma = [&](int cc) __attribute__((always_inline)) {
        buf_load_lds16(apool_img, pool_voff, cc << 7, pscr + wid * 1024);
        if (wid < 4) buf_load_lds16(acode_img, pool_coff, cc << 6, pscr + 8192 + wid * 1024);
    };
    auto pool_scatter_q = [&](int cc, int q) __attribute__((always_inline)) {
        char* ab = abuf0 + (cc & 1) * A_BYTES;
        const uint4 pool_v = *reinterpret_cast<const uint4*>(pscr + tid * 16);
        const uint2 pool_c = *reinterpret_cast<const uint2*>(pscr + 8192 + pw * 64 + pg * 8);
        const u16* vh = reinterpret_cast<const u16*>(&pool_v);
        const unsigned char* ch = reinterpret_cast<const unsigned char*>(&pool_c);
        const int wy = pw >> 3, wx = pw & 7;
        // SWAR select: byte k of (code ^ q*0x01010101) is zero where channel k routes to position q.  Per-byte exact zero test
        // without cross-byte carries (bytes are < 0x80): ((x & 0x7f..) + 0x7f..) sets bit 7 of every NON-zero byte; a sign-extending
        // 1-bit field extract turns the inverted flag into a 16-bit lane mask.
        const unsigned qq = (unsigned)q * 0x01010101u;
        const unsigned x0 = pool_c.x ^ qq, x1 = pool_c.y ^ qq;
        const unsigned m0 = ~(((x0 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x0) & 0x80808080u;
        const unsigned m1 = ~(((x1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x1) & 0x80808080u;
        auto lanes = [](unsigned m, int e) __attribute__((always_inline)) {           // elements e (low half) and e+1 (high half) of m's four
            const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(m, 8 * e + 7, 1), hi = (unsigned)__builtin_amdgcn_sbfe(m, 8 * e + 15, 1);
            return (lo & 0x0000ffffu) | (hi & 0xffff0000u);
        };
        uint4 o;
        o.x = pool_v.x & lanes(m0, 0); o.y = pool_v.y & lanes(m0, 2); o.z = pool_v.z & lanes(m1, 0); o.w = pool_v.w & lanes(m1, 2);
        (void)vh; (void)ch;
        const int py = 2 * wy + (q >> 1) + 1, px = 2 * wx + (q & 1) + 1;
        const int f = ((px >> 1) + 4 * (py & 1)) & 7;
        *reinterpret_cast<uint4*>(ab + (py * 18 + px) * 128 + ((pg ^ f) << 4)) = o;
    };
    auto pool_scatter = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pool_scatter_q(cc, q);
    };
    // ---- B (weight) half tiles: LDS row r of half h <-> output channel n0 + (r>>5)*64 + h*32 + (r&31)
    const u16* const b_blk = p.B + (long)n0 * p.ldb;
    int b_voff[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wid * 2 + q) * 8 + lrow;
        const int chunk = (cpos ^ ((r >> 1) & 7)) << 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) b_voff[h][q] = (int)((((r >> 5) * 64 + h * 32 + (r & 31)) * p.ldb + chunk) * 2);
    }
    auto stage_b = [&](int step) __attribute__((always_inline)) {
        char* base = bbuf0 + ((step & 1) * 2) * HT + wid * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            buf_load_lds16(b_blk, b_voff[h][0], step << 7, base + h * HT);
            buf_load_lds16(b_blk, b_voff[h][1], step << 7, base + h * HT + 1024);
        }
    };

    // ---- fragment addressing
    const int l31 = lane & 31, kh = lane >> 5;
    int a_row[4], a_px[4], a_py[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = wr * 128 + i * 32 + l31;
        const int W = m >> 2, q = m & 3;
        a_py[i] = 2 * (W >> 3) + (q >> 1);
        a_px[i] = 2 * (W & 7) + (q & 1);
        a_row[i] = a_py[i] * 18 + a_px[i];
    }
    const int bsw = (l31 >> 1) & 7;
    int bko[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bko[ks] = ((ks * 2 + kh) ^ bsw) << 4;
    const int b_rd = (wc * 32 + l31) * 128;
    s16x8 af[2][4], bf[2][4];
    auto read_a = [&](int h, int cc, int ky, int kx) __attribute__((always_inline)) {
        const char* ab = abuf0 + (cc & 1) * A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ii = 2 * h + i;
            const char* rowp = ab + (a_row[ii] + ky * 18 + kx) * 128;
            const int f = (((a_px[ii] + kx) >> 1) + 4 * ((a_py[ii] + ky) & 1)) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const s16x8*>(rowp + (((ks * 2 + kh) ^ f) << 4));
        }
    };
    auto read_b = [&](int step) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const char* base = bbuf0 + ((step & 1) * 2 + h) * HT + b_rd;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bf[h][ks] = *reinterpret_cast<const s16x8*>(base + bko[ks]);
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

    const int ncc = Cin >> 6;
    const int nsteps = ncc * 9;
    // ---- prologue: patch of chunk 0, weights of steps 0 and 1
    if constexpr (ASRC) {
        pool_dma(0);
        for (int i = tid; i < 2 * A_BYTES / 16; i += 512)       // both patch buffers: the border stays zero for the whole kernel
            *reinterpret_cast<uint4*>(abuf0 + i * 16) = make_uint4(0, 0, 0, 0);
        SGC_WAIT_VM(0);
        __syncthreads();                                          // scratch of chunk 0 landed (all waves), zero fill visible
        pool_scatter(0);
        stage_b(0);
        if (nsteps > 1) { stage_b(1); SGC_WAIT_VM(4); } else SGC_WAIT_VM(0);
        SGC_WAIT_LGKM0();
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i < npiece) stage_a_piece(i, 0);
        stage_b(0);
        if (nsteps > 1) { stage_b(1); SGC_WAIT_VM(4); } else SGC_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    int step = 0;
    for (int cc = 0; cc < ncc; ++cc) {
        const bool more_cc = cc + 1 < ncc;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++step) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            // phase X: rows a0 x both weight halves
            read_a(0, cc, ky, kx); read_b(step);
            SGC_WAIT_LGKM0();
            SGC_PP_BARRIER();
            half(0);
            // phase Y: rows a1
            read_a(1, cc, ky, kx);
            if constexpr (ASRC) {
                // pooled source: tap 0 DMAs the next chunk's window values + routing bytes into the scratch (issued AFTER this phase's
                // four weight loads so that the counted wait leaves them in flight), tap 1's vmcnt(4) retires them in every wave
                // (they are older than its own weight loads) and its barrier publishes them, tap 2 scatters them into the other
                // patch buffer (last read in chunk cc-1; the scratch itself is next written in tap 0 of chunk cc+1).
                if (step + 2 < nsteps) {
                    stage_b(step + 2);
                    if (more_cc && tap == 0) {
                        pool_dma(cc + 1);
                        if (npool == 2) SGC_WAIT_VM(6); else SGC_WAIT_VM(5);
                    } else SGC_WAIT_VM(4);
                } else SGC_WAIT_VM(0);                      // last two steps: more_cc is false there (nsteps = 9 * ncc)
                // The scatter (~180 VALU operations, re-reads of the scratch, four ds_write_b128) costs the launch +4.5 % here (tap 2,
                // phase Y) - about what the un-pool pass saves by not writing the 21 GB tensor.  Measured alternatives: phase X of
                // tap 2 +5.6 %, spread over taps 2..5 +18 %, inside the wave's own MFMA phase the register file overflows (spills
                // inside the counted-vmcnt loop are not an option).
                if (more_cc && tap == 2) pool_scatter(cc + 1);

            } else if (step + 2 < nsteps) {
                stage_b(step + 2);
                if (more_cc && tap < npiece) {
                    switch (tap) {          // a_off[] must be indexed by a constant to stay in registers
                        case 0: stage_a_piece(0, cc + 1); break;
                        case 1: stage_a_piece(1, cc + 1); break;
                        case 2: stage_a_piece(2, cc + 1); break;
                        case 3: stage_a_piece(3, cc + 1); break;
                        case 4: stage_a_piece(4, cc + 1); break;
                        default: stage_a_piece(5, cc + 1); break;
                    }
                    SGC_WAIT_VM(5);
                } else SGC_WAIT_VM(4);
            } else SGC_WAIT_VM(0);
            SGC_WAIT_LGKM0();
            SGC_PP_BARRIER();
            half(1);
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    if constexpr (EPI == EPI_STORE) {
        if (p.epi_lds) { nt_epilogue_store16<ELEM>(p, acc, m0, n0, wr, wc, lane, wid, smem); return; }
    }
    if constexpr (EPI == EPI_POOL) {
        if (p.epi_lds) { nt_epilogue_pool16<ELEM, false>(p, acc, m0, n0, wr, wc, lane, wid, smem, 0); return; }
    }
    nt_epilogue<ELEM, EPI, 4, 2>(p, acc, m0, n0, wr, wc, lane);
}

template <int ELEM, int EPI>
static int launch_conv16_halo_pp(NtParams p, hipStream_t stream) {
    constexpr int LDS0 = 2 * 328 * 128 + 4 * 16384;
    constexpr int LDS = (EPI == EPI_STORE && LDS0 < EPI_LDS_BYTES) ? EPI_LDS_BYTES : LDS0;
    constexpr int LDS_POOLED = LDS0 + 64 * 128 + 64 * 64;        // + the 12 KiB scratch of the pooled source (158 KiB of 160)
    p.tiles_m = p.M / 256;
    p.tiles_n = p.N / 256;
    {
        // Walk 0 (N tile = block id mod tiles_n) keeps ONE weight tile per XCD L2-resident when it is < ~3 MiB (conv3 forward:
        // 2.4 MiB; FETCH_SIZE 25.6e6 KiB, stable) but makes the 4 XCD groups of an image's 4 N tiles each fetch its patch; walk 1
        // (XCD-contiguous image ranges) fetches every patch once and streams the weight tiles, which thrash the 4 MiB L2:
        // 28.7e6 - 45.5e6 KiB from run to run.  Launch time is the same within box noise (round 2: 60.2 vs 59.6 ms in one
        // alternated pair, 60.8 / 61.2 vs 60.9 / 61.0 in another; profiles/r02_halo_walk_ab.txt, r02_hook_sweep.txt) - the launch
        // is power-bound, not traffic-bound - so the forward keeps the walk with the lower, stable traffic.  A larger weight tile
        // (conv3 data gradient: 4.7 MiB) thrashes either way and takes walk 1.  (SgcTuning::halo_walk forces, experiments only.)
        const int hw = sgc_tuning().halo_walk;
        p.halo_walk = hw >= 0 ? hw : ((long)p.K * 512 > (3L << 20) ? 1 : 0);
    }
    if constexpr (ELEM == ELEM_BF16 && EPI == EPI_STORE) {
        if (p.Apool) {                                 // A operand given as pooled rows + routing byte: un-pooled inside the block
            auto kern = conv16_halo_pp_kernel<ELEM, EPI, 1>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_POOLED);
            SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), LDS_POOLED, stream, p);
            SGC_CHECK_LAUNCH();
            return SGC_OK;
        }
    }
    auto kern = conv16_halo_pp_kernel<ELEM, EPI>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
