// conv3 data gradient over listed windows on the SPARSE matrix cores (v_smfmac_f32_32x32x32_bf16): the patch form of
// gemm_nt_pp_kernel<SEG> (csrc/gemm_nt_pp.h, csrc/kernels_shared.hip: sgc_windows_dgrad_patches) with the structural zeros of the
// un-pooled gradient taken out of the instruction stream.
//
// The gradient of conv3's output before ReLU + 2x2 max-pool (model.py:145-147) has ONE non-zero per window and channel: channel c of
// window e routes to own pixel route(e, c) in 0..3 (4 = killed by the ReLU).  The gradient of patch pixel pp of the window's 4 x 4
// input patch is  sum over the (own pixel q, tap t = pp - q) combinations that exist  of  dy3[e, q, :] x W_t.  For a pixel with two
// combinations (q0, q1) order the contraction index as k = 2 c + j: the pair (2c, 2c+1) holds dy3[e, q0, c], dy3[e, q1, c] - at most
// one of them non-zero - so every group of four consecutive k holds at most two non-zeros: the 2:4 pattern.  The compressed operand
// of such a SET S = {q0, q1} is the POOLED row dy[e, :] with the channels that route elsewhere zeroed (one value per channel: 1024
// bf16 per window, the size of one un-pooled row for twice its K), the index bits say whether a channel went to q0 or q1.  Only four
// sets occur: {0,1}, {2,3} (own pixels of one window row) and {0,2}, {1,3} (one window column).  The 20 output slots of the dense
// form (16 patch pixels, the four centre pixels in two slots of two combinations each) map to:
//   edge pixels   (py in {0,3}, px in {1,2}): set {0,1} / {2,3};   (px in {0,3}, py in {1,2}): set {0,2} / {1,3}
//   centre pixels (two slots each): slot part 0 = set {0,1}, part 1 = set {2,3} (the split the dense kernel already makes)
//   corner pixels (one combination): set {0,1} or {2,3} with the weights of the absent combination ZERO (costs the issue time of the
//                 dense K = 1024 product it replaces - a sparse instruction covers twice the K - and keeps the kernel to one form)
// so every block is ONE set x ONE weight matrix B_slot [512 c_in][2048 = (c_out, j)], K = 1024 channels = 32 K tiles of 32 channels:
// 20 slots x 32 sparse tiles of 16 instructions per wave against 36 dense K = 1024 products of 32 x 16 instructions.
//
// Operand layout of the instruction: gemm_tn_sp.h (measured).  With k = 2 c + j an instruction (32 k) covers 16 channels; lane half
// ha holds the groups of channels 8 ha .. 8 ha + 7, one kept value per channel = 16 contiguous bytes of the masked pooled row, so the
// A fragment is a plain ds_read_b128 of a 64-byte LDS row (32 channels per K tile); index nibble of the lane's group g' (channels
// 8 ha + 2 g', + 1): first kept position j(channel) in {0,1}, second 2 + j(channel + 1).  B fragment: elements 0-7 = k 8 hb .. 8 hb + 7,
// 8-15 = k 16 + 8 hb ..: two ds_read_b128 of the 128-byte weight row (64 k per K tile).
//
// Block = the ping-pong block of gemm_nt_pp.h / gemm_tn_sp.h: 256 windows x 256 c_in, 8 waves of 128 x 64, half tiles in an 8-slot
// ring of 16 KiB slots: A half = 128 rows x 64 B of masked values (one load per wave) + 1 KiB of index words ([lane half][row] u32:
// conflict-free ds_read_b32; one load, wave 0 for A0 / wave 1 for A1), B half = 128 rows x 128 B (two loads per wave).
#include "common.h"
#include "gemm_nt.h"
#include "gemm_tn.h"

typedef __bf16 bf16x8_sp __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16_sp __attribute__((ext_vector_type(16)));
typedef short s16x16_sp_t __attribute__((ext_vector_type(16)));

static inline int grid_cap_sp(long items, long per_block, int cap) {
    long b = (items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

struct NtSpParams {
    const u16* Ac;            // [4 sets][entries][1024] bf16: pooled rows masked to the set
    const unsigned* Ic;       // [4 sets][entries / 256][2 halves][32 K tiles][2 lane halves][128 rows] u32 index words
    const u16* B;             // [20 slots][512][2048] bf16
    u16* C;                   // patch [entries][16][512] bf16 (one row per patch pixel)
    int entries;              // multiple of 256
    int tiles_m;
};

// slot (0..19, the dense kernel's order: pp natural, centre pixels two slots) -> set index (0 {0,1}, 1 {2,3}, 2 {0,2}, 3 {1,3})
__host__ __device__ __forceinline__ int nt_sp_slot_set(int slot, int* pp_out = nullptr, int* part_out = nullptr) {
    int acc = 0, pp = 0, part = 0;
    for (pp = 0; pp < 16; ++pp) {
        const int wdt = (((pp >> 2) == 1 || (pp >> 2) == 2) && ((pp & 3) == 1 || (pp & 3) == 2)) ? 2 : 1;
        if (slot < acc + wdt) { part = slot - acc; break; }
        acc += wdt;
    }
    if (pp_out) *pp_out = pp;
    if (part_out) *part_out = part;
    const int py = pp >> 2, px = pp & 3;
    const bool ymid = py == 1 || py == 2, xmid = px == 1 || px == 2;
    if (ymid && xmid) return part;                     // centre: part 0 = own pixels {0,1}, part 1 = {2,3}
    if (!ymid && xmid) return py == 0 ? 0 : 1;         // top / bottom edge: both own pixels of window row qy
    if (ymid && !xmid) return px == 0 ? 2 : 3;         // left / right edge: both own pixels of window column qx
    return py == 0 ? 0 : 1;                            // corner: own pixel (qy, qx) = (py == 3, px == 3) inside the row set of qy
}

// (Measured and dropped: issuing the global -> LDS pieces INSIDE the matrix slots, between the sparse instructions, instead of in the
// load sections - 5.10 against 5.10 ms: the load sections are not issue-bound.  What the block waits for is the ARRIVAL of its
// operands: matrix pipes 37 % busy, LDS 26 % busy (profiles/r05_mid_pmc_*.csv); see the tile walk below.)
// STAGES: K tiles of the operand ring.  2 (shipped): the ring of gemm_nt_pp.h (a half tile's successor is issued two K tiles ahead); 3:
// three K tiles ahead - a stage is 50 KiB here ([A0 9][A1 9][B0 16][B1 16] KiB: the masked rows are half the size of a dense operand),
// so three fit the 160 KiB and the epilogue reuses them.  Measured, alternated in one box (profiles/r05_sparse_dgrad_walk_ab.txt): three
// stages 5.17 / 5.19 / 5.15 ms against 5.05 / 5.05 / 5.00 for two - a deeper ring does not help (as gemm_tn_sp.h found for the weight
// gradient), although the operands' arrival is what the block waits for: with the masked rows of ONE M tile fed to every block
// (all operands L2-resident; an experiment, results wrong) the launch takes 4.43 instead of 5.08 ms, i.e. 46 instead of 36 GB/s of
// global -> LDS traffic per CU - the dense block's rate (43 GB/s at 1 450 TFLOP/s).  The sparse blocks run at the rate the CUs' load
// path delivers 50 KiB per K tile, not at the matrix pipes' (37 % busy): what would help is fewer operand bytes per instruction.
template <int STAGES>
__global__ __launch_bounds__(512, 2) void gemm_nt_sp_kernel(const NtSpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ST = 51200;                        // stage stride; in a stage: kind 0 A0 at 0, 3 A1 at 9216, 1 B0 at 18432, 2 B1 at 34816
    auto slot_base = [&](int t, int kind) __attribute__((always_inline)) {
        const int koff = kind == 0 ? 0 : (kind == 3 ? 9216 : (kind == 1 ? 18432 : 34816));
        return smem + (STAGES == 2 ? (t & 1) : (t % 3)) * ST + koff;
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    // Tile walk.  A patch = 16 M tiles x the two channel halves of ONE patch pixel (32 blocks = what an XCD holds: the pixel's weights
    // stay in its L2); patches are dealt to the 8 XCDs (block id % 8 = XCD) M-group major, so at any time the whole chip works on the 16
    // pixels of a FEW M groups: their masked rows (16 tiles x 4 sets x 0.6 MB) are fetched from HBM once and re-read - by the other
    // pixels of the set, on other XCDs - from the Infinity Cache.  Measured against the walk of gemm_nt_pp_kernel<SEG> (every XCD its own
    // contiguous range of M groups: eight groups' rows + the weights exceed the 256 MB of that cache): 5.07 against 5.67 ms per launch,
    // alternated (profiles/r05_sparse_dgrad_walk_ab.txt).  The four centre pixels run both of their sets (64 K tiles instead of 32), so
    // the deal is by work: per period of two M groups (G0, G1) every XCD gets one centre pixel and three others -
    //   XCD x < 4 :  centre x of G0,  other 8+x of G0,  other x of G1,    other 4+x of G1
    //   XCD 4 + y :  other y of G0,   other 4+y of G0,  centre y of G1,   other 8+y of G1          (5 units of 32 K tiles each).
    int tm, pp, nhalf;
    {
        const int x = blockIdx.x & 7, jj = blockIdx.x >> 3;
        const int seq = jj >> 5, w = jj & 31;
        const int period = seq >> 2, n = seq & 3, y = x & 3;
        int gsel, centre, idx;
        if (x < 4) { gsel = n >> 1; centre = n == 0; idx = n == 0 ? y : (n == 1 ? 8 + y : (n == 2 ? y : 4 + y)); }
        else { gsel = n >> 1; centre = n == 2; idx = n == 0 ? y : (n == 1 ? 4 + y : (n == 2 ? y : 8 + y)); }
        const int G = 2 * period + gsel;
        if (G * 16 >= p.tiles_m) return;
        // centre pixels 5, 6, 9, 10; the twelve others in natural order: 0 1 2 3 4 7 8 11 12 13 14 15
        pp = centre ? (5 + (idx & 1) + 4 * (idx >> 1)) : (idx < 5 ? idx : (idx == 5 ? 7 : (idx == 6 ? 8 : idx + 4)));
        tm = G * 16 + (w >> 1);
        nhalf = w & 1;
        if (tm >= p.tiles_m) return;
    }
    const bool is_centre = ((pp >> 2) == 1 || (pp >> 2) == 2) && ((pp & 3) == 1 || (pp & 3) == 2);
    // weight matrices in the order of the dense form's 20 slots (centre pixels: two consecutive matrices, one per set)
    const int slot = pp + (pp > 5) + (pp > 6) + (pp > 9) + (pp > 10);
    const int set = nt_sp_slot_set(slot);          // first (or only) set of the pixel; a centre pixel's second pass: set + 1, matrix slot + 1
    const int m0 = tm * 256;

    // ---- staging sources
    // A half h: LDS row r (0..127) <-> window m0 + (r>>6)*128 + h*64 + (r&63); wave w stages rows 16w .. 16w+15 (64 B each)
    const int arow = wid * 16 + (lane >> 2);
    const int achunk = (lane & 3) ^ ((arow >> 2) & 3);
    // wave-uniform base = this M tile's first row (64-bit), per-lane offsets relative to it (< 512 KiB): the list may hold any number of
    // windows - offsets from the set's first row would leave the buffer descriptor's 2 GiB range at 2^20 windows and read zeros
    const u16* const a_set = p.Ac + ((long)set * p.entries + m0) * 1024;
    int a_voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) a_voff[h] = (((arow >> 6) * 128 + h * 64 + (arow & 63)) * 1024 + achunk * 8) * 2;
    // index words of half h, K tile t: 1 KiB contiguous at ((((set * tiles_m + tm) * 2 + h) * 32 + t) * 256) words
    const unsigned* const i_blk = p.Ic + ((long)(set * p.tiles_m + tm) * 2) * 32 * 256;
    // B half h: LDS row r <-> c_in nhalf*256 + (r>>5)*64 + h*32 + (r&31); wave w stages rows 16w .. 16w+15 (two loads of 8 rows x 128 B)
    const u16* const b_blk = p.B + (long)slot * 512 * 2048 + (long)(nhalf * 256) * 2048;
    const int lrow = lane >> 3, cpos = lane & 7;
    int b_voff[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wid * 2 + q) * 8 + lrow;
        const int chunk = (cpos ^ ((r >> 1) & 7)) << 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) b_voff[h][q] = (((r >> 5) * 64 + h * 32 + (r & 31)) * 2048 + chunk) * 2;
    }
    // second pass of a centre pixel (K tiles 32 .. 63): the next set's rows / index words, the next weight matrix (wave-uniform bases)
    const long a_pass = (long)p.entries * 1024, i_pass = (long)p.tiles_m * 2 * 32 * 256, b_pass = 512L * 2048;
    auto stage = [&](int kind, int t) __attribute__((always_inline)) {
        char* base = slot_base(t, kind);
        const int pass = t >> 5, tt = t & 31;
        if (kind == 0 || kind == 3) {
            const int h = kind ? 1 : 0;
            buf_load_lds16(a_set + pass * a_pass, a_voff[h], tt << 6, base + wid * 1024);                   // 32 channels = 64 B per K tile
            if (wid == h) buf_load_lds16(i_blk + pass * i_pass, lane << 4, (h * 32 + tt) << 10, base + 8192);
        } else {
            buf_load_lds16(b_blk + pass * b_pass, b_voff[kind - 1][0], tt << 7, base + wid * 2048);          // 64 k = 128 B per K tile
            buf_load_lds16(b_blk + pass * b_pass, b_voff[kind - 1][1], tt << 7, base + wid * 2048 + 1024);
        }
    };

    // ---- fragment reads
    const int l31 = lane & 31, kh = lane >> 5;
    s16x8 af[2][2];                 // [tile i of the half][s]
    unsigned ai[2];                 // index word of tile i (low 16 bits: s = 0, high: s = 1)
    s16x8 bfr[2][2][2];             // [b half j][s][first / second 8 elements]
    auto read_a = [&](int h, int par) __attribute__((always_inline)) {
        const char* base = slot_base(par, h ? 3 : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wr * 64 + i * 32 + l31;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                af[i][s] = *reinterpret_cast<const s16x8*>(base + r * 64 + (((2 * s + kh) ^ ((r >> 2) & 3)) << 4));
            ai[i] = *reinterpret_cast<const unsigned*>(base + 8192 + kh * 512 + r * 4);
        }
    };
    const int b_rd = (wc * 32 + l31) * 128, bsw = (l31 >> 1) & 7;
    auto read_b = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const char* base = slot_base(par, 1 + h) + b_rd;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)          // elements 8c .. 8c+7 of the instruction's B operand: k = 16 c + 8 kh .. of its 32
                    bfr[h][s][c] = *reinterpret_cast<const s16x8*>(base + (((4 * s + 2 * c + kh) ^ bsw) << 4));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mm = [&](int a, int s, int i, int j) __attribute__((always_inline)) {
        const bf16x8_sp av = __builtin_bit_cast(bf16x8_sp, af[i][s]);
        const s16x16_sp_t bw = __builtin_shufflevector(bfr[j][s][0], bfr[j][s][1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        const bf16x16_sp bv = __builtin_bit_cast(bf16x16_sp, bw);
        if (s == 0) acc[2 * a + i][j] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(av, bv, acc[2 * a + i][j], (int)ai[i], 0, 0);
        else acc[2 * a + i][j] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(av, bv, acc[2 * a + i][j], (int)ai[i], 0, 1);
    };
    auto half = [&](int a) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mm(a, s, i, j);
        SGC_PP_BARRIER();
    };

    const int nk = is_centre ? 64 : 32;
    // pieces per wave and K tile: A0 1 (+1 index load on wave 0), B0 2, B1 2, A1 1 (+1 on wave 1)
    if constexpr (STAGES == 2) {
        stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
        stage(0, 1); stage(1, 1); stage(2, 1);
        if (wid == 0) SGC_WAIT_VM(6); else SGC_WAIT_VM(5);       // tile 0 landed; [A0 B0 B1](1) may stay in flight
    } else {
        stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
        stage(0, 1); stage(1, 1); stage(2, 1); stage(3, 1);
        stage(0, 2); stage(1, 2); stage(2, 2);
        if (wid == 0) SGC_WAIT_VM(13); else if (wid == 1) SGC_WAIT_VM(12); else SGC_WAIT_VM(11);      // tile 0 landed; tile 1 (6 / 7 pieces) and [A0 B0 B1](2) (5 / 6) may fly
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    // counted waits.  Two stages: phase X(t) issues A1(t+1), phase Y(t) [A0 B0 B1](t+2); the wait that leaves the last four half tiles
    // in flight is vmcnt(7) on waves 0 and 1 and vmcnt(6) on the others (as gemm_tn_sp_kernel).  Three stages: X(t) issues A1(t+2),
    // Y(t) [A0 B0 B1](t+3); close X needs A1(t) [issued X(t-2)], close Y needs [A0 B0 B1](t+1) [issued Y(t-2)]: in both cases the four
    // groups issued since may fly: 2 x (5 + 1) = 12 pieces, + the index loads of waves 0 and 1 (two each).
    auto close = [&](bool steady) __attribute__((always_inline)) {
        if (!steady) SGC_WAIT_VM(0);
        else if (STAGES == 2) { if (wid < 2) SGC_WAIT_VM(7); else SGC_WAIT_VM(6); }
        else { if (wid < 2) SGC_WAIT_VM(14); else SGC_WAIT_VM(12); }
        SGC_WAIT_LGKM0();
        SGC_PP_BARRIER();
    };
    auto tile = [&](int it, int par, auto steady_c) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;       // tile it + STAGES exists
        read_a(0, par); read_b(par);
        if (it + STAGES - 1 < nk) stage(3, it + STAGES - 1);
        close(STEADY);
        half(0);
        read_a(1, par);
        if (STEADY) { stage(0, it + STAGES); stage(1, it + STAGES); stage(2, it + STAGES); }
        close(STEADY);
        half(1);
    };
    int it = 0;
#pragma unroll 1
    for (; it + STAGES < nk; ++it) tile(it, it, std::true_type{});
#pragma unroll 1
    for (; it < nk; ++it) tile(it, it, std::false_type{});
    if (wr == 0) __builtin_amdgcn_s_barrier();

    NtParams q{};
    q.C = p.C; q.M = p.entries; q.ldc = 16 * 512; q.bias = nullptr;
    nt_epilogue_store16<ELEM_BF16>(q, acc, m0, pp * 512 + nhalf * 256, wr, wc, lane, wid, smem);
}

// ---------------------------------------------------------------------------------------------------- operands
// B_slot[n][2 c + j] = W[c_out = c][c_in = n][tap (py - qy_j, px - qx_j)] for the two own pixels (q_0, q_1) of the slot's set; 0 where
// the tap does not exist (corner pixels).  w: conv3_1.weight f32 [1024][512][3][3] (model.py:114).
__global__ __launch_bounds__(256) void nt_sp_weights_kernel(const float* __restrict__ w, u16* __restrict__ B) {
    const int slot = blockIdx.y;
    int pp, part;
    const int set = nt_sp_slot_set(slot, &pp, &part);
    const int py = pp >> 2, px = pp & 3;
    const int q0 = set == 0 ? 0 : (set == 1 ? 2 : (set == 2 ? 0 : 1)), q1 = set == 0 ? 1 : (set == 1 ? 3 : (set == 2 ? 2 : 3));
    int tap[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = j ? q1 : q0;
        const int ky = py - (q >> 1), kx = px - (q & 1);
        tap[j] = (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) ? ky * 3 + kx : -1;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 512L * 1024; i += (long)gridDim.x * 256) {
        const int n = (int)(i >> 10), c = (int)(i & 1023);
        const float* src = w + ((long)c * 512 + n) * 9;
        const unsigned lo = tap[0] >= 0 ? f32_to_bf16_bits(src[tap[0]]) : 0u;
        const unsigned hi = tap[1] >= 0 ? f32_to_bf16_bits(src[tap[1]]) : 0u;
        reinterpret_cast<unsigned*>(B)[((long)slot * 512 + n) * 1024 + c] = lo | (hi << 16);
    }
}

// Masked pooled rows + index words of the four sets for the listed windows 0 .. n_sparse-1 (a multiple of 256), and the conv3 bias
// partial sums of those windows (what windows_unpool_kernel provides for the entries it un-pools).  A workgroup walks GROUPS of 64
// consecutive windows (one (M tile, wave row, half) of the GEMM block: the 64 rows r0 .. r0+63 of its index lines); one wavefront per
// window, lane = 16 channels = one instruction (K tile lane/2, s = lane & 1); the masked rows leave as 2 KiB per wavefront, the index
// words are collected in LDS ([set][K tile][lane half][64 rows]) and leave as 256-byte runs.
__global__ __launch_bounds__(256) void nt_sp_pack_kernel(const u16* __restrict__ dywm, const unsigned char* __restrict__ am,
                                                         const int* __restrict__ gather, const int* __restrict__ dest, int n_sparse,
                                                         u16* __restrict__ Ac, unsigned* __restrict__ Ic, float* __restrict__ bias_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sidx = reinterpret_cast<unsigned*>(smem);            // [4 sets][32 K tiles][2 lane halves][64 rows]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_m = n_sparse >> 8, n_groups = n_sparse >> 6;
    float bs[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bs[k] = 0.f;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        for (int wi = wv; wi < 64; wi += 4) {
            const int e = grp * 64 + wi;
            const long row = gather[e], drow = dest ? (long)dest[e] : row;
            uint4 v[2];
            v[0] = *reinterpret_cast<const uint4*>(dywm + drow * 1024 + lane * 16);
            v[1] = *reinterpret_cast<const uint4*>(dywm + drow * 1024 + lane * 16 + 8);
            const uint4 cd = *reinterpret_cast<const uint4*>(am + row * 1024 + lane * 16);
            const u16* vh = reinterpret_cast<const u16*>(v);
            const unsigned cw[4] = {cd.x, cd.y, cd.z, cd.w};           // 16 routing bytes (0..3 own pixel, 4 = killed by the ReLU)
            const unsigned vw[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
            // SWAR over the routing bytes (all < 0x80): eq[q][d] has bit 0 of byte b set where channel 4d + b routes to own pixel q
            unsigned eq[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned x = cw[d] ^ (0x01010101u * (unsigned)q);
                    eq[q][d] = (~(x + 0x7f7f7f7fu) >> 7) & 0x01010101u;
                }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned live = eq[0][d] | eq[1][d] | eq[2][d] | eq[3][d];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if ((live >> (8 * b)) & 1u) bs[4 * d + b] += bf16_bits_to_f32(vh[4 * d + b]);
            }
#pragma unroll
            for (int set = 0; set < 4; ++set) {
                const int q0 = set == 0 ? 0 : (set == 1 ? 2 : (set == 2 ? 0 : 1)), q1 = set == 0 ? 1 : (set == 1 ? 3 : (set == 2 ? 2 : 3));
                unsigned ow[8];
                unsigned jb[4];                            // per code dword: the four j bits of its channels at bit positions 0, 2, 4, 6
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned in = (eq[q0][d] | eq[q1][d]) * 0xffu;                // 0xff where the channel belongs to the set
                    ow[2 * d] = vw[2 * d] & __builtin_amdgcn_perm(0u, in, 0x01010000u);          // channels 4d, 4d+1: bytes 0, 1 of ``in`` doubled
                    ow[2 * d + 1] = vw[2 * d + 1] & __builtin_amdgcn_perm(0u, in, 0x03030202u);  // channels 4d+2, 4d+3
                    jb[d] = (eq[q1][d] * 0x01041040u) >> 24;
                }
                // index nibble of the channel pair (2g', 2g'+1): first kept position j(even) in {0,1}, second 2 + j(odd): 8 | j_even | j_odd << 2
                const unsigned bits0 = 0x8888u | jb[0] | (jb[1] << 8), bits1 = 0x8888u | jb[2] | (jb[3] << 8);     // lane half 0 / 1
                u16* dst = Ac + ((long)set * n_sparse + e) * 1024 + lane * 16;
                *reinterpret_cast<uint4*>(dst) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                *reinterpret_cast<uint4*>(dst + 8) = make_uint4(ow[4], ow[5], ow[6], ow[7]);
                // the 32-bit word of (K tile, lane half): s = 0 from the even lane (low 16 bits), s = 1 from the odd lane (high)
                const unsigned mine = bits0 | (bits1 << 16);
                const unsigned other = __shfl_xor(mine, 1);
                if (!(lane & 1)) {
                    const int t = lane >> 1;
                    unsigned* ib = sidx + ((set * 32 + t) * 2) * 64 + wi;
                    ib[0] = (mine & 0xffffu) | (other << 16);                 // lane half 0
                    ib[64] = (mine >> 16) | (other & 0xffff0000u);            // lane half 1
                }
            }
        }
        __syncthreads();
        {   // group grp = windows e0 .. e0+63: M tile tm, rows r0 .. r0+63 of half h
            const int e0 = grp * 64, tm = e0 >> 8, el = e0 & 255;
            const int h = (el >> 6) & 1, r0 = (el >> 7) * 64;
            for (int i = threadIdx.x; i < 4 * 32 * 2 * 64; i += 256) {
                const int rr = i & 63, line = i >> 6;                      // line = (set * 32 + t) * 2 + ha
                const int set = line >> 6, t = (line >> 1) & 31, ha = line & 1;
                Ic[(((((long)(set * tiles_m + tm) * 2 + h) * 32 + t) * 2 + ha) * 128) + r0 + rr] = sidx[i];
            }
        }
        __syncthreads();
    }
    if (bias_part) {
        float* red = reinterpret_cast<float*>(smem);       // [4][1024]
#pragma unroll
        for (int k = 0; k < 16; ++k) red[wv * 1024 + lane * 16 + k] = bs[k];
        __syncthreads();
        for (int c = threadIdx.x; c < 1024; c += 256)
            bias_part[(long)blockIdx.x * 1024 + c] = (red[c] + red[1024 + c]) + (red[2048 + c] + red[3072 + c]);
    }
}

static int launch_gemm_nt_sp(NtSpParams p, hipStream_t stream) {
    constexpr int LDS = 3 * 51200;                                  // three stages (the epilogue's 128 KiB fit inside)
    if (p.entries <= 0) return SGC_OK;
    if (p.entries & 255) return SGC_ERR_ARG;
    p.tiles_m = p.entries >> 8;
    auto kern = gemm_nt_sp_kernel<2>;          // <3>: measured, no gain (see the kernel's header)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int groups = (p.tiles_m + 15) >> 4, periods = (groups + 1) >> 1;
    const unsigned grid = (unsigned)(periods * 4 * 8 * 32);                 // per period (two M groups): four patches of 32 blocks on each of the 8 XCDs
    SGC_LAUNCH(kern, dim3(grid), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

extern "C" {

int sgc_windows_dgrad_sparse_weights(const float* conv3_weight, void* w3sp, void* stream) {
    SGC_LAUNCH(nt_sp_weights_kernel, dim3(512, 20), dim3(256), 0, (hipStream_t)stream, conv3_weight, (u16*)w3sp);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_windows_dgrad_sparse_pack(const void* dywm, const unsigned char* argmax, const int* gather, const int* dest, int n_sparse,
                                  void* pack_a, void* pack_i, float* bias_part, int* n_parts, void* stream) {
    if (n_parts) *n_parts = 0;
    if (n_sparse <= 0) return SGC_OK;
    if (n_sparse & 255) return SGC_ERR_ARG;
    const int blocks = grid_cap_sp(n_sparse >> 6, 1, 768);           // groups of 64 windows; 64 KiB of LDS: two workgroups per CU
    if (n_parts && bias_part) *n_parts = blocks;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nt_sp_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    SGC_LAUNCH(nt_sp_pack_kernel, dim3(blocks), dim3(256), 65536, (hipStream_t)stream, (const u16*)dywm, argmax, gather, dest, n_sparse, (u16*)pack_a,
               (unsigned*)pack_i, bias_part);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_windows_dgrad_patches_sparse(const void* pack_a, const void* pack_i, int n_sparse, const void* w3sp, void* patch, void* stream) {
    if (n_sparse <= 0) return SGC_OK;
    if (n_sparse & 255) return SGC_ERR_ARG;
    NtSpParams p{};
    p.Ac = (const u16*)pack_a; p.Ic = (const unsigned*)pack_i; p.B = (const u16*)w3sp; p.C = (u16*)patch; p.entries = n_sparse;
    return launch_gemm_nt_sp(p, (hipStream_t)stream);
}

}  // extern "C"
