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
#include <type_traits>

enum { AMODE_PLAIN = 0, AMODE_CONV = 1, AMODE_CONV_GATHER = 2 };   // GATHER: conv rows come from a list of 2x2 windows (see NtParams::gather)
enum { EPI_STORE = 0, EPI_BIAS_TANH = 1, EPI_BIAS_RELU = 2, EPI_POOL = 3, EPI_FC2 = 4, EPI_RELUMASK = 5, EPI_STORE_F32 = 6,
       EPI_STORE_F32T = 7 };   // F32T (ping-pong block only): accumulators held TRANSPOSED (operands swapped in the MFMA), see nt_epilogue_f32t

struct NtParams {
    const u16* A; const u16* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int lgS, Cin;               // AMODE_CONV: image side S = 1<<lgS, channels per tap
    const float* bias;          // [N] or nullptr
    const u16* mask_src;        // EPI_RELUMASK: forward activation [M][ldc], gradient passes where it is > 0
    const float* lsub; const float* lobj; const int* sub_idx; const int* obj_idx;   // EPI_FC2
    unsigned char* argmax;      // EPI_POOL (optional)
    u16* C2;                    // EPI_POOL (optional): bf16 copy of the pooled output (operand of the fc1 weight gradient)
    float scale;                // EPI_RELUMASK / dropout scale
    unsigned drop_seed; int drop_enable;
    int tiles_m, tiles_n;
    int epi_lds;                // 1: LDS-staged 16-byte stores for EPI_STORE in the 8-wave kernels (set by the launcher)
    int halo_walk;              // conv16_halo_pp_kernel: 0 = block id -> (image, N tile) directly, 1 = XCD-contiguous image ranges
    int patch_aligned;          // gemm_nt_pp_kernel: 1 = grid padded to whole 32-tile patches (set by the launcher for large grids)
    const u16* Apool; const unsigned char* Acode;   // conv16_halo_pp_kernel<.., ASRC = 1>: the A operand as POOLED rows [img*64 windows][Cin]
                                                    // + routing byte (0..3 = position inside the 2x2 window, 4 = none); un-pooled on the fly
    const int* gather; const int* gather_n;         // AMODE_CONV_GATHER: row m = 4*e + q is pixel q of window gather[e] = image*(S/2)^2 + window;
                                                    // *gather_n = number of list entries (device side; p.M is only the launch bound);
                                                    // EPI_POOL writes entry e to pooled row gather[e]
    // window-major row space of the shared fc1 (csrc/kernels_shared.hip): rows grouped by pooling window, groups padded to 256 rows
    const int* dest;                                // AMODE_CONV_GATHER + EPI_POOL: y / bf16 copy of entry e go to row dest[e] (argmax stays at gather[e])
    const int* wm_goff;                             // EPI_POOL: y / bf16 copy of pooled row r go to row wm_goff[r & 63] + (r >> 6) (argmax stays at r)
    float* raw; int raw_first;                      // AMODE_CONV_GATHER + EPI_POOL: the accumulators (no bias / ReLU / pooling) of the entries
                                                    // e >= raw_first also go to raw[(e - raw_first) * 4 + pixel][ldc] (linear pairs, kernels_shared.hip)
    const int* tile_group; long group_stride;       // gemm_nt_pp_kernel: M tile t multiplies with B + tile_group[t] * group_stride
    u16* Cx; int x_first;                           // nt_epilogue_f32t (fc1 over the window-major rows): rows whose index inside their group (row -
                                                    // wm_goff[tile_group[tile]]) is >= x_first - the pair-specific X rows - leave as f16 to Cx [M][N]
                                                    // instead of f32 to C; the per-object rows in front of them (2-D prefix sums follow) stay f32
    int nt_store;                                   // nt_epilogue_f32t / nt_epilogue_store16: non-temporal stores (tools/fc1_windows_microbench.py)
    unsigned long long* clk;                        // gemm_nt_pp_kernel (tools/fc1_windows_microbench.py): per-block wall clocks summed: [0] main loop, [1] epilogue, [2] blocks
    long seg_stride; int seg_bpad, seg_split, patch_gn;                  // gemm_nt_pp_kernel<SEG>: elements between the segments (own pixels q) of a row of A; padding of B_pp's rows
    int stagger, stagger_phases;                    // gemm_nt_pp_kernel: the blocks of the FIRST generation (one per CU) start (id/8 % phases) * stagger
                                                    // sleep units (~4 us) late, so that the CUs' store phases do not coincide (0: off)
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

template <int ELEM, int EPI, int TM, int TN, bool GATHER = false>
__device__ __forceinline__ void nt_epilogue(const NtParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc, int lane, int m_limit = -1) {
    const int M = GATHER ? m_limit : p.M;
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
                    if (prow * 4 < M) {
                        long arow = prow, yrow = prow;
                        if constexpr (GATHER) {
                            arow = p.gather[prow]; yrow = p.dest ? p.dest[prow] : arow;
                            if (p.raw && prow >= p.raw_first) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) p.raw[((long)(prow - p.raw_first) * 4 + q) * p.ldc + col] = acc[i][j][4 * w + q];
                            }
                        }
                        else if (p.wm_goff) yrow = p.wm_goff[prow & 63] + (prow >> 6);
                        if (!(v > 0.f)) { v = 0.f; am = 4; }        // ReLU killed: no gradient path
                        out[yrow * p.ldc + col] = to_elem<ELEM>(v);
                        if (p.C2) p.C2[yrow * p.ldc + col] = f32_to_bf16_bits(v);
                        if (p.argmax) p.argmax[arow * p.ldc + col] = (unsigned char)am;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row >= M) continue;
                    float v = acc[i][j][r];
                    long o = (long)row * p.ldc + col;
                    if constexpr (GATHER && EPI == EPI_STORE)      // gathered 16-bit store: back to the window-major row of the listed window
                        o = ((long)p.gather[row >> 2] * 4 + (row & 3)) * p.ldc + col;
                    if constexpr (EPI == EPI_STORE) {
                        reinterpret_cast<u16*>(p.C)[o] = to_elem<ELEM>(v + bias);
                    } else if constexpr (EPI == EPI_STORE_F32) {
                        reinterpret_cast<float*>(p.C)[o] = v + bias;
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

// f32 output of the 8-wave 256x256 block with the operands SWAPPED in the MFMA (acc = B_frag x A_frag^T): a lane then owns one output
// ROW (m = lane & 31) and four consecutive COLUMNS per register quad, so the tile leaves as 32 16-byte stores per lane instead of the
// 128 4-byte stores of the plain C layout (fc1 over window-major rows: 5 GB of f32 products per launch, K = 1024 - the epilogue is
// as long as the main loop).  Same products, same order of accumulation over k: bit-identical to EPI_STORE_F32.
__device__ __forceinline__ void nt_epilogue_f32t(const NtParams& p, f32x16 (&acc)[4][2], int m0, int n0, int wr, int wc, int lane) {
    const int h = lane >> 5, cl = lane & 31;
    float* out = reinterpret_cast<float*>(p.C);
    if (!out) return;                                  // tools/fc1_windows_microbench.py: the launch without its stores
    const int g_first = p.Cx ? p.wm_goff[p.tile_group[m0 >> 8]] + p.x_first : 0;      // first X row of this tile's group
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wr * 128 + i * 32 + cl;
        if (row >= p.M) continue;
        const bool xrow = p.Cx && row >= g_first;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int col = n0 + wc * 64 + j * 32 + 8 * rq + 4 * h;
                f32x4 v = {acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]};
                if (p.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + col);
                    v += b;
                }
                if (xrow) {                                // a pair-specific row: one of ~7 products added per pair, f16 carries it
                    uint2 hv;
                    hv.x = (unsigned)f32_to_f16_bits(v[0]) | ((unsigned)f32_to_f16_bits(v[1]) << 16);
                    hv.y = (unsigned)f32_to_f16_bits(v[2]) | ((unsigned)f32_to_f16_bits(v[3]) << 16);
                    *reinterpret_cast<uint2*>(p.Cx + (long)row * p.N + col) = hv;
                    continue;
                }
                f32x4* dst = reinterpret_cast<f32x4*>(out + (long)row * p.ldc + col);
#ifdef SGC_EXPERIMENTS      // tools/fc1_windows_microbench.py: tile-contiguous output, cache-policy bits of the store (1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt)
                if (p.nt_store == 8)
                    dst = reinterpret_cast<f32x4*>(out + (((long)(m0 >> 8) * p.tiles_n + (n0 >> 8)) << 16) + ((row - m0) << 8) + (col - n0));
                if (p.nt_store == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
                else if (p.nt_store == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
                else if (p.nt_store == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
                else if (p.nt_store == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
                else
#endif
                *dst = v;
            }
        }
    }
}

// Tile walk order: the tiles_m x tiles_n grid is traversed in 16 x 16 super-tiles (N fastest inside a super-tile), so
// the ~256 blocks in flight cover a square patch of the output: each A row-panel is shared by 16 blocks and each B
// row-panel by 16 blocks through L2 / Infinity Cache.  With the plain "N fastest" order a GEMM with many N tiles
// (fc1 data gradient: 126 x 256 tiles) re-streams the whole B operand from HBM for every M tile (63 GB per launch).
__device__ __forceinline__ void supertile_map(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GS = 16;
    const int a = id / (GS * tiles_n);
    const int hm = min(GS, tiles_m - a * GS);
    const int id2 = id - a * GS * tiles_n;
    const int sn = (tiles_n + GS - 1) / GS;
    const int b = min(id2 / (hm * GS), sn - 1);
    const int id3 = id2 - b * hm * GS;
    const int wn = min(GS, tiles_n - b * GS);
    tm = a * GS + id3 / wn;
    tn = b * GS + id3 % wn;
}

__device__ __forceinline__ void supertile_map_g(int id, int tiles_m, int tiles_n, int GM, int GN, int& tm, int& tn) {
    const int a = id / (GM * tiles_n);
    const int hm = min(GM, tiles_m - a * GM);
    const int id2 = id - a * GM * tiles_n;
    const int sn = (tiles_n + GN - 1) / GN;
    const int b = min(id2 / (hm * GN), sn - 1);
    const int id3 = id2 - b * hm * GN;
    const int wn = min(GN, tiles_n - b * GN);
    tm = a * GM + id3 / wn;
    tn = b * GN + id3 % wn;
}

// LDS-staged epilogue for 16-bit EPI_STORE outputs of the 8-wave 256x256 block (conv3 / fc1 / conv2 data gradients write
// 4-9 GB per launch): the MFMA C layout gives each lane ONE column of 16 rows per tile, i.e. 2-byte global stores.  Here
// each wave transposes its 128x64 sub-tile through its own 18 KiB LDS region (row pitch 144 B): neighbouring lanes swap a
// value (DPP) so that every lane writes one packed dword (two adjacent columns), then the wave reads rows back as 16-byte
// pieces and issues 16 full-line 16-byte stores per lane instead of 128 2-byte ones.  Needs 8 x 16 KiB of LDS.
// Row pitch 128 B (no padding): the read-back is a ds_read_b128 whose 16-lane groups cover rows (r, r+1, r+2, r+3) x half a row -
// with 32 dwords per row the four pieces land on disjoint quarters of the 64 banks; the dword writes of an (even, odd) lane pair
// would fall on one bank two-way (free for a ds_write_b32, but counted): odd rows swap their halves.  Rounds 2-4 padded the rows to 144 B:
// conflict-light writes, but two-way conflicts on every read (SQ_LDS_BANK_CONFLICT 5.6 % of the LDS-active cycles of the window
// data gradient, 2.8 % of fc1's).
constexpr int EPI_LDS_PITCH = 128;
constexpr int EPI_LDS_BYTES = 8 * 128 * EPI_LDS_PITCH;
// MASK (EPI_RELUMASK through this epilogue, round 5): the value is multiplied by p.scale before it is rounded, and the read-back zeroes the
// elements whose forward activation p.mask_src (f16, the output's shape) is not > 0 - 16-byte mask loads beside the 16-byte stores; the
// generic epilogue did this with a 2-byte load and a 2-byte store per element (fc2's data gradient: 0.50 ms for 0.5 GB).  Same values.
template <int ELEM, bool MASK = false>
__device__ __forceinline__ void nt_epilogue_store16(const NtParams& p, f32x16 (&acc)[4][2], int m0, int n0, int wr, int wc, int lane,
                                                    int wid, char* smem) {
    __syncthreads();                                   // every wave is done reading operand tiles
    char* reg = smem + wid * (128 * EPI_LDS_PITCH);
    const int h = lane >> 5, cl = lane & 31;
    const bool odd = cl & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + cl;
            const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                float v0 = acc[i][j][2 * rp] + bias, v1 = acc[i][j][2 * rp + 1] + bias;
                if constexpr (MASK) { v0 *= p.scale; v1 *= p.scale; }
                const float recv = __shfl_xor(odd ? v0 : v1, 1);
                const int r0 = ((2 * rp) & 3) + 8 * ((2 * rp) >> 2) + 4 * h;       // row of register 2rp; register 2rp+1 is r0+1
                const float flo = odd ? recv : v0, fhi = odd ? v1 : recv;
                unsigned packed;
                if constexpr (ELEM == ELEM_BF16) packed = f32x2_to_bf16x2_bits(flo, fhi);      // one v_cvt_pk_bf16_f32
                else packed = (unsigned)to_elem<ELEM>(flo) | ((unsigned)to_elem<ELEM>(fhi) << 16);
                const int row = i * 32 + r0 + (odd ? 1 : 0);
                // odd rows keep their two 64-byte halves swapped: the dwords of an (even, odd) lane pair - rows R and R + 1, same column -
                // then fall on different banks (conflict-free writes), and the read-back below stays conflict-free
                *reinterpret_cast<unsigned*>(reg + row * EPI_LDS_PITCH + (((j * 32 + (cl & ~1)) * 2) ^ ((row & 1) << 6))) = packed;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    u16* out = reinterpret_cast<u16*>(p.C);
    const int c8 = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rowl = it * 8 + rsub;
        uint4 v = *reinterpret_cast<const uint4*>(reg + rowl * EPI_LDS_PITCH + ((c8 * 16) ^ ((rowl & 1) << 6)));
        const int row = m0 + wr * 128 + rowl;
        if (row < p.M) {
            const long o = (long)row * p.ldc + n0 + wc * 64 + c8 * 8;
            if constexpr (MASK) {
                const uint4 f = *reinterpret_cast<const uint4*>(p.mask_src + o);
                const u16* fh = reinterpret_cast<const u16*>(&f);
                u16* vh = reinterpret_cast<u16*>(&v);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (!(from_elem<ELEM_F16>(fh[k]) > 0.f)) vh[k] = 0;
            }
            *reinterpret_cast<uint4*>(out + o) = v;
        }
    }
}

// Block configuration: WR x WC wavefronts, each owning a (TM*32) x (TN*32) output tile.
//   small: 2x2 waves of 64x64   -> 128x128 block, 64 KiB LDS, 2 blocks/CU  (small problems, N % 256 != 0)
//   big  : 2x4 waves of 128x64  -> 256x256 block, 128 KiB LDS, 1 block/CU  (half the LDS bytes per MFMA)
// ABL (test hook only): 0 = normal; 1 = no global loads inside the K loop; 2 = no loads and no barriers;
// 3 = loads and barriers only (no LDS reads / MFMA).  Used by tools/gemm_microbench.py to attribute time.
template <int ELEM, int AMODE, int EPI, int WR, int WC, int TM, int TN, int ABL = 0>
__global__ __launch_bounds__(WR * WC * 64, 2) void gemm_nt_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WR * WC;
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int AI = BM / (8 * NW), BI = BN / (8 * NW);      // global_load_lds instructions per wave per K tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    supertile_map(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    int Mlim = p.M;
    if constexpr (AMODE == AMODE_CONV_GATHER) {
        Mlim = min(p.M, 4 * *p.gather_n);
        if (m0 >= Mlim) return;                     // the grid is sized for the bound, the list is usually shorter
    }

    // ---- loader addresses: wave w stages rows w*AI*8 .. of A and w*BI*8 .. of B, 8 rows (1 KiB) per instruction
    const int lrow = lane >> 3, cpos = lane & 7;
    const u16* a_ptr[AI];
    const u16* b_ptr[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = wid * AI * 8 + i * 8 + lrow;
        const int chunk = cpos ^ ((row >> 1) & 7);
        int m = m0 + row; if (m > Mlim - 1) m = Mlim - 1;
        if constexpr (AMODE == AMODE_CONV_GATHER) m = p.gather[m >> 2] * 4 + (m & 3);
        if constexpr (AMODE != AMODE_PLAIN) a_ptr[i] = p.A + conv_row_base(m, p.lgS, p.Cin) + chunk * 8;
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
        if constexpr (AMODE != AMODE_PLAIN) {
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
        if (ABL != 5 || (kt % 9) == 0) {       // ABL 5: A staged on one K tile in nine (timing model of halo reuse)
#pragma unroll
            for (int i = 0; i < AI; ++i)
                __builtin_amdgcn_global_load_lds(GLB_PTR(a_ptr[i] + aoff), LDS_PTR(abase + i * 1024), 16, 0, 0);
        }
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
        if (ABL != 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if ((ABL == 0 || ABL == 3 || ABL == 5) && kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* ab = smem + (kt & 1) * BUF_BYTES;
        const char* bb = ab + A_BYTES;
        if (ABL == 3) continue;
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

    if constexpr (EPI == EPI_STORE && TM == 4 && TN == 2 && WR * WC == 8) {
        if (p.epi_lds) { nt_epilogue_store16<ELEM>(p, acc, m0, n0, wr, wc, lane, wid, smem); return; }
    }
    if constexpr (AMODE == AMODE_CONV_GATHER) nt_epilogue<ELEM, EPI, TM, TN, true>(p, acc, m0, n0, wr, wc, lane, Mlim);
    else nt_epilogue<ELEM, EPI, TM, TN>(p, acc, m0, n0, wr, wc, lane);
}

#ifdef SGC_EXPERIMENTS
inline int sgc_gemm_ring() { return sgc_tuning().gemm_ring; }      // 4-stage ring kernel (experiment; default off, see below)

#endif

inline int sgc_gemm_cfg() { return sgc_tuning().gemm_cfg; }        // forced block configuration (experiments only), common.h

template <int ELEM, int AMODE, int EPI, int WR, int WC, int TM, int TN, int ABL = 0>
static int launch_gemm_nt_cfg(NtParams p, hipStream_t stream) {
    constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
    constexpr int LDS0 = 2 * (BM + BN) * 128;
    constexpr int LDS = (EPI == EPI_STORE && TM == 4 && TN == 2 && WR * WC == 8 && LDS0 < EPI_LDS_BYTES) ? EPI_LDS_BYTES : LDS0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.N / BN;
    auto kern = gemm_nt_kernel<ELEM, AMODE, EPI, WR, WC, TM, TN, ABL>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(WR * WC * 64), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}


#ifdef SGC_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------
// 4-stage ring variant of the 256x256 block (BK = 32 per stage, 32 KiB per stage, 128 KiB LDS): global_load_lds for
// stage kt+3 is issued while stage kt is being multiplied, a COUNTED s_waitcnt vmcnt leaves two stages in flight
// across the raw s_barrier (a __syncthreads() would drain them).  Measured motivation (tools/gemm_microbench.py,
// 32768x4096x8192 bf16): with the 2-stage loop the load round trip of one 64 KiB stage is 1.36 us against 1.5 us of
// LDS-read + MFMA work per stage and the two overlap poorly (2.1 us per stage); the ring keeps 64-96 KiB of loads in
// flight per CU.  RESULT: correct (tests run it with SGC_GEMM_CFG=3) but 5-10 % SLOWER than the 2-stage loop (995 vs
// 1105 TFLOP/s on that GEMM; conv3 fwd 70.5 vs 69.0 ms): the load path is bandwidth- not latency-bound (loads-only
// ablation: 12.4 TB/s of L2->LDS traffic), so deeper prefetch buys nothing and BK=32 doubles the barrier count.
// Kept off by default as a documented experiment.  64-byte rows: chunk swizzle c ^ ((row>>2)&3) keeps the ds_read_b128 lane groups conflict-free.
template <int ELEM, int AMODE, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WR = 2, WC = 4, TM = 4, TN = 2, NS = 4;
    constexpr int BM = 256, BN = 256;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, BUF_BYTES = A_BYTES + B_BYTES;   // 32 KiB per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = blockIdx.x % p.tiles_n, tm = blockIdx.x / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // loader: one instruction = 16 rows x 64 B; wave w stages rows w*32 .. w*32+31 of A and of B (2 + 2 instructions)
    const int lrow = lane >> 2, cpos = lane & 3;
    const u16* a_ptr[2];
    const u16* b_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wid * 32 + i * 16 + lrow;
        const int chunk = cpos ^ ((row >> 2) & 3);
        int m = m0 + row; if (m > p.M - 1) m = p.M - 1;
        if constexpr (AMODE == AMODE_CONV) a_ptr[i] = p.A + conv_row_base(m, p.lgS, p.Cin) + chunk * 8;
        else a_ptr[i] = p.A + (long)m * p.lda + chunk * 8;
        b_ptr[i] = p.B + (long)(n0 + row) * p.ldb + chunk * 8;
    }
    const int Wp = (1 << p.lgS) + 2;
    auto stage = [&](int buf, int kt) {            // kt counts 32-wide K tiles
        long aoff;
        if constexpr (AMODE == AMODE_CONV) {
            const int k64 = kt >> 1;
            const int cc = k64 / 9, tap = k64 - cc * 9;
            const int ky = tap / 3, kx = tap - 3 * ky;
            aoff = (long)(ky * Wp + kx) * p.Cin + (cc << 6) + ((kt & 1) << 5);
        } else {
            aoff = (long)kt << 5;
        }
        const long boff = (long)kt << 5;
        char* abase = smem + buf * BUF_BYTES + wid * 2048;
        char* bbase = abase + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_ptr[i] + aoff), LDS_PTR(abase + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_ptr[i] + boff), LDS_PTR(bbase + i * 1024), 16, 0, 0);
        }
    };

    const int wr = wid / WC, wc = wid % WC;
    const int kh = lane >> 5;
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ra = wr * TM * 32 + i * 32 + (lane & 31);
        a_off[i] = ra * 64; a_sw[i] = (ra >> 2) & 3;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int rb = wc * TN * 32 + i * 32 + (lane & 31);
        b_off[i] = rb * 64; b_sw[i] = (rb >> 2) & 3;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K >> 5;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 2) stage(2, 2);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt must have landed; the (up to) two younger stages stay in flight: 4 loads per stage per lane
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // every wave's share of stage kt is in LDS; stage kt-1 is fully consumed
        if (kt + 3 < nk) stage((kt + 3) & 3, kt + 3);
        const char* ab = smem + (kt & 3) * BUF_BYTES;
        const char* bb = ab + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
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
    nt_epilogue<ELEM, EPI, TM, TN>(p, acc, m0, n0, wr, wc, lane);
}

template <int ELEM, int AMODE, int EPI>
static int launch_gemm_nt_ring(NtParams p, hipStream_t stream) {
    constexpr int LDS = 4 * 512 * 64;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 256;
    auto kern = gemm_nt_ring_kernel<ELEM, AMODE, EPI>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
#endif  // SGC_EXPERIMENTS


// ---------------------------------------------------------------------------------------------------------------
// Halo-staged implicit 3x3 convolution for 16x16 maps (conv3 forward and data gradient): one workgroup = one image
// (256 pixels) x 256 output channels.  For every 64-channel chunk the 18x18 zero-padded patch of the image is staged
// in LDS ONCE (41 KiB) and the nine taps read their A fragments from it at shifted pixel rows; only the weight tile
// (32 KiB) is staged per (chunk, tap).  L2->LDS traffic per K step falls from 64 KiB to 36.6 KiB (-43 %), which is
// what limits the plain implicit GEMM (tools/gemm_microbench.py: the load stream and the LDS-read+MFMA loop overlap
// poorly; staging A on one K tile in nine was measured +10 %).
// Patch row r = py*18 + px (128 B per row); 16-B chunk swizzle c ^ f(py,px), f = ((px>>1) + 4*(py&1)) & 7, keeps every
// ds_read_b128 lane group (pixels of windows {0,3,5,6} / {1,2,4,7} on two image rows) conflict-free for all nine taps.
template <int ELEM, int EPI>
__global__ __launch_bounds__(512, 2) void conv16_halo_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WC = 4, TM = 4, TN = 2;
    constexpr int A_BYTES = 328 * 128;            // 324 patch rows, padded to 41 x 8 rows
    constexpr int B_BYTES = 256 * 128;
    char* const abuf0 = smem;
    char* const bbuf0 = smem + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = blockIdx.x % p.tiles_n, img = blockIdx.x / p.tiles_n;
    const int m0 = img * 256, n0 = tn * 256;
    const int Cin = p.Cin;

    // ---- A patch loader: 41 instructions of 8 rows; wave w issues instructions w, w+8, ... (<= 6)
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
    const u16* const a_img = p.A + (long)img * (324L * Cin);
    auto stage_a = [&](int buf, int cc) {
        char* base = abuf0 + buf * A_BYTES;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int t = wid + 8 * i;
            if (t < 41)
                __builtin_amdgcn_global_load_lds(GLB_PTR(a_img + a_off[i] + (cc << 6)), LDS_PTR(base + t * 1024), 16, 0, 0);
        }
    };
    // ---- B (weight) loader: as in the plain kernel
    const u16* b_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wid * 32 + i * 8 + lrow;
        b_ptr[i] = p.B + (long)(n0 + row) * p.ldb + ((cpos ^ ((row >> 1) & 7)) << 3);
    }
    auto stage_b = [&](int buf, int step) {
        char* base = bbuf0 + buf * B_BYTES + wid * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_ptr[i] + ((long)step << 6)), LDS_PTR(base + i * 1024), 16, 0, 0);
    };

    // ---- fragment addressing
    const int wr = wid / WC, wc = wid % WC;
    const int kh = lane >> 5;
    int a_row[TM], a_px[TM], a_py[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wr * 128 + i * 32 + (lane & 31);
        const int W = m >> 2, q = m & 3;
        a_py[i] = 2 * (W >> 3) + (q >> 1);
        a_px[i] = 2 * (W & 7) + (q & 1);
        a_row[i] = a_py[i] * 18 + a_px[i];
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

    const int ncc = Cin >> 6;
    const int nsteps = ncc * 9;
    stage_a(0, 0);
    stage_b(0, 0);
    int step = 0;
    for (int cc = 0; cc < ncc; ++cc) {
        const char* ab = abuf0 + (cc & 1) * A_BYTES;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++step) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step + 1 < nsteps) stage_b((step + 1) & 1, step + 1);
            if (tap == 0 && cc + 1 < ncc) stage_a((cc + 1) & 1, cc + 1);
            const char* bb = bbuf0 + (step & 1) * B_BYTES;
            const int ky = tap / 3, kx = tap - 3 * ky;
            int ar[TM], af_sw[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ar[i] = (a_row[i] + ky * 18 + kx) * 128;
                af_sw[i] = (((a_px[i] + kx) >> 1) + 4 * ((a_py[i] + ky) & 1)) & 7;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = ks * 2 + kh;
                s16x8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const s16x8*>(ab + ar[i] + ((c ^ af_sw[i]) << 4));
#pragma unroll
                for (int i = 0; i < TN; ++i) bf[i] = *reinterpret_cast<const s16x8*>(bb + b_off[i] + ((c ^ b_sw[i]) << 4));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<ELEM>(af[i], bf[j], acc[i][j]);
            }
        }
    }
    if constexpr (EPI == EPI_STORE) {
        if (p.epi_lds) { nt_epilogue_store16<ELEM>(p, acc, m0, n0, wr, wc, lane, wid, smem); return; }
    }
    nt_epilogue<ELEM, EPI, TM, TN>(p, acc, m0, n0, wr, wc, lane);
}

template <int ELEM, int EPI>
static int launch_conv16_halo(NtParams p, hipStream_t stream) {
    constexpr int LDS = 2 * 328 * 128 + 2 * 256 * 128;
    p.tiles_m = p.M / 256;
    p.tiles_n = p.N / 256;
    auto kern = conv16_halo_kernel<ELEM, EPI>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SGC_LAUNCH(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), LDS, stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

#include "gemm_nt_pp.h"
#ifdef SGC_EXPERIMENTS      // main-loop variants that were measured and rejected (profiles/README.md); SGC_EXPERIMENTS=1 builds them
#include "gemm_nt_w4.h"     // for tools/gemm_microbench.py - they are not part of the product library
#include "gemm_nt_pp1.h"
#endif

inline int sgc_gemm_pp() { return sgc_tuning().gemm_pp; }          // ping-pong 256x256 loops (0: the 2-stage loops)
inline int sgc_conv_halo() { return sgc_tuning().conv_halo; }      // halo-staged implicit convolution (0: plain implicit GEMM)

template <int ELEM, int AMODE, int EPI>
static int launch_gemm_nt(NtParams p, hipStream_t stream) {
    if (p.M <= 0) return SGC_OK;
    if ((p.K & 63) || (p.N & 127) || p.K <= 0) return SGC_ERR_ARG;
    if (AMODE == AMODE_CONV && ((p.Cin & 63) || p.K != 9 * p.Cin)) return SGC_ERR_ARG;
    const int cfg = sgc_gemm_cfg();
    p.epi_lds = sgc_tuning().epi_lds;      // 0 keeps the direct 2-byte stores
    const bool big_ok = (p.N % 256) == 0;
    if constexpr (AMODE == AMODE_CONV && (EPI == EPI_POOL || EPI == EPI_STORE)) {
        if (p.lgS == 4 && big_ok && (p.M % 256) == 0 && (cfg == 4 || cfg == 7 || (cfg == 0 && sgc_conv_halo() && (long)p.M * p.N >= 256L * 256 * 256))) {
            if (cfg == 7 || (cfg == 0 && sgc_gemm_pp())) return launch_conv16_halo_pp<ELEM, EPI>(p, stream);
            return launch_conv16_halo<ELEM, EPI>(p, stream);
        }
    }
    const bool big = big_ok && (cfg == 2 || cfg == 3 || (cfg == 0 && (long)p.M * p.N >= 256L * 256 * 256));
#ifdef SGC_EXPERIMENTS
    if (big && (cfg == 3 || (cfg == 0 && sgc_gemm_ring()))) return launch_gemm_nt_ring<ELEM, AMODE, EPI>(p, stream);
#endif
    if constexpr (AMODE == AMODE_PLAIN) {
        if (big_ok && (cfg == 5 || (big && cfg == 0 && sgc_gemm_pp()))) return launch_gemm_nt_pp<ELEM, EPI>(p, stream);
    }
    if (big) return launch_gemm_nt_cfg<ELEM, AMODE, EPI, 2, 4, 4, 2>(p, stream);
    return launch_gemm_nt_cfg<ELEM, AMODE, EPI, 2, 2, 2, 2>(p, stream);
}
