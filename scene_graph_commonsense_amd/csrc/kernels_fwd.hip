// Forward-path HBM-bound kernels of the pairwise relation head (gfx950) and their C-ABI launchers.
// Layout conventions (all channels-last, 16-bit):
//   x        [n_img*1024][XC=384]   packed cat(feature, depth), pixel-major rows (y*32+x), zero padded channels
//   a_img    [n_img*1024][128]      tanh(conv1) per role
//   a_pad    [n_obj][34][34][128]   per-object masked map, zero border (conv2 halo)
//   U, V     [n_obj*1024][512]      per-object conv2 halves, rows window-major (m = 4*(Y*16+X) + dy*2+dx)
//   z_pad    [n_pair][18][18][512]  relu/maxpool(U_i + V_j), zero border (conv3 halo)
//   y        [n_pair*64][1024]      conv3 output after relu+pool, rows = pooled window py*8+px
#include "common.h"
#include "gemm_nt.h"

// ------------------------------------------------------------------------------------------------ pack
// cat(feature, depth) f32 NCHW -> f16 [img*HW][XC] (reference train_test.py:194-195 builds the same concat).
__global__ __launch_bounds__(256) void pack_nhwc_kernel(const float* __restrict__ f0, int C0, const float* __restrict__ f1,
                                                        int C1, u16* __restrict__ out, int HW, int XC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16* tile = reinterpret_cast<u16*>(smem);          // [64][XC+8]
    const int ldt = XC + 8;
    const int img = blockIdx.y, p0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = w; c < XC; c += 4) {
        float v = 0.f;
        if (c < C0) v = f0[((long)img * C0 + c) * HW + p0 + lane];
        else if (c < C0 + C1) v = f1[((long)img * C1 + (c - C0)) * HW + p0 + lane];
        tile[lane * ldt + c] = f32_to_f16_bits(v);
    }
    __syncthreads();
    const int chunks = XC / 8;
    for (int i = threadIdx.x; i < 64 * chunks; i += 256) {
        const int r = i / chunks, ch = i - r * chunks;
        const uint4 v = *reinterpret_cast<const uint4*>(tile + r * ldt + ch * 8);
        *reinterpret_cast<uint4*>(out + ((long)img * HW + p0 + r) * XC + ch * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------ masks
// a_pad[o][py][px][:] = border ? 0 : (inside bbox ? a_img[img[o]][y*F+x][:] : cst[:])
// (reference train_test.py:166-168 mask build + :194-195 multiply, with tanh(conv1(0)) = tanh(bias) outside.)
__global__ __launch_bounds__(256) void mask_objects_kernel(const u16* __restrict__ a_img, const int* __restrict__ obj_img,
                                                           const int* __restrict__ bbox, const u16* __restrict__ cst,
                                                           u16* __restrict__ a_pad, int n_obj, int F, int D) {
    const int P = F + 2;
    const int cpp = D / 8;                              // 16-byte chunks per pixel
    const long total = (long)n_obj * P * P * cpp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpp);
        long r = i / cpp;
        const int px = (int)(r % P); r /= P;
        const int py = (int)(r % P);
        const int o = (int)(r / P);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (px > 0 && px <= F && py > 0 && py <= F) {
            const int x = px - 1, y = py - 1;
            const int* b = bbox + 4 * o;                // x0,x1,y0,y1 (already normalised to slice semantics)
            if (x >= b[0] && x < b[1] && y >= b[2] && y < b[3])
                v = *reinterpret_cast<const uint4*>(a_img + ((long)obj_img[o] * F * F + y * F + x) * D + ch * 8);
            else
                v = *reinterpret_cast<const uint4*>(cst + ch * 8);
        }
        *reinterpret_cast<uint4*>(a_pad + i * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------ expansion
// z_pad[p][Y+1][X+1][c] = max_q relu(U[sub[p]][4W+q][c] + V[obj[p]][4W+q][c]),  W = Y*16+X
// (reference model.py:141-144: cat -> conv2_1 -> relu -> maxpool, with conv2_1(cat(a,b)) = U + V.)
// One wavefront per (pair, window): lane l owns channels 8l..8l+7 (16-byte loads/stores, 1 KiB per wave row).
template <int ELEM_OUT>
__global__ __launch_bounds__(256) void pair_expand_kernel(const u16* __restrict__ U, const u16* __restrict__ V,
                                                          const int* __restrict__ sub, const int* __restrict__ obj,
                                                          u16* __restrict__ z, long n_items) {
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        const int p = (int)(it >> 8), W = (int)(it & 255);
        const u16* up = U + ((long)sub[p] * 1024 + 4 * W) * 512 + lane * 8;
        const u16* vp = V + ((long)obj[p] * 1024 + 4 * W) * 512 + lane * 8;
        float best[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) best[k] = 0.f;      // relu floor
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4*>(up + q * 512);
            const uint4 b = *reinterpret_cast<const uint4*>(vp + q * 512);
            const u16* ah = reinterpret_cast<const u16*>(&a);
            const u16* bh = reinterpret_cast<const u16*>(&b);
#pragma unroll
            for (int k = 0; k < 8; ++k) best[k] = fmaxf(best[k], f16_bits_to_f32(ah[k]) + f16_bits_to_f32(bh[k]));
        }
        uint4 o;
        u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
        for (int k = 0; k < 8; ++k) oh[k] = to_elem<ELEM_OUT>(best[k]);
        const int Y = W >> 4, X = W & 15;
        *reinterpret_cast<uint4*>(z + (((long)p * 18 + Y + 1) * 18 + X + 1) * 512 + lane * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------ dense expansion
// All ordered pairs of one image at once, written as FULL channel rows.  One workgroup owns (image, window W, tile of JT
// object-role objects j): it stages the four source pixels x 512 channels of V_j for its JT objects in LDS (JT x 4 KiB);
// every wavefront then takes subjects i = wave, wave + 8, ...: it keeps the four pixels of U_i in registers (lane l owns
// channels 8l..8l+7) and streams the V_j from LDS, so that each store instruction of the wave writes one complete
// 1 KiB row z[p][y][x][0..511] (and, for training, the bf16 copy and the 512 B of relu/maxpool routing bytes).
// HBM traffic is the algorithmic minimum for z; U quads are re-read once per object tile through L2 (n x 4 KiB per block,
// 2 GB per step at N=64, B=8).  The first version of this kernel owned (window, 64-channel chunk) and wrote 128-byte
// pieces of 32 different pairs per store instruction: 3.0 TB/s; full rows reach the rate of the un-pool kernel.
// u + (float)half of ``packed`` as one instruction (v_fma_mix_f32: every source is an f32 or one half of a dword; 1.0 x V + U)
__device__ __forceinline__ float add_f16lo_f32(unsigned packed, float u) {
    float r;
    asm("v_fma_mix_f32 %0, 1.0, %1, %2 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(packed), "v"(u));
    return r;
}
__device__ __forceinline__ float add_f16hi_f32(unsigned packed, float u) {
    float r;
    asm("v_fma_mix_f32 %0, 1.0, %1, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(packed), "v"(u));
    return r;
}
__device__ __forceinline__ float max3_f32(float a, float b, float c) {          // (fmaxf would first canonicalise every operand: + 1 instruction each)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
constexpr int EXPAND_JT = 16;
__global__ __launch_bounds__(512) void pair_expand_dense_kernel(const u16* __restrict__ U, const u16* __restrict__ V,
                                                                const int* __restrict__ img_ptr, const int* __restrict__ pid,
                                                                int pid_ld, u16* __restrict__ z, u16* __restrict__ zb,
                                                                unsigned char* __restrict__ amz, const int* __restrict__ pixrect) {
    __shared__ __attribute__((aligned(16))) char sv[EXPAND_JT * 4096];
    const int W = blockIdx.x, jt = blockIdx.y, img = blockIdx.z;
    const int o0 = img_ptr[img], n = img_ptr[img + 1] - o0;
    const int j0 = jt * EXPAND_JT;
    if (j0 >= n) return;
    const int nj = min(EXPAND_JT, n - j0);
    // stage V: object jj -> 4 pixels x 1 KiB, contiguous in global memory (window-major pixel order)
    for (int it = threadIdx.x; it < nj * 256; it += 512) {
        const int jj = it >> 8, part = it & 255;
        const u16* src = V + ((long)(o0 + j0 + jj) * 1024 + 4 * W) * 512 + part * 8;
        *reinterpret_cast<uint4*>(sv + jj * 4096 + part * 16) = *reinterpret_cast<const uint4*>(src);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Y = W >> 4, X = W & 15;
    const long zoff = ((long)(Y + 1) * 18 + X + 1) * 512 + lane * 8;
    const long aoff = (long)W * 256 + lane * 4;
    // the U quad of the wave's NEXT subject is requested before the current one is expanded: the 16-object store loop covers its latency
    uint4 unext[4];
    if (wid < n) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            unext[q] = *reinterpret_cast<const uint4*>(U + ((long)(o0 + wid) * 1024 + 4 * W + q) * 512 + lane * 8);
    }
    for (int i = wid; i < n; i += 8) {
        float uf[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u16* ah = reinterpret_cast<const u16*>(&unext[q]);
#pragma unroll
            for (int k = 0; k < 8; ++k) uf[q][k] = f16_bits_to_f32(ah[k]);
        }
        if (i + 8 < n) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                unext[q] = *reinterpret_cast<const uint4*>(U + ((long)(o0 + i + 8) * 1024 + 4 * W + q) * 512 + lane * 8);
        }
        const int* prow = pid + (long)(o0 + i) * pid_ld + j0;
        // lane jj looks up pair jj of the tile: not requested (diagonal) or - with pixrect - a pair that does not need this pixel
        int pl = -1;
        if (lane < nj) {
            pl = prow[lane];
            if (pl >= 0 && pixrect && !in_pixel_rect(pixrect[pl], Y, X)) pl = -1;
        }
        unsigned long long todo = __ballot(pl >= 0);
        while (todo) {
            const int jj = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1;
            const int p = __builtin_amdgcn_readlane(pl, jj);
            float best[8];
            unsigned arg[8];                                // 32-bit: as bytes the compiler spends ~145 sub-dword instructions per item on them
#pragma unroll
            for (int k = 0; k < 8; ++k) { best[k] = 0.f; arg[k] = 4; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 b = *reinterpret_cast<const uint4*>(sv + jj * 4096 + q * 1024 + lane * 16);
                const u16* bh = reinterpret_cast<const u16*>(&b);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float sm = uf[q][k] + f16_bits_to_f32(bh[k]);
                    if (sm > best[k]) { best[k] = sm; arg[k] = (unsigned)q; }
                }
            }
            if (z) {
                uint4 o;
                u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
                for (int k = 0; k < 8; ++k) oh[k] = f32_to_f16_bits(best[k]);
                *reinterpret_cast<uint4*>(z + (long)p * (18 * 18 * 512) + zoff) = o;
            }
            if (zb) {
                uint4 o;
                u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
                for (int k = 0; k < 8; ++k) oh[k] = f32_to_bf16_bits(best[k]);
                *reinterpret_cast<uint4*>(zb + (long)p * (18 * 18 * 512) + zoff) = o;
            }
            if (amz) {                                  // two 4-bit routing codes per byte: channel 2k low, 2k+1 high
                unsigned ao = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) ao |= arg[k] << (4 * k);
                *reinterpret_cast<unsigned*>(amz + (long)p * (256 * 256) + aoff) = ao;
            }
        }
    }
}

// The same launch with the live (subject, object) combinations of a workgroup LISTED first (round 5).  On the shared-window path only
// ~15 % of the (pair, pixel) items are live (the pixel rectangles), and the loop above spent most of its time finding them: every
// wavefront walked all subjects of the image, loaded each one's four U rows whether or not a partner was live, and reached each
// partner's rectangle through two dependent global loads (pair index, then its rectangle) inside the loop; the matrix of live items
// is also very uneven across the eight wavefronts.  Here, per chunk of 64 subjects: (1) all 512 threads test their two (subject,
// object) combinations and the survivors are compacted - in (subject, object) order - into LDS with the per-subject run ends; (2) the
// list is cut into eight equal pieces, one per wavefront; a wavefront loads U rows only for subjects that have a live item and asks
// for the next subject's rows when it starts on the current one.  Same arithmetic per item: same bits.
// (4 waves per SIMD = two workgroups per CU: without the bound the scheduler keeps all 32 sums of an item live - 158 registers, one
//  workgroup per CU, and the launch ran 0.92 instead of 0.82 ms with FEWER instructions)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void pair_expand_dense_list_kernel(const u16* __restrict__ U, const u16* __restrict__ V,
                                                                     const int* __restrict__ img_ptr, const int* __restrict__ pid,
                                                                     int pid_ld, u16* __restrict__ z, u16* __restrict__ zb,
                                                                     unsigned char* __restrict__ amz, const int* __restrict__ pixrect) {
    __shared__ __attribute__((aligned(16))) char sv[EXPAND_JT * 4096];
    __shared__ unsigned short live[1024];                 // (subject within the chunk) << 4 | object within the tile
    __shared__ int lpair[1024];
    __shared__ int subj_cnt[64], subj_end[64];
    const int W = blockIdx.x, jt = blockIdx.y, img = blockIdx.z;
    const int o0 = img_ptr[img], n = img_ptr[img + 1] - o0;
    const int j0 = jt * EXPAND_JT;
    if (j0 >= n) return;
    const int nj = min(EXPAND_JT, n - j0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Y = W >> 4, X = W & 15;
    const long zoff = ((long)(Y + 1) * 18 + X + 1) * 512 + lane * 8;
    const long aoff = (long)W * 256 + lane * 4;
    bool staged = false;
    for (int ibase = 0; ibase < n; ibase += 64) {
        // ---- (1) the chunk's live combinations, c = subject * 16 + object, two per thread (c = tid, tid + 512)
        int pl[2];
        unsigned long long bal[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = h * 512 + tid, il = c >> 4, jj = c & 15, i = ibase + il;
            pl[h] = -1;
            if (i < n && jj < nj) {
                pl[h] = pid[(long)(o0 + i) * pid_ld + j0 + jj];
                if (pl[h] >= 0 && pixrect && !in_pixel_rect(pixrect[pl[h]], Y, X)) pl[h] = -1;
            }
            bal[h] = __ballot(pl[h] >= 0);
            if ((lane & 15) == 0) subj_cnt[h * 32 + wid * 4 + (lane >> 4)] = __popcll((bal[h] >> lane) & 0xFFFFull);
        }
        __syncthreads();
        if (wid == 0) {                                   // inclusive scan of the 64 per-subject counts
            int v = subj_cnt[lane];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(v, d);
                if (lane >= d) v += t;
            }
            subj_end[lane] = v;
        }
        __syncthreads();
        const int total = subj_end[63];
        if (total == 0) continue;                         // uniform: nothing of this chunk touches the pixel
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (pl[h] >= 0) {
                const int c = h * 512 + tid, s4 = (h * 8 + wid) * 4;          // first subject of this wavefront's 64 combinations
                const int pos = (s4 ? subj_end[s4 - 1] : 0) +
                                (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal[h] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[h], 0u));
                live[pos] = (unsigned short)c;
                lpair[pos] = pl[h];
            }
        }
        if (!staged) {                                    // V of the tile's objects: 4 pixels x 1 KiB each, once per workgroup
            for (int it = tid; it < nj * 256; it += 512) {
                const int jj = it >> 8, part = it & 255;
                const u16* src = V + ((long)(o0 + j0 + jj) * 1024 + 4 * W) * 512 + part * 8;
                *reinterpret_cast<uint4*>(sv + jj * 4096 + part * 16) = *reinterpret_cast<const uint4*>(src);
            }
            staged = true;
        }
        __syncthreads();
        // ---- (2) an eighth of the list per wavefront
        const int per = (total + 7) >> 3;
        int k = wid * per;
        const int kend = min(total, k + per);
        if (k >= kend) continue;
        int il = __builtin_amdgcn_readfirstlane((int)live[k]) >> 4;
        uint4 unext[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            unext[q] = *reinterpret_cast<const uint4*>(U + ((long)(o0 + ibase + il) * 1024 + 4 * W + q) * 512 + lane * 8);
        while (k < kend) {
            float uf[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u16* ah = reinterpret_cast<const u16*>(&unext[q]);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) uf[q][kk] = f16_bits_to_f32(ah[kk]);
            }
            const int run_end = min(kend, __builtin_amdgcn_readfirstlane(subj_end[il]));
            if (run_end < kend) {                         // the next subject with a live item: its rows are on their way during this run
                il = __builtin_amdgcn_readfirstlane((int)live[run_end]) >> 4;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    unext[q] = *reinterpret_cast<const uint4*>(U + ((long)(o0 + ibase + il) * 1024 + 4 * W + q) * 512 + lane * 8);
            }
            for (; k < run_end; ++k) {
                const int jj = __builtin_amdgcn_readfirstlane((int)live[k]) & 15;
                const int p = __builtin_amdgcn_readfirstlane(lpair[k]);
                float best[8];
                unsigned arg[8];                                // 32-bit: as bytes the compiler spends ~145 sub-dword instructions per item on them
                {
                    // Round 6: 195 -> ~140 vector instructions per item, same bits.  (1) the sum U + V as ONE v_fma_mix_f32 (1.0 x the f16
                    // half of V's dword + the f32 U: the conversion is exact, the sum is rounded once - what v_cvt_f32_f16 + v_add_f32
                    // gave); (2) the maximum as two v_max3_f32 instead of four compare + select pairs; (3) the route = the FIRST pixel whose
                    // sum equals the maximum, 4 when nothing is positive - what the sequential "sm > best" chain selects.
                    uint4 bq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const uint4*>(sv + jj * 4096 + q * 1024 + lane * 16);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        float sm[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned w = reinterpret_cast<const unsigned*>(&bq[q])[kk >> 1];
                            sm[q] = (kk & 1) ? add_f16hi_f32(w, uf[q][kk]) : add_f16lo_f32(w, uf[q][kk]);
                        }
                        const float m = max3_f32(max3_f32(0.f, sm[0], sm[1]), sm[2], sm[3]);
                        const bool alive = m > 0.f;
                        best[kk] = m + 0.f;                     // m >= 0; a maximum of -0.0 and +0.0 may come back as either: -0 + 0 = +0
                        unsigned a = 3u;                        // a chain of selects, last write wins: the first pixel that holds the maximum
                        a = (sm[2] == m) ? 2u : a;
                        a = (sm[1] == m) ? 1u : a;
                        a = (sm[0] == m) ? 0u : a;
                        arg[kk] = alive ? a : 4u;
                    }
                }
                if (z) {
                    uint4 o;
                    u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) oh[kk] = f32_to_f16_bits(best[kk]);
                    *reinterpret_cast<uint4*>(z + (long)p * (18 * 18 * 512) + zoff) = o;
                }
                if (zb) {
                    uint4 o;
                    u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) oh[kk] = f32_to_bf16_bits(best[kk]);
                    *reinterpret_cast<uint4*>(zb + (long)p * (18 * 18 * 512) + zoff) = o;
                }
                if (amz) {
                    unsigned ao = 0;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) ao |= arg[kk] << (4 * kk);
                    *reinterpret_cast<unsigned*>(amz + (long)p * (256 * 256) + aoff) = ao;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ head
// Hierarchical Bayesian head + candidate reduction (reference model.py:176-184, evaluator.py:160-174).
// Wt is [512][64] f32 (column r = output row r): rows [0,R) fine relations (three segments), R..R+2 super
// logits (fc5), R+3 connectivity (fc4).  One wavefront per pair, lane r owns output row r; segment
// max / sum / argmax are wavefront shuffle reductions.
struct HeadParams {
    const float* p; const float* Wt; const float* bias; int n_pairs;
    int ng, np, ns; int hier; float invT1, invT2, invT3;
    float* rel; float* sup; float* conn; float* cand_conf; int* cand_pred; const unsigned char* iou_mask;
};

__device__ __forceinline__ float wave_max_pred(float v, bool in) {
    float x = in ? v : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ float wave_sum_pred(float v, bool in) {
    float x = in ? v : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

// Four pairs per wavefront pass: a lane keeps p[pair][lane + 64 i] of its four pairs in registers and the 512 products walk k in
// ascending order - p[k] arrives by v_readlane (an SGPR operand of the FMA), W[k][lane] by ONE LDS read shared by the four pairs.
// Round 5: the one-pair loop read p[k] and W[k][lane] from LDS for every FMA (1 024 LDS instructions per pair, one wavefront per SIMD
// waiting on each): 0.29 ms; the sums are the same fmaf chain, bit for bit.
constexpr int HEAD_PAIRS = 4;
__global__ __launch_bounds__(512) void bayes_head_kernel(const HeadParams hp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Wl = reinterpret_cast<float*>(smem);             // [512][64]
    for (int i = threadIdx.x; i < 512 * 64; i += blockDim.x) Wl[i] = hp.Wt[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int R = hp.ng + hp.np + hp.ns;
    const float bias = hp.bias[lane];
    for (int pr0 = (blockIdx.x * nw + w) * HEAD_PAIRS; pr0 < hp.n_pairs; pr0 += gridDim.x * nw * HEAD_PAIRS) {
        int preg[HEAD_PAIRS][8];
#pragma unroll
        for (int j = 0; j < HEAD_PAIRS; ++j) {
            const float* prow = hp.p + (long)min(pr0 + j, hp.n_pairs - 1) * 512;
#pragma unroll
            for (int i = 0; i < 8; ++i) preg[j][i] = __float_as_int(prow[lane + 64 * i]);
        }
        float accs[HEAD_PAIRS];
#pragma unroll
        for (int j = 0; j < HEAD_PAIRS; ++j) accs[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll 8
            for (int l = 0; l < 64; ++l) {
                const float wv = Wl[(i * 64 + l) * 64 + lane];
#pragma unroll
                for (int j = 0; j < HEAD_PAIRS; ++j) accs[j] = fmaf(__int_as_float(__builtin_amdgcn_readlane(preg[j][i], l)), wv, accs[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < HEAD_PAIRS; ++j) {
            const int pr = pr0 + j;
            if (pr >= hp.n_pairs) break;                    // wave-uniform
            const float acc = accs[j] + bias;
            const bool is_conn = lane == (hp.hier ? R + 3 : R);
            if (is_conn) hp.conn[pr] = acc;
            const bool masked = hp.iou_mask && !hp.iou_mask[pr];
            if (hp.hier) {
                const bool in_sup = lane >= R && lane < R + 3;
                const float smax = wave_max_pred(acc, in_sup);
                const float ssum = wave_sum_pred(expf(acc - smax), in_sup);
                const float slog = acc - smax - logf(ssum);             // valid on the three super lanes
                if (in_sup) hp.sup[(long)pr * 3 + (lane - R)] = slog;
                const int seg = lane < hp.ng ? 0 : (lane < hp.ng + hp.np ? 1 : 2);
                const float invT = seg == 0 ? hp.invT1 : (seg == 1 ? hp.invT2 : hp.invT3);
                const float x = acc * invT;
                float out = 0.f;
#pragma unroll
                for (int sg = 0; sg < 3; ++sg) {
                    const bool in = lane < R && seg == sg;
                    const float m = wave_max_pred(x, in);
                    const float sm = wave_sum_pred(expf(x - m), in);
                    const float sl = __shfl(slog, R + sg);
                    const float lp = x - m - logf(sm) + sl;
                    if (in) out = lp;
                    // candidate: max log-prob of the segment and its first argmax (evaluator.py:160-174)
                    const float cm = wave_max_pred(lp, in);
                    const int am = wave_min_int((in && lp == cm) ? lane : 1 << 20);
                    if (lane == 0) {
                        hp.cand_conf[(long)pr * 3 + sg] = masked ? -INFINITY : cm;
                        hp.cand_pred[(long)pr * 3 + sg] = am;
                    }
                }
                if (lane < R) hp.rel[(long)pr * R + lane] = out;
            } else {
                const bool in = lane < R;
                if (in) hp.rel[(long)pr * R + lane] = acc;
                const float cm = wave_max_pred(acc, in);
                const int am = wave_min_int((in && acc == cm) ? lane : 1 << 20);
                if (lane == 0) {
                    hp.cand_conf[pr] = masked ? -INFINITY : cm;
                    hp.cand_pred[pr] = am;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

int sgc_pack_image_nhwc(const float* f0, int C0, const float* f1, int C1, void* x_out, int n_img, int HW, int XC,
                        void* stream) {
    if (HW % 64 || XC % 8 || C0 + C1 > XC) return SGC_ERR_ARG;
    if (n_img <= 0) return SGC_OK;
    SGC_LAUNCH(pack_nhwc_kernel, dim3(HW / 64, n_img), dim3(256), 64 * (XC + 8) * 2, (hipStream_t)stream,
                       f0, C0, f1, C1, (u16*)x_out, HW, XC);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_conv1_tanh(const void* x, const void* w1r, const float* b1, void* a_img, int n_rows, int XC, void* stream) {
    NtParams p{};
    p.A = (const u16*)x; p.B = (const u16*)w1r; p.C = a_img; p.M = n_rows; p.N = 128; p.K = XC;
    p.lda = XC; p.ldb = XC; p.ldc = 128; p.bias = b1;
    return launch_gemm_nt<ELEM_F16, AMODE_PLAIN, EPI_BIAS_TANH>(p, (hipStream_t)stream);
}

int sgc_object_masked_maps(const void* a_img, const int* obj_img, const int* bbox, const void* cst, void* a_pad,
                           int n_obj, int F, int D, void* stream) {
    if (D % 8) return SGC_ERR_ARG;
    if (n_obj <= 0) return SGC_OK;
    const long total = (long)n_obj * (F + 2) * (F + 2) * (D / 8);
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    SGC_LAUNCH(mask_objects_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)a_img, obj_img,
                       bbox, (const u16*)cst, (u16*)a_pad, n_obj, F, D);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// U (or V) [n_obj*1024][512] = conv3x3(a_pad, w2r[512][2][9][64]) (+ bias for the object role)
int sgc_conv2_object(const void* a_pad, const void* w2r, const float* bias, void* out, int n_obj, void* stream) {
    NtParams p{};
    p.A = (const u16*)a_pad; p.B = (const u16*)w2r; p.C = out; p.M = n_obj * 1024; p.N = 512; p.K = 9 * 128;
    p.ldb = 9 * 128; p.ldc = 512; p.lgS = 5; p.Cin = 128; p.bias = bias;
    return launch_gemm_nt<ELEM_F16, AMODE_CONV, EPI_STORE>(p, (hipStream_t)stream);
}

int sgc_pair_expand(const void* U, const void* V, const int* sub_idx, const int* obj_idx, void* z_pad, int n_pairs,
                    int out_elem, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    const long items = (long)n_pairs * 256;
    const long want = (items + 3) / 4;
    const int blocks = (int)(want > 262144 ? 262144 : want);
    if (out_elem == ELEM_F16)
        SGC_LAUNCH(pair_expand_kernel<ELEM_F16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)U,
                           (const u16*)V, sub_idx, obj_idx, (u16*)z_pad, items);
    else
        SGC_LAUNCH(pair_expand_kernel<ELEM_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)U,
                           (const u16*)V, sub_idx, obj_idx, (u16*)z_pad, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// Dense form of the expansion: every ordered pair (i, j) of every image, pair index looked up in pid[n_obj][pid_ld]
// (-1 = skip).  img_ptr [n_img+1] object ranges; max_n = largest object count of an image (<= 150).
// Any of z_pad_f16 / z_pad_bf16 / amz may be NULL.
int sgc_pair_expand_dense_windows(const void* U, const void* V, const int* img_ptr, const int* pid, int pid_ld, int n_img, int max_n,
                                  void* z_pad_f16, void* z_pad_bf16, unsigned char* amz, const int* pixel_rect, void* stream) {
    if (max_n > 150 || max_n < 1) return SGC_ERR_ARG;
    if (n_img <= 0) return SGC_OK;
#ifdef SGC_EXPERIMENTS
    static const int listed = [] { const char* e = getenv("SGC_EXPAND_LIST"); return e ? atoi(e) : 1; }();     // A/B: profiles/r05_expand_ab.txt
#else
    constexpr int listed = 1;
#endif
    if (listed && pixel_rect)
        SGC_LAUNCH(pair_expand_dense_list_kernel, dim3(256, (max_n + EXPAND_JT - 1) / EXPAND_JT, n_img), dim3(512), 0, (hipStream_t)stream,
                   (const u16*)U, (const u16*)V, img_ptr, pid, pid_ld, (u16*)z_pad_f16, (u16*)z_pad_bf16, amz, pixel_rect);
    else
        SGC_LAUNCH(pair_expand_dense_kernel, dim3(256, (max_n + EXPAND_JT - 1) / EXPAND_JT, n_img), dim3(512), 0, (hipStream_t)stream,
                   (const u16*)U, (const u16*)V, img_ptr, pid, pid_ld, (u16*)z_pad_f16, (u16*)z_pad_bf16, amz, pixel_rect);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_pair_expand_dense(const void* U, const void* V, const int* img_ptr, const int* pid, int pid_ld, int n_img, int max_n,
                          void* z_pad_f16, void* z_pad_bf16, unsigned char* amz, void* stream) {
    return sgc_pair_expand_dense_windows(U, V, img_ptr, pid, pid_ld, n_img, max_n, z_pad_f16, z_pad_bf16, amz, nullptr, stream);
}

// y [n_pairs*64][1024] (+argmax u8) = maxpool2(relu(conv3x3(z_pad, w3r[1024][8][9][64]) + b3))
int sgc_conv3_relu_pool(const void* z_pad, const void* w3r, const float* b3, void* y, unsigned char* argmax, void* y_bf16,
                        int n_pairs, void* stream) {
    NtParams p{};
    p.A = (const u16*)z_pad; p.B = (const u16*)w3r; p.C = y; p.M = n_pairs * 256; p.N = 1024; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 1024; p.lgS = 4; p.Cin = 512; p.bias = b3; p.argmax = argmax; p.C2 = (u16*)y_bf16;
    return launch_gemm_nt<ELEM_F16, AMODE_CONV, EPI_POOL>(p, (hipStream_t)stream);
}

// h1 [n_pairs][4096] = dropout(relu(y[n_pairs][65536] * w1p^T + b))
int sgc_fc1_relu(const void* y, const void* w1p, const float* b, void* h1, int n_pairs, int K, int drop_enable,
                 unsigned drop_seed, void* stream) {
    NtParams p{};
    p.A = (const u16*)y; p.B = (const u16*)w1p; p.C = h1; p.M = n_pairs; p.N = 4096; p.K = K;
    p.lda = K; p.ldb = K; p.ldc = 4096; p.bias = b; p.drop_enable = drop_enable; p.drop_seed = drop_seed; p.scale = 2.f;
    return launch_gemm_nt<ELEM_F16, AMODE_PLAIN, EPI_BIAS_RELU>(p, (hipStream_t)stream);
}

// p [n_pairs][512] f32 = dropout(relu(h1 * w2m^T + b + Lsub[sub_idx] + Lobj[obj_idx]))
int sgc_fc2_labels_relu(const void* h1, const void* w2m, const float* b, const float* lsub, const float* lobj,
                        const int* sub_idx, const int* obj_idx, float* p_out, int n_pairs, int drop_enable,
                        unsigned drop_seed, void* stream) {
    NtParams p{};
    p.A = (const u16*)h1; p.B = (const u16*)w2m; p.C = p_out; p.M = n_pairs; p.N = 512; p.K = 4096;
    p.lda = 4096; p.ldb = 4096; p.ldc = 512; p.bias = b; p.lsub = lsub; p.lobj = lobj; p.sub_idx = sub_idx;
    p.obj_idx = obj_idx; p.drop_enable = drop_enable; p.drop_seed = drop_seed; p.scale = 2.f;
#ifdef SGC_EXPERIMENTS
    static const int big = [] { const char* e = getenv("SGC_FC2_BIG"); return e ? atoi(e) : 1; }();       // A/B: profiles/r05_small_kernels.txt
#else
    constexpr int big = 1;
#endif
    if (big && n_pairs >= 16384) {          // N = 512: two column tiles; from 64 row tiles on the 256 x 256 ping-pong block (the size rule of launch_gemm_nt starts at 256^3 outputs)
        p.epi_lds = 0;
        return launch_gemm_nt_pp<ELEM_F16, EPI_FC2>(p, (hipStream_t)stream);
    }
    return launch_gemm_nt<ELEM_F16, AMODE_PLAIN, EPI_FC2>(p, (hipStream_t)stream);
}

int sgc_bayes_head(const float* p, const float* Wt, const float* bias, int n_pairs, int ng, int np, int ns, int hier,
                   float T1, float T2, float T3, float* rel, float* sup, float* conn, float* cand_conf, int* cand_pred,
                   const unsigned char* iou_mask, void* stream) {
    if (ng + np + ns + 4 > 64 || ng <= 0) return SGC_ERR_ARG;
    if (n_pairs <= 0) return SGC_OK;
    HeadParams hp{p, Wt, bias, n_pairs, ng, np, ns, hier, 1.f / T1, 1.f / T2, 1.f / T3, rel, sup, conn, cand_conf,
                  cand_pred, iou_mask};
    const int lds = 512 * 64 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bayes_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    // eight wavefronts per workgroup (one weight image in LDS per CU) when there are pairs for them, four pairs per wavefront pass
    const int waves = n_pairs >= 256 * 8 * HEAD_PAIRS ? 8 : 4;
    int blocks = (n_pairs + waves * HEAD_PAIRS - 1) / (waves * HEAD_PAIRS);
    if (blocks > 256) blocks = 256;
    SGC_LAUNCH(bayes_head_kernel, dim3(blocks), dim3(64 * waves), lds, (hipStream_t)stream, hp);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
