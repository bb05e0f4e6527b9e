// conv3 over SHARED windows: the part of conv3_1 -> ReLU -> 2x2 max-pool (reference model.py:145-147) that is the same for
// every pair with the same subject (or the same object) is computed once per object instead of once per pair.
//
// Why it is exact.  Object o's masked map is tanh(conv1(features)) inside its box and the constant tanh(b1) outside
// (train_test.py:194-195 multiplies by the box mask before conv1_1/conv1_2).  So the conv2 half V_o equals V_bg - the half of
// an object with an EMPTY box - at every 32-grid pixel whose 3x3 neighbourhood misses the box, z_ij = pool(relu(U_i + V_j)) equals
// A_i = pool(relu(U_i + V_bg)) at every 16-grid pixel outside D16_j, and conv3's output for the pair equals conv3(A_i) at every
// pixel whose 3x3 neighbourhood misses D16_j.  Per object that region is the complement of one rectangle R_o of the 8x8 grid of
// conv3's pooling windows (`object_windows`).  For the pair (i, j) a window is
//   I : outside R_j                 -> y_ij[w] = y of the pseudo-pair (i, bg)[w]     (identical inputs, identical arithmetic)
//   J : inside R_j, outside R_i     -> y_ij[w] = y of the pseudo-pair (bg, j)[w]
//   X : inside R_i and R_j          -> computed for the pair (`gemm_nt_kernel<AMODE_CONV_GATHER>` over the list of X windows)
// On the benchmark's boxes 11.5 % of the windows are X (13 % on VG-like box statistics, tools/background_sparsity.py).
#include "gemm_nt.h"

struct WRect { int x0, x1, y0, y1; };                  // half-open on the 8x8 window grid; x1 <= x0: empty

__device__ __forceinline__ void axis_windows(int b0, int b1, int& w0, int& w1) {
    int lo = b0 < 0 ? 0 : b0, hi = b1 > 32 ? 32 : b1;
    if (hi <= lo) { w0 = 0; w1 = 0; return; }
    lo = lo > 0 ? lo - 1 : 0;  hi = hi < 32 ? hi + 1 : 32;        // conv2_1 is 3x3 on the 32-grid
    lo >>= 1;                  hi = (hi + 1) >> 1;                // 2x2 max-pool -> 16-grid
    lo = lo > 0 ? lo - 1 : 0;  hi = hi < 16 ? hi + 1 : 16;        // conv3_1 is 3x3
    w0 = lo >> 1;              w1 = (hi + 1) >> 1;                // 2x2 max-pool -> 8-grid
}

// bbox: x0,x1,y0,y1 in slice semantics (the mask is [y0:y1, x0:x1], csrc/kernels_fwd.hip:mask_objects_kernel)
__device__ __forceinline__ WRect object_windows(const int* __restrict__ b) {
    WRect r;
    axis_windows(b[0], b[1], r.x0, r.x1);
    axis_windows(b[2], b[3], r.y0, r.y1);
    if (r.x1 <= r.x0 || r.y1 <= r.y0) r = WRect{0, 0, 0, 0};
    return r;
}

__device__ __forceinline__ WRect pair_windows(const WRect& a, const WRect& b) {
    WRect r{max(a.x0, b.x0), min(a.x1, b.x1), max(a.y0, b.y0), min(a.y1, b.y1)};
    if (r.x1 <= r.x0 || r.y1 <= r.y0) r = WRect{0, 0, 0, 0};
    return r;
}

__global__ __launch_bounds__(256) void shared_count_kernel(const int* __restrict__ bbox, const int* __restrict__ sub,
                                                           const int* __restrict__ obj, int n_pairs, int* __restrict__ count) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const WRect x = pair_windows(object_windows(bbox + 4 * sub[p]), object_windows(bbox + 4 * obj[p]));
    count[p] = (x.x1 - x.x0) * (x.y1 - x.y0);
}

// gather[e] = pair * 64 + window for the X windows of every pair, pairs in list order, windows row-major
__global__ __launch_bounds__(256) void shared_fill_kernel(const int* __restrict__ bbox, const int* __restrict__ sub,
                                                          const int* __restrict__ obj, int n_pairs, const int* __restrict__ incl,
                                                          int* __restrict__ gather) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const WRect x = pair_windows(object_windows(bbox + 4 * sub[p]), object_windows(bbox + 4 * obj[p]));
    int e = p ? incl[p - 1] : 0;
    for (int wy = x.y0; wy < x.y1; ++wy)
        for (int wx = x.x0; wx < x.x1; ++wx) gather[e++] = p * 64 + wy * 8 + wx;
}

// Rows of the I / J windows: copies of the per-object rows.  One wavefront per (pair, window): 2 KiB of y (+2 KiB bf16 copy,
// +1 KiB routing codes).  y_obj rows: pseudo-pair (i, bg) = i, pseudo-pair (bg, j) = n_obj + j.
__global__ __launch_bounds__(256) void shared_assemble_kernel(const int* __restrict__ bbox, const int* __restrict__ sub,
                                                              const int* __restrict__ obj, long n_rows, int n_obj,
                                                              const uint4* __restrict__ y_obj, const uint4* __restrict__ am_obj,
                                                              const uint4* __restrict__ ybf_obj, uint4* __restrict__ y,
                                                              uint4* __restrict__ am, uint4* __restrict__ ybf) {
    const int lane = threadIdx.x & 63;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (long)gridDim.x * 4) {
        const int p = (int)(row >> 6), w = (int)(row & 63);
        const int wy = w >> 3, wx = w & 7;
        const int i = sub[p], j = obj[p];
        const WRect rj = object_windows(bbox + 4 * j);
        long src;
        if (!(wx >= rj.x0 && wx < rj.x1 && wy >= rj.y0 && wy < rj.y1)) {
            src = (long)i * 64 + w;
        } else {
            const WRect ri = object_windows(bbox + 4 * i);
            if (wx >= ri.x0 && wx < ri.x1 && wy >= ri.y0 && wy < ri.y1) continue;        // X: written by the gathered convolution
            src = ((long)n_obj + j) * 64 + w;
        }
        const uint4 a = y_obj[src * 128 + lane], b = y_obj[src * 128 + 64 + lane];
        y[row * 128 + lane] = a;
        y[row * 128 + 64 + lane] = b;
        if (ybf) {
            const uint4 c = ybf_obj[src * 128 + lane], d = ybf_obj[src * 128 + 64 + lane];
            ybf[row * 128 + lane] = c;
            ybf[row * 128 + 64 + lane] = d;
        }
        if (am) am[row * 64 + lane] = am_obj[src * 64 + lane];
    }
}

extern "C" {

int sgc_shared_windows_count(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, int* count, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(shared_count_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx, n_pairs, count);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_shared_windows_fill(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, const int* count_incl, int* gather,
                            void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(shared_fill_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx, n_pairs,
               count_incl, gather);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// y[gather[e]][1024] (+argmax, +bf16 copy) = maxpool2(relu(conv3x3(z_pad, w3r) + b3)) for the listed windows only.
// max_entries bounds the launch; *gather_n (device) is the real list length.
int sgc_conv3_relu_pool_windows(const void* z_pad, const void* w3r, const float* b3, const int* gather, const int* gather_n,
                                int max_entries, void* y, unsigned char* argmax, void* y_bf16, void* stream) {
    if (max_entries <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)z_pad; p.B = (const u16*)w3r; p.C = y; p.M = max_entries * 4; p.N = 1024; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 1024; p.lgS = 4; p.Cin = 512; p.bias = b3; p.argmax = argmax; p.C2 = (u16*)y_bf16;
    p.gather = gather; p.gather_n = gather_n;
    return launch_gemm_nt_cfg<ELEM_F16, AMODE_CONV_GATHER, EPI_POOL, 2, 4, 4, 2>(p, (hipStream_t)stream);
}

int sgc_shared_windows_assemble(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, int n_obj, const void* y_obj,
                                const unsigned char* argmax_obj, const void* y_obj_bf16, void* y, unsigned char* argmax, void* y_bf16,
                                void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    if ((argmax && !argmax_obj) || (y_bf16 && !y_obj_bf16)) return SGC_ERR_ARG;
    const long rows = (long)n_pairs * 64;
    const long want = (rows + 3) / 4;
    const int blocks = (int)(want > 131072 ? 131072 : want);
    SGC_LAUNCH(shared_assemble_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx, rows, n_obj,
               (const uint4*)y_obj, (const uint4*)argmax_obj, (const uint4*)y_obj_bf16, (uint4*)y, (uint4*)argmax, (uint4*)y_bf16);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
