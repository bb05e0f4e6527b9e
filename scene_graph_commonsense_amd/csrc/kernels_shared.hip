// conv3 and fc1 over SHARED windows: the part of conv3_1 -> ReLU -> 2x2 max-pool -> fc1 (reference model.py:145-149) that is the
// same for every pair with the same subject (or the same object) is computed once per object instead of once per pair, and the part
// that is the same for every object of an image once per image.
//
// Why it is exact.  Object o's masked map is tanh(conv1(features)) inside its box and the constant tanh(b1) outside
// (train_test.py:194-195 multiplies by the box mask before conv1_1/conv1_2).  So the conv2 half V_o equals V_bg - the half of
// an object with an EMPTY box - at every 32-grid pixel whose 3x3 neighbourhood misses the box, z_ij = pool(relu(U_i + V_j)) equals
// A_i = pool(relu(U_i + V_bg)) at every 16-grid pixel outside D16_j, and conv3's output for the pair equals conv3(A_i) at every
// pixel whose 3x3 neighbourhood misses D16_j.  Per object that region is the complement of one rectangle R_o of the 8x8 grid of
// conv3's pooling windows (`object_windows`).  For the pair (i, j) a window is
//   I : outside R_j                 -> y_ij[w] = y of the pseudo-pair (i, bg)[w]     (identical inputs, identical arithmetic)
//   J : inside R_j, outside R_i     -> y_ij[w] = y of the pseudo-pair (bg, j)[w]
//   X : inside R_i and R_j          -> computed for the pair
// and, one level up, a pseudo-pair (o, bg) equals the all-background map (bg, bg) outside R_o.  What is computed per window is the
// WINDOW LIST: the X windows of the real pairs and the windows R_o of the pseudo-pairs (gemm_nt_pp_kernel<ACG> forward; un-pool,
// im2col, column GEMMs, col2im backward); whole conv3 maps remain only for the n_img background maps.  On the benchmark's boxes
// 11.5 % of the windows are X (13 % on VG-like box statistics, tools/background_sparsity.py).
// fc1 is a sum over the 64 windows and runs over the same rows in a window-major row space (see "fc1 over shared windows" below).
// tests: tests/test_shared_identity_cpu.py (the identity on the reference's literal graph, float64), tests/test_shared_conv3_gpu.py
// (bit-identity of the forward, gradients), tests/test_shared_kernels_gpu.py (every entry point against numpy / torch).
#include "gemm_tn.h"

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

__device__ __forceinline__ bool in_rect(const WRect& r, int wx, int wy) { return wx >= r.x0 && wx < r.x1 && wy >= r.y0 && wy < r.y1; }

__device__ __forceinline__ WRect pair_windows(const WRect& a, const WRect& b) {
    WRect r{max(a.x0, b.x0), min(a.x1, b.x1), max(a.y0, b.y0), min(a.y1, b.y1)};
    if (r.x1 <= r.x0 || r.y1 <= r.y0) r = WRect{0, 0, 0, 0};
    return r;
}

// Pixels of the 16-grid within one pixel of a pair's X windows, packed Y0 | Y1 << 5 | X0 << 10 | X1 << 15 (half-open; 0 = none):
// the only pixels where a real pair's z (conv3 input), routing codes and dz exist in the shared-window path.
__device__ __forceinline__ int pack_pixel_rect(const WRect& x) {
    if (x.x1 <= x.x0) return 0;
    const int Y0 = max(2 * x.y0 - 1, 0), Y1 = min(2 * x.y1 + 1, 16), X0 = max(2 * x.x0 - 1, 0), X1 = min(2 * x.x1 + 1, 16);
    return Y0 | (Y1 << 5) | (X0 << 10) | (X1 << 15);
}
__global__ __launch_bounds__(256) void shared_count_kernel(const int* __restrict__ bbox, const int* __restrict__ sub,
                                                           const int* __restrict__ obj, int n_pairs, int* __restrict__ count,
                                                           int* __restrict__ pixrect) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const WRect x = pair_windows(object_windows(bbox + 4 * sub[p]), object_windows(bbox + 4 * obj[p]));
    count[p] = (x.x1 - x.x0) * (x.y1 - x.y0);
    if (pixrect) pixrect[p] = pack_pixel_rect(x);
}

// Second level of sharing: the per-object maps themselves equal the all-background map outside R_o, so a pseudo-pair (o, bg) /
// (bg, o) needs conv3 only on the windows of R_o; they are appended to the window list as entries of the "pair" n_real + ps.
//   count[ps] = |R_o|, pixrect[ps] = the pixels within one pixel of R_o   (o = ps mod n_obj; arrays already offset to the pseudo-pairs)
__global__ __launch_bounds__(256) void shared_count_objects_kernel(const int* __restrict__ bbox, int n_obj, int* __restrict__ count,
                                                                   int* __restrict__ pixrect) {
    const int ps = blockIdx.x * 256 + threadIdx.x;
    if (ps >= 2 * n_obj) return;
    const WRect r = object_windows(bbox + 4 * (ps >= n_obj ? ps - n_obj : ps));
    count[ps] = (r.x1 - r.x0) * (r.y1 - r.y0);
    pixrect[ps] = pack_pixel_rect(r);
}

__global__ __launch_bounds__(256) void shared_fill_objects_kernel(const int* __restrict__ bbox, int n_obj, int n_real,
                                                                  const int* __restrict__ incl, int* __restrict__ gather) {
    const int ps = blockIdx.x * 256 + threadIdx.x;
    if (ps >= 2 * n_obj) return;
    const WRect r = object_windows(bbox + 4 * (ps >= n_obj ? ps - n_obj : ps));
    const int pair = n_real + ps;
    int e = pair ? incl[pair - 1] : 0;
    for (int wy = r.y0; wy < r.y1; ++wy)
        for (int wx = r.x0; wx < r.x1; ++wx) gather[e++] = pair * 64 + wy * 8 + wx;
}

// Rows of the pseudo-pairs outside R_o: copies of the background map of the object's image (y_bg [n_img*64][1024] etc.), written to
// the window-major row goff[w] + ps (y, bf16 copy) and to the pair-major routing row (n_real + ps)*64 + w.
__global__ __launch_bounds__(256) void shared_fill_object_rows_kernel(const int* __restrict__ bbox, const int* __restrict__ obj_img,
                                                                      int n_obj, const int* __restrict__ goff,
                                                                      const uint4* __restrict__ y_bg, const uint4* __restrict__ ybf_bg,
                                                                      const uint4* __restrict__ am_bg, uint4* __restrict__ ywm,
                                                                      uint4* __restrict__ ywm_bf, uint4* __restrict__ am_ps, long n_rows) {
    const int lane = threadIdx.x & 63;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (long)gridDim.x * 4) {
        const int ps = (int)(row >> 6), w = (int)(row & 63);
        const int o = ps >= n_obj ? ps - n_obj : ps;
        if (in_rect(object_windows(bbox + 4 * o), w & 7, w >> 3)) continue;         // computed for the object (window-list entry)
        const long src = (long)obj_img[o] * 64 + w, dst = (long)goff[w] + ps;
        ywm[dst * 128 + lane] = y_bg[src * 128 + lane];
        ywm[dst * 128 + 64 + lane] = y_bg[src * 128 + 64 + lane];
        if (ywm_bf) {
            ywm_bf[dst * 128 + lane] = ybf_bg[src * 128 + lane];
            ywm_bf[dst * 128 + 64 + lane] = ybf_bg[src * 128 + 64 + lane];
        }
        if (am_ps) am_ps[row * 64 + lane] = am_bg[src * 64 + lane];
    }
}

// Transpose of that copy: dy_bg[b*64 + w] = sum over the pseudo-pairs ps of image b with w outside R_o of dywm[goff[w] + ps]
__global__ __launch_bounds__(256) void shared_bg_grad_kernel(const int* __restrict__ bbox, const int* __restrict__ img_ptr, int n_obj,
                                                             const int* __restrict__ goff, const u16* __restrict__ dywm,
                                                             u16* __restrict__ dy_bg, long n_items) {
    // one wavefront per (image, window, half of the 1024 channels).  Round 5: the rows that contribute are found by a ballot (lane = object)
    // and requested eight at a time before they are added - in object order, role 0 then role 1, as before: same sums.  The first form
    // walked the 2 x n objects one dependent 2 KiB load after the other on 512 wavefronts: 0.23 ms of latency for 130 MB.
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        const int half = (int)(it & 1);
        const long bw = it >> 1;
        const int b = (int)(bw >> 6), w = (int)(bw & 63);
        const int o0 = img_ptr[b], o1 = img_ptr[b + 1];
        const u16* rows = dywm + (long)goff[w] * 1024 + half * 512 + lane * 8;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int role = 0; role < 2; ++role)
            for (int base = o0; base < o1; base += 64) {
                const int o = base + lane;
                const bool ok = o < o1 && !in_rect(object_windows(bbox + 4 * o), w & 7, w >> 3);
                unsigned long long m = __ballot(ok);
                while (m) {
                    uint4 g[8];
                    int nv = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (m) {
                            const int bit = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            g[u] = *reinterpret_cast<const uint4*>(rows + (long)(role * n_obj + base + bit) * 1024);
                            nv = u + 1;
                        }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < nv) {
                            const u16* gh = reinterpret_cast<const u16*>(&g[u]);
#pragma unroll
                            for (int k = 0; k < 8; ++k) acc[k] += bf16_bits_to_f32(gh[k]);
                        }
                }
            }
        uint4 oa;
        u16* oah = reinterpret_cast<u16*>(&oa);
#pragma unroll
        for (int k = 0; k < 8; ++k) oah[k] = f32_to_bf16_bits(acc[k]);
        *reinterpret_cast<uint4*>(dy_bg + bw * 1024 + half * 512 + lane * 8) = oa;
    }
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


// ------------------------------------------------------------------------------------------------ backward
// Transpose of the assembly: the gradient of a per-object row is the sum of the gradients of its copies.
//   dy_obj[o*64 + w]           = sum over pairs p with subject o whose window w is I (outside the object's rectangle)   of dy[p*64 + w]
//   dy_obj[(n_obj + o)*64 + w] = sum over pairs p with object  o whose window w is J (inside R_o, outside the subject's) of dy[p*64 + w]
// One wavefront per (role, object, window): f32 sums in pair-list order (no atomics), four 2 KiB rows in flight.
__global__ __launch_bounds__(256) void shared_assemble_bwd_kernel(const int* __restrict__ bbox, const int* __restrict__ sub,
                                                                  const int* __restrict__ obj, const int* __restrict__ sub_ptr,
                                                                  const int* __restrict__ sub_list, const int* __restrict__ obj_ptr,
                                                                  const int* __restrict__ obj_list, int n_obj,
                                                                  const u16* __restrict__ dy, u16* __restrict__ dy_obj, long n_items) {
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        const int role = it >= (long)n_obj * 64 ? 1 : 0;
        const int rem = (int)(it - (long)role * n_obj * 64);
        const int o = rem >> 6, w = rem & 63, wy = w >> 3, wx = w & 7;
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        long vp[4];
        int nv = 0;
        auto flush = [&]() __attribute__((always_inline)) {
            uint4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < nv) {
                    const u16* r = dy + (vp[u] * 64 + w) * 1024 + lane * 8;
                    a[u] = *reinterpret_cast<const uint4*>(r);
                    b[u] = *reinterpret_cast<const uint4*>(r + 512);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < nv) {
                    const u16* ah = reinterpret_cast<const u16*>(&a[u]);
                    const u16* bh = reinterpret_cast<const u16*>(&b[u]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { acc[k] += bf16_bits_to_f32(ah[k]); acc[8 + k] += bf16_bits_to_f32(bh[k]); }
                }
            nv = 0;
        };
        const WRect ro = object_windows(bbox + 4 * o);
        if (role == 0) {
            for (int i = sub_ptr[o]; i < sub_ptr[o + 1]; ++i) {
                const int p = sub_list[i];
                if (!in_rect(object_windows(bbox + 4 * obj[p]), wx, wy)) { vp[nv++] = p; if (nv == 4) flush(); }
            }
        } else if (in_rect(ro, wx, wy)) {
            for (int i = obj_ptr[o]; i < obj_ptr[o + 1]; ++i) {
                const int p = obj_list[i];
                if (!in_rect(object_windows(bbox + 4 * sub[p]), wx, wy)) { vp[nv++] = p; if (nv == 4) flush(); }
            }
        }
        if (nv) flush();
        uint4 oa, ob;
        u16* oah = reinterpret_cast<u16*>(&oa);
        u16* obh = reinterpret_cast<u16*>(&ob);
#pragma unroll
        for (int k = 0; k < 8; ++k) { oah[k] = f32_to_bf16_bits(acc[k]); obh[k] = f32_to_bf16_bits(acc[8 + k]); }
        u16* dst = dy_obj + it * 1024 + lane * 8;
        *reinterpret_cast<uint4*>(dst) = oa;
        *reinterpret_cast<uint4*>(dst + 512) = ob;
    }
}

// Un-pool of the listed windows: dy3x[4e + q][c] = (argmax[gather[e]][c] == q) ? dy[gather[e]][c] : 0  (ReLU + max-pool backward,
// model.py:146-147), rows e >= *gather_n up to entries_pad are zero (padding of the K dimension of the weight-gradient GEMM);
// bias_part[block][c] = the block's share of sum_e dy[..][c] over live routes (conv3 bias gradient).
__global__ __launch_bounds__(256) void windows_unpool_kernel(const u16* __restrict__ dy, const unsigned char* __restrict__ am,
                                                             const int* __restrict__ gather, const int* __restrict__ gather_n,
                                                             const int* __restrict__ dest, int entries_pad, u16* __restrict__ dy3x,
                                                             float* __restrict__ bias_part, int entry0) {
    __shared__ float red[4][1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int E = *gather_n - entry0;             // gather / dest / dy3x point at entry ``entry0`` of the list
    float bs[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bs[k] = 0.f;
    for (int e = blockIdx.x * 4 + wv; e < entries_pad; e += gridDim.x * 4) {
        uint4 o[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[q][0] = make_uint4(0, 0, 0, 0); o[q][1] = make_uint4(0, 0, 0, 0); }
        if (e < E) {
            const long row = gather[e];
            const long drow = dest ? (long)dest[e] : row;       // window-major gradient rows (shared fc1) or pair-major
            uint4 v[2];
            v[0] = *reinterpret_cast<const uint4*>(dy + drow * 1024 + lane * 16);
            v[1] = *reinterpret_cast<const uint4*>(dy + drow * 1024 + lane * 16 + 8);
            const uint4 cd = *reinterpret_cast<const uint4*>(am + row * 1024 + lane * 16);
            const u16* vh = reinterpret_cast<const u16*>(v);
            const unsigned char* ch = reinterpret_cast<const unsigned char*>(&cd);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned code = ch[k];
                if (code < 4u) bs[k] += bf16_bits_to_f32(vh[k]);
#pragma unroll
                for (int q = 0; q < 4; ++q) reinterpret_cast<u16*>(o[q])[k] = code == (unsigned)q ? vh[k] : (u16)0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u16* dst = dy3x + ((long)e * 4 + q) * 1024 + lane * 16;
            *reinterpret_cast<uint4*>(dst) = o[q][0];
            *reinterpret_cast<uint4*>(dst + 8) = o[q][1];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) red[wv][lane * 16 + k] = bs[k];
    __syncthreads();
    for (int c = threadIdx.x; c < 1024; c += 256)
        bias_part[(long)blockIdx.x * 1024 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// zcol[4e + q][tap][512] = z_pad_bf16[pair][y + ky][x + kx][:]: the rows of the weight-gradient GEMM's second operand.
// One wavefront per row: its nine 1 KiB neighbourhood rows are requested before the first is stored.
__global__ __launch_bounds__(256) void windows_im2col_kernel(const u16* __restrict__ zbf, const int* __restrict__ gather,
                                                             const int* __restrict__ gather_n, long n_rows, u16* __restrict__ zcol) {
    const int lane = threadIdx.x & 63;
    const int E = *gather_n;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (long)gridDim.x * 4) {
        const int e = (int)(row >> 2), q = (int)(row & 3);
        uint4 v[9];
        if (e < E) {
            const int g = gather[e];
            const int pair = g >> 6, w = g & 63;
            const int y = 2 * (w >> 3) + (q >> 1), x = 2 * (w & 7) + (q & 1);
            const u16* src = zbf + (((long)pair * 18 + y) * 18 + x) * 512 + lane * 8;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v[tap] = *reinterpret_cast<const uint4*>(src + ((tap / 3) * 18 + (tap % 3)) * 512);
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v[tap] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<uint4*>(zcol + (row * 9 + tap) * 512 + lane * 8) = v[tap];
    }
}

// zpatch[e][4 py + px][512] = z_pad_bf16[pair][2 wy + py][2 wx + px][:]: the 4 x 4 input patch of every listed window (padded
// coordinates), the second operand of the weight-gradient product in its PATCH form (gemm_tn.h: BMODE_PATCH) - 16 rows per window
// instead of the 36 of the im2col form.  One wavefront per (entry, patch row).
// SRC_F16: the source map is the forward's f16 z (no bf16 copy of the real pairs is kept); values are converted on the way.
__device__ __forceinline__ uint4 f16x8_to_bf16x8(uint4 v) {
    const u16* h = reinterpret_cast<const u16*>(&v);
    uint4 o;
    unsigned* od = reinterpret_cast<unsigned*>(&o);
#pragma unroll
    for (int k = 0; k < 4; ++k) od[k] = f32x2_to_bf16x2_bits(f16_bits_to_f32(h[2 * k]), f16_bits_to_f32(h[2 * k + 1]));
    return o;
}
template <bool SRC_F16>
__global__ __launch_bounds__(256) void windows_im2patch_kernel(const u16* __restrict__ zbf, const int* __restrict__ gather,
                                                               const int* __restrict__ gather_n, long n_rows, u16* __restrict__ zpatch, int e0) {
    const int lane = threadIdx.x & 63;
    const int E = *gather_n;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (long)gridDim.x * 4) {
        const int e = e0 + (int)(row >> 2), py = (int)(row & 3);             // output rows count from entry e0
        uint4 v[4];
        if (e < E) {
            const int g = gather[e];
            const int pair = g >> 6, w = g & 63;
            const u16* src = zbf + (((long)pair * 18 + 2 * (w >> 3) + py) * 18 + 2 * (w & 7)) * 512 + lane * 8;
#pragma unroll
            for (int px = 0; px < 4; ++px) v[px] = *reinterpret_cast<const uint4*>(src + px * 512);
        } else {
#pragma unroll
            for (int px = 0; px < 4; ++px) v[px] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) *reinterpret_cast<uint4*>(zpatch + (row * 4 + px) * 512 + lane * 8) = SRC_F16 ? f16x8_to_bf16x8(v[px]) : v[px];
    }
}

// dz[pair][pixel] = sum over the taps of col[row of the source pixel][tap]: the scatter half of the transposed convolution, written
// as a gather so that every dz row has one writer.  Only the pixels within one pixel of the pair's X windows exist; one workgroup
// per pair.
__device__ __forceinline__ void col2im_rect(const u16* __restrict__ col, const WRect& x, int off, long p, u16* __restrict__ dz) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (x.x1 <= x.x0) return;
    const int wdt = x.x1 - x.x0;
    const int Y0 = max(2 * x.y0 - 1, 0), Y1 = min(2 * x.y1 + 1, 16), X0 = max(2 * x.x0 - 1, 0), X1 = min(2 * x.x1 + 1, 16);
    const int nx = X1 - X0, n = (Y1 - Y0) * nx;
    for (int t = wv; t < n; t += 4) {
        const int y = Y0 + t / nx, xx = X0 + t % nx;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int qy = y - ky + 1, qx = xx - kx + 1;
            if (qy >= 2 * x.y0 && qy < 2 * x.y1 && qx >= 2 * x.x0 && qx < 2 * x.x1) {
                const long row = 4L * (off + ((qy >> 1) - x.y0) * wdt + ((qx >> 1) - x.x0)) + (qy & 1) * 2 + (qx & 1);
                const uint4 v = *reinterpret_cast<const uint4*>(col + (row * 9 + tap) * 512 + lane * 8);
                const u16* vh = reinterpret_cast<const u16*>(&v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf16_bits_to_f32(vh[k]);
            }
        }
        uint4 ov;
        u16* oh = reinterpret_cast<u16*>(&ov);
#pragma unroll
        for (int k = 0; k < 8; ++k) oh[k] = f32_to_bf16_bits(acc[k]);
        const int m3 = 4 * ((y >> 1) * 8 + (xx >> 1)) + (y & 1) * 2 + (xx & 1);
        *reinterpret_cast<uint4*>(dz + (p * 256 + m3) * 512 + lane * 8) = ov;
    }
}

__global__ __launch_bounds__(256) void windows_col2im_kernel(const u16* __restrict__ col, const int* __restrict__ bbox,
                                                             const int* __restrict__ sub, const int* __restrict__ obj,
                                                             const int* __restrict__ incl, u16* __restrict__ dz) {
    const int p = blockIdx.x;
    const int e0 = p ? incl[p - 1] : 0;
    if (incl[p] == e0) return;                 // no entries of its own in this list (no X window, or a linear pair: no dz either)
    const WRect x = pair_windows(object_windows(bbox + 4 * sub[p]), object_windows(bbox + 4 * obj[p]));
    col2im_rect(col, x, e0, p, dz);
}

// PATCH form of the same sum (sgc_windows_dgrad_patches): the PATCH_SLOTS = 20 rows of window e hold the 9-tap sums for the 16 pixels
// (2 wy - 1 + py, 2 wx - 1 + px) of its input patch over its own pixels - pp = 4 py + px in natural order, the four centre pixels in
// two consecutive rows (two of their four (own pixel, tap) combinations each: the product keeps K <= 2048, see gemm_nt_pp_kernel<SEG>);
// dz of a pixel adds the rows of the (at most 2 x 2) windows of the pair's rectangle whose patch covers it.
constexpr int PATCH_SLOTS = 20;
// Two row layouts in one buffer: the entries e < n16 (the real pairs' windows whose data gradient ran on the sparse matrix cores,
// csrc/kernels_dgrad_sp.hip) have 16 rows - one per patch pixel, the centre pixels' two halves already summed in the accumulators -
// at patch + e * 16 * 512; the entries behind them have the dense form's PATCH_SLOTS rows at patch + n16 * 16 * 512 + (e - n16) *
// PATCH_SLOTS * 512.  n16 = 0: every entry in the dense form.
__device__ __forceinline__ void patch_sum_rect(const u16* __restrict__ patch, const WRect& x, int off, long p, u16* __restrict__ dz, int n16) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (x.x1 <= x.x0) return;
    const int wdt = x.x1 - x.x0;
    const int Y0 = max(2 * x.y0 - 1, 0), Y1 = min(2 * x.y1 + 1, 16), X0 = max(2 * x.x0 - 1, 0), X1 = min(2 * x.x1 + 1, 16);
    const int nx = X1 - X0, n = (Y1 - Y0) * nx;
    for (int t = wv; t < n; t += 4) {
        const int y = Y0 + t / nx, xx = X0 + t % nx;
        // the rows that cover this pixel, in the order they are added (window row-major; a dense-form centre pixel has two) - all of them
        // requested before the first is added (round 5: they used to be loaded and added one after the other behind uniform branches)
        long rows[8];
        int nr = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int wy = ((y + 1) >> 1) - a;                 // windows whose patch rows 2 wy - 1 .. 2 wy + 2 contain y
            if (wy < x.y0 || wy >= x.y1) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int wx = ((xx + 1) >> 1) - b;
                if (wx < x.x0 || wx >= x.x1) continue;
                const int py = y - 2 * wy + 1, px = xx - 2 * wx + 1, pp = py * 4 + px;
                bool centre = (py == 1 || py == 2) && (px == 1 || px == 2);          // dense form: two slots (two of the four combinations each)
                const int ew = off + (wy - x.y0) * wdt + (wx - x.x0);
                long row;
                if (ew < n16) { row = 16L * ew + pp; centre = false; }
                else row = 16L * n16 + (long)PATCH_SLOTS * (ew - n16) + pp + (pp > 5) + (pp > 6) + (pp > 9) + (pp > 10);
                rows[nr++] = row;
                if (centre) rows[nr++] = row + 1;
            }
        }
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nr) v[u] = *reinterpret_cast<const uint4*>(patch + rows[u] * 512 + lane * 8);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nr) {
                const u16* vh = reinterpret_cast<const u16*>(&v[u]);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf16_bits_to_f32(vh[k]);
            }
        uint4 ov;
        u16* oh = reinterpret_cast<u16*>(&ov);
#pragma unroll
        for (int k = 0; k < 8; ++k) oh[k] = f32_to_bf16_bits(acc[k]);
        const int m3 = 4 * ((y >> 1) * 8 + (xx >> 1)) + (y & 1) * 2 + (xx & 1);
        *reinterpret_cast<uint4*>(dz + (p * 256 + m3) * 512 + lane * 8) = ov;
    }
}

__global__ __launch_bounds__(256) void windows_patch_sum_kernel(const u16* __restrict__ patch, const int* __restrict__ bbox,
                                                                const int* __restrict__ sub, const int* __restrict__ obj,
                                                                const int* __restrict__ incl, u16* __restrict__ dz, int n16) {
    const int p = blockIdx.x;
    const int e0 = p ? incl[p - 1] : 0;
    if (incl[p] == e0) return;                 // no entries of its own in this list (no X window, or a linear pair: no dz either)
    const WRect x = pair_windows(object_windows(bbox + 4 * sub[p]), object_windows(bbox + 4 * obj[p]));
    patch_sum_rect(patch, x, e0, p, dz, n16);
}

__global__ __launch_bounds__(256) void windows_patch_sum_objects_kernel(const u16* __restrict__ patch, const int* __restrict__ bbox, int n_obj,
                                                                        int n_real, const int* __restrict__ incl, u16* __restrict__ dz, int n16) {
    const int ps = blockIdx.x, pair = n_real + ps;
    const WRect x = object_windows(bbox + 4 * (ps >= n_obj ? ps - n_obj : ps));
    patch_sum_rect(patch, x, pair ? incl[pair - 1] : 0, pair, dz, n16);
}

// the same for the window-list entries of the pseudo-pairs (pair index n_real + ps, windows R_o)
__global__ __launch_bounds__(256) void windows_col2im_objects_kernel(const u16* __restrict__ col, const int* __restrict__ bbox, int n_obj,
                                                                     int n_real, const int* __restrict__ incl, u16* __restrict__ dz) {
    const int ps = blockIdx.x, pair = n_real + ps;
    const WRect x = object_windows(bbox + 4 * (ps >= n_obj ? ps - n_obj : ps));
    col2im_rect(col, x, pair ? incl[pair - 1] : 0, pair, dz);
}

// Pair contraction (csrc/kernels_bwd.hip:pair_contract_kernel) for the shared-window backward: a real pair contributes to pixel
// (Y, X) only if the pixel lies within one pixel of its X windows (elsewhere dz does not exist); every object also has its
// pseudo-pair (o, bg) / (bg, o) at pair index n_real + o / n_real + n_obj + o, and the background object of image b (index
// n_obj + b) collects the other side of the pseudo-pairs of that image's objects.
// PIPE (round 5): the candidates that pass the pixel-rectangle test are compacted into a per-wavefront list first, and the rows of the
// NEXT group of four are requested before the current group is routed into the accumulators (the plain form asks for four rows, waits,
// routes them, and only then learns the next four pair indices from the ballot); same candidates in the same order: same sums.
template <bool PIPE>
__global__ __launch_bounds__(256) void pair_contract_windows_kernel(const u16* __restrict__ dz, const unsigned char* __restrict__ amz,
                                                                    const int* __restrict__ ptr, const int* __restrict__ list,
                                                                    const int* __restrict__ pixrect, const int* __restrict__ img_ptr,
                                                                    int role, int n_real, int n_obj, int bg_maps, u16* __restrict__ dU,
                                                                    long n_items) {
    __shared__ int cand_s[4][64];
    const int lane = threadIdx.x & 63;
    int* const cand = cand_s[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        const int o = (int)(it >> 8), W = (int)(it & 255);
        const int Y = W >> 4, X = W & 15;
        const int m3 = 4 * ((Y >> 1) * 8 + (X >> 1)) + (Y & 1) * 2 + (X & 1);
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[q][k] = 0.f;
        auto add4 = [&](const long (&vp)[4], int nv) __attribute__((always_inline)) {
            uint4 g[4];
            unsigned a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < nv) {
                    g[u] = *reinterpret_cast<const uint4*>(dz + (vp[u] * 256 + m3) * 512 + lane * 8);
                    a[u] = *reinterpret_cast<const unsigned*>(amz + (vp[u] * 256 + W) * 256 + lane * 4);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < nv) {
                    const u16* gh = reinterpret_cast<const u16*>(&g[u]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float v = bf16_bits_to_f32(gh[k]);
                        const unsigned code = (a[u] >> (4 * k)) & 15u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q][k] += (code == (unsigned)q) ? v : 0.f;
                    }
                }
        };
        auto request = [&](int u0, int cnt, uint4 (&g)[4], unsigned (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u0 + u < cnt) {
                    const long v = (long)__builtin_amdgcn_readfirstlane(cand[u0 + u]);
                    g[u] = *reinterpret_cast<const uint4*>(dz + (v * 256 + m3) * 512 + lane * 8);
                    a[u] = *reinterpret_cast<const unsigned*>(amz + (v * 256 + W) * 256 + lane * 4);
                }
        };
        auto route = [&](int u0, int cnt, const uint4 (&g)[4], const unsigned (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u0 + u < cnt) {
                    const u16* gh = reinterpret_cast<const u16*>(&g[u]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float v = bf16_bits_to_f32(gh[k]);
                        const unsigned code = (a[u] >> (4 * k)) & 15u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q][k] += (code == (unsigned)q) ? v : 0.f;
                    }
                }
        };
        // candidates of this (object, pixel), 64 at a time: lane k looks at the k-th one, a ballot keeps those whose rows exist
        auto scan = [&](int i0, int i1, bool real) __attribute__((always_inline)) {
            for (int base = i0; base < i1; base += 64) {
                const int i = base + lane;
                int pk = 0;
                bool ok = false;
                if (i < i1) {
                    pk = real ? list[i] : n_real + (role ? i : n_obj + i);
                    ok = in_pixel_rect(pixrect[pk], Y, X);                  // pixrect covers real pairs, pseudo-pairs and background maps
                }
                unsigned long long m = __ballot(ok);
                if constexpr (PIPE) {
                    const int cnt = __popcll(m);
                    if (cnt == 0) continue;
                    __builtin_amdgcn_wave_barrier();                        // the previous chunk's list has been read
                    if (ok) cand[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = pk;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    uint4 g0[4], g1[4];
                    unsigned a0[4], a1[4];
                    request(0, cnt, g0, a0);
                    for (int u0 = 0; u0 < cnt; u0 += 8) {
                        if (u0 + 4 < cnt) request(u0 + 4, cnt, g1, a1);
                        route(u0, cnt, g0, a0);
                        if (u0 + 4 < cnt) {
                            if (u0 + 8 < cnt) request(u0 + 8, cnt, g0, a0);
                            route(u0 + 4, cnt, g1, a1);
                        }
                    }
                    continue;
                }
                while (m) {
                    long vp[4];
                    int nv = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (m) {
                            const int b = __ffsll((long long)m) - 1;
                            m &= m - 1;
                            vp[u] = (long)__shfl(pk, b);
                            nv = u + 1;
                        }
                    add4(vp, nv);
                }
            }
        };
        if (o < n_obj) {
            // every real pair of o writes its dz inside the pixel rectangle of o's pseudo-pair (X = R_o n R_j lies in R_o): outside it
            // nothing contributes and the partner list is not walked (60 % of an object's cells on the benchmark's boxes)
            long vp[4];
            vp[0] = (long)n_real + (role ? n_obj + o : o);
            if (in_pixel_rect(pixrect[vp[0]], Y, X)) {
                scan(ptr[o], ptr[o + 1], true);
                add4(vp, 1);
            }
        } else {
            scan(img_ptr[o - n_obj], img_ptr[o - n_obj + 1], false);
            if (bg_maps) {                                                  // the all-background map of this image: pair (bg_b, bg_b)
                long vp[4];
                vp[0] = (long)n_real + 2 * n_obj + (o - n_obj);
                add4(vp, 1);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 ov;
            u16* oh = reinterpret_cast<u16*>(&ov);
#pragma unroll
            for (int k = 0; k < 8; ++k) oh[k] = f32_to_bf16_bits(acc[q][k]);
            const int yy = 2 * Y + (q >> 1) + 1, xx = 2 * X + (q & 1) + 1;
            *reinterpret_cast<uint4*>(dU + (((long)o * 34 + yy) * 34 + xx) * 512 + lane * 8) = ov;
        }
    }
}

// ================================================================================================ fc1 over shared windows
// fc1 is a sum over conv3's 64 pooling windows, h_pre[p] = b + sum_w W1[w] * y_p[w], and y_p[w] is a copy of a per-object row on the
// I / J windows.  All rows that fc1 multiplies live in ONE "window-major" row space: group w = [the 2*n_obj pseudo-pair rows of
// window w][the X entries of window w, pair order][zero rows up to a multiple of 256], so that fc1 is one grouped GEMM
// (row tile t multiplies with the weight slice of its group) instead of a [P, 65536] GEMM:
//   O[r] = Y[r] * W1[w(r)]^T                      (f32; pseudo rows: T_o[w], X rows: the pair's partial product for that window)
//   S_o  = 2-D inclusive prefix sums of T_o over the 8x8 window grid (9x9 with a zero border)
//   h_pre[p] = b + S_i[all] - S_i[R_j] + S'_j[R_j] - S'_j[X_p] + sum_{e in X_p} O[dest[e]]          (rectangle sums = 4 look-ups)
// which is the same sum in a different order (f32 round-off instead of bit equality with the [P, 65536] GEMM).
// Row pitch of the f32 products owm in floats: 4096 + 256 B.  With the power-of-two pitch every row of a 256 x 256 output tile starts
// 16 KiB after the previous one and the tile's stores drain 10 % slower (tools/fc1_windows_microbench.py: 3.18 -> 2.86 ms per launch
// with 16-byte stores; profiles/r03_fc1_windows_microbench*.txt).
static inline int owm_pitch() { return sgc_tuning().owm_pitch; }     // 4160
__global__ __launch_bounds__(256) void fc1_integral_kernel(const float* __restrict__ owm, int pitch, const int* __restrict__ goff, int n2,
                                                           float* __restrict__ S) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(idx & 4095);
    const int ps = (int)(idx >> 12);
    if (ps >= n2) return;
    float* So = S + (long)ps * 81 * 4096 + c;
    float col[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) col[x] = 0.f;
#pragma unroll
    for (int x = 0; x < 9; ++x) So[(long)x * 4096] = 0.f;
    for (int y = 0; y < 8; ++y) {
        float run = 0.f;
        So[(long)((y + 1) * 9) * 4096] = 0.f;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            run += owm[((long)goff[y * 8 + x] + ps) * pitch + c];
            col[x] += run;
            So[(long)((y + 1) * 9 + x + 1) * 4096] = col[x];
        }
    }
}

// T[j] = S'_j[R_j]: the sum of object j's own rectangle of its (background, j) prefix sums - the one rectangle term of the assembly that
// depends on ONE object only (same expression as rect_acc, so the assembled h1 keeps its bits): read once per pair instead of four
// corner vectors per pair (13 -> 10 vectors of 16 KB per pair).
__global__ __launch_bounds__(256) void fc1_own_rect_kernel(const float* __restrict__ S, const int* __restrict__ bbox, int n_obj, float* __restrict__ T) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(idx & 4095), j = (int)(idx >> 12);
    if (j >= n_obj) return;
    const WRect r = object_windows(bbox + 4 * j);
    const float* So = S + ((long)n_obj + j) * 81 * 4096 + c;
    float v = 0.f;
    if (r.x1 > r.x0)
        v = (So[(long)(r.y1 * 9 + r.x1) * 4096] - So[(long)(r.y0 * 9 + r.x1) * 4096]) -
            (So[(long)(r.y1 * 9 + r.x0) * 4096] - So[(long)(r.y0 * 9 + r.x0) * 4096]);
    T[(long)j * 4096 + c] = v;
}

__device__ __forceinline__ void rect_acc(float (&acc)[16], const float* __restrict__ So, const WRect& r, float sign, int c0) {
    const float* a = So + (long)(r.y1 * 9 + r.x1) * 4096 + c0;
    const float* b = So + (long)(r.y0 * 9 + r.x1) * 4096 + c0;
    const float* c = So + (long)(r.y1 * 9 + r.x0) * 4096 + c0;
    const float* d = So + (long)(r.y0 * 9 + r.x0) * 4096 + c0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float4 fa = reinterpret_cast<const float4*>(a)[v], fb = reinterpret_cast<const float4*>(b)[v];
        const float4 fc = reinterpret_cast<const float4*>(c)[v], fd = reinterpret_cast<const float4*>(d)[v];
        acc[4 * v + 0] += sign * ((fa.x - fb.x) - (fc.x - fd.x));
        acc[4 * v + 1] += sign * ((fa.y - fb.y) - (fc.y - fd.y));
        acc[4 * v + 2] += sign * ((fa.z - fb.z) - (fc.z - fd.z));
        acc[4 * v + 3] += sign * ((fa.w - fb.w) - (fc.w - fd.w));
    }
}

// h1[p] = dropout(relu(h_pre[p])) as above; one workgroup per pair, thread t owns channels 16t .. 16t+15
// (reference model.py:148-149: fc1 -> ReLU -> dropout; the keep bit is the one sgc_fc1_relu uses: hash(seed, p*4096 + c))
__global__ __launch_bounds__(256) void fc1_assemble_kernel(const float* __restrict__ S, const float* __restrict__ owm, int pitch,
                                                           const int* __restrict__ bbox, const int* __restrict__ sub,
                                                           const int* __restrict__ obj, const int* __restrict__ incl,
                                                           const int* __restrict__ dest, int n_obj, const float* __restrict__ bias,
                                                           int drop_enable, unsigned seed, float scale, u16* __restrict__ h1,
                                                           const float* __restrict__ own, const int* __restrict__ order,
                                                           const u16* __restrict__ oxh) {
    // ``order`` (optional): workgroup b assembles pair order[b].  With the pairs sorted by SUBJECT (the contraction's CSR list) the ~63
    // consecutive workgroups of a subject read their five subject-side vectors from the same 1.3 MB prefix table S_i, which then stays
    // in the L2s; in the reference's pair order (graph_iter, direction, image) neighbours share nothing
    const int p = order ? order[blockIdx.x] : blockIdx.x, c0 = threadIdx.x * 16;
    const int i = sub[p], j = obj[p];
    const WRect ri = object_windows(bbox + 4 * i), rj = object_windows(bbox + 4 * j);
    const WRect x = pair_windows(ri, rj);
    const float* Si = S + (long)i * 81 * 4096;
    const float* Sj = S + ((long)n_obj + j) * 81 * 4096;
    float acc[16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float4 b = reinterpret_cast<const float4*>(bias + c0)[v];
        const float4 t = reinterpret_cast<const float4*>(Si + (long)80 * 4096 + c0)[v];
        acc[4 * v + 0] = b.x + t.x; acc[4 * v + 1] = b.y + t.y; acc[4 * v + 2] = b.z + t.z; acc[4 * v + 3] = b.w + t.w;
    }
    if (rj.x1 > rj.x0) {
        rect_acc(acc, Si, rj, -1.f, c0);
        if (own) {                                   // S'_j[R_j], pre-summed per object (fc1_own_rect_kernel): the same value
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t = reinterpret_cast<const float4*>(own + (long)j * 4096 + c0)[v];
                acc[4 * v + 0] += 1.f * t.x; acc[4 * v + 1] += 1.f * t.y; acc[4 * v + 2] += 1.f * t.z; acc[4 * v + 3] += 1.f * t.w;
            }
        } else {
            rect_acc(acc, Sj, rj, 1.f, c0);
        }
        if (x.x1 > x.x0) rect_acc(acc, Sj, x, -1.f, c0);
    }
    const int e0 = p ? incl[p - 1] : 0, e1 = incl[p];
    if (oxh) {                                       // the pair's X products as f16 rows [rows][4096] (sgc_fc1_windows_gemm_x16)
        for (int e = e0; e < e1; ++e) {
            const uint4* o = reinterpret_cast<const uint4*>(oxh + (long)dest[e] * 4096 + c0);
            const uint4 t0 = o[0], t1 = o[1];
            const u16* h0 = reinterpret_cast<const u16*>(&t0);
            const u16* h1v = reinterpret_cast<const u16*>(&t1);
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc[k] += f16_bits_to_f32(h0[k]); acc[8 + k] += f16_bits_to_f32(h1v[k]); }
        }
    } else {
        for (int e = e0; e < e1; ++e) {
            const float* o = owm + (long)dest[e] * pitch + c0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t = reinterpret_cast<const float4*>(o)[v];
                acc[4 * v + 0] += t.x; acc[4 * v + 1] += t.y; acc[4 * v + 2] += t.z; acc[4 * v + 3] += t.w;
            }
        }
    }
    uint4 out[2];
    u16* oh = reinterpret_cast<u16*>(out);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float v = fmaxf(acc[k], 0.f);
        if (drop_enable) v = dropout_keep(seed, (uint32_t)(p * 4096 + c0 + k)) ? v * scale : 0.f;
        oh[k] = f32_to_f16_bits(v);
    }
    uint4* dst = reinterpret_cast<uint4*>(h1 + (long)p * 4096 + c0);
    dst[0] = out[0];
    dst[1] = out[1];
}

// ---- fc1 backward over the window-major rows
// Gradient of the per-object rows (transpose of the assembly): G[(role, o)][w] = sum of dh1 over the pairs of o whose window w was
// a copy of o's row - role 0: all windows outside the partner's rectangle; role 1: windows inside o's and outside the partner's.
// Per object that is a small matrix product  G[64 windows][4096] = M^T[64][pairs] x DH[pairs][4096]  with a 0/1 mask matrix M
// (about 63 pairs at N = 64): 34 GFLOP over the minibatch - nothing for the matrix cores, but 1.7 ms as 64 masked VALU additions per
// pair and channel (round 2), and 2.0 ms as a rectangle difference array in LDS (4 dependent read-modify-writes per pair: latency).
// Here: v_mfma_f32_32x32x16_bf16 with A = mask (row = window, built in registers from the pairs' 64-bit window masks, exact 0 / 1),
// B = the pairs' gradient rows (column = channel; a lane's 8 consecutive pairs are 8 two-byte loads - the rows are 8 KiB apart),
// f32 accumulation over the pairs in list order of 16-pair steps, bf16 output.  One workgroup per (role, object, 1024 channels),
// a wave per 32-channel tile.  Rows go to the window-major space: gwm[goff[w] + role*n_obj + o].
constexpr int GSUM_MAXP = 512;
__global__ __launch_bounds__(256) void fc1_gsum_kernel(const u16* __restrict__ dh, const int* __restrict__ bbox, const int* __restrict__ sub,
                                                       const int* __restrict__ obj, const int* __restrict__ sub_ptr,
                                                       const int* __restrict__ sub_list, const int* __restrict__ obj_ptr,
                                                       const int* __restrict__ obj_list, const int* __restrict__ goff, int n_obj,
                                                       u16* __restrict__ gwm) {
    __shared__ int s_pair[GSUM_MAXP];
    __shared__ unsigned long long s_mask[GSUM_MAXP];
    const int chunk = blockIdx.x & 3;
    const int ps = blockIdx.x >> 2;
    const int role = ps >= n_obj ? 1 : 0, o = ps - role * n_obj;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int* ptr = role ? obj_ptr : sub_ptr;
    const int* list = role ? obj_list : sub_list;
    const int i0 = ptr[o], i1 = ptr[o + 1];
    const WRect ro = object_windows(bbox + 4 * o);
    unsigned long long own = 0ull;
    for (int y = ro.y0; y < ro.y1; ++y) own |= (((1ull << ro.x1) - (1ull << ro.x0)) & 0xffull) << (8 * y);
    const int l31 = lane & 31, kh = lane >> 5;
    for (int tile = 0; tile < 8; ++tile) {
        const int c = chunk * 1024 + (wid * 8 + tile) * 32 + l31;
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int base = i0; base < i1; base += GSUM_MAXP) {
            const int n = min(GSUM_MAXP, i1 - base);
            if (tile == 0 || i1 - i0 > GSUM_MAXP) {          // the usual case (<= 512 partners): staged once, before the first tile
                __syncthreads();
                for (int k = threadIdx.x; k < n; k += 256) {
                    const int p = list[base + k];
                    const WRect r = object_windows(bbox + 4 * (role ? sub[p] : obj[p]));
                    unsigned long long pm = 0ull;
                    for (int y = r.y0; y < r.y1; ++y) pm |= (((1ull << r.x1) - (1ull << r.x0)) & 0xffull) << (8 * y);
                    s_pair[k] = p;
                    s_mask[k] = role ? (own & ~pm) : ~pm;
                }
                __syncthreads();
            }
            for (int k0 = 0; k0 < n; k0 += 16) {
                // this lane's 8 pairs of the 16-pair step: k0 + 8*kh + j
                s16x8 bfr, af0, af1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + 8 * kh + j;
                    const bool ok = k < n;
                    const unsigned long long m = ok ? s_mask[k] : 0ull;
                    bfr[j] = ok ? (short)dh[(long)s_pair[k] * 4096 + c] : (short)0;
                    af0[j] = ((m >> l31) & 1ull) ? (short)0x3F80 : (short)0;             // bf16 1.0: windows 0..31
                    af1[j] = ((m >> (32 + l31)) & 1ull) ? (short)0x3F80 : (short)0;      // windows 32..63
                }
                acc[0] = mfma32<ELEM_BF16>(af0, bfr, acc[0]);
                acc[1] = mfma32<ELEM_BF16>(af1, bfr, acc[1]);
            }
        }
        // C layout: column = lane & 31 (channel), row (window inside the 32) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                gwm[((long)goff[w] + ps) * 4096 + c] = f32_to_bf16_bits(acc[j][r]);
            }
    }
}

// X rows of the window-major gradient: gwm[dest[e]] = dh1[pair of entry e]; and the padding rows of every group are zeroed in gwm
// and in ywm_bf16 (they take part in the contraction of the weight gradient).
__global__ __launch_bounds__(256) void fc1_xrows_kernel(const u16* __restrict__ dh, const int* __restrict__ gather, const int* __restrict__ dest,
                                                        int n_entries, const int* __restrict__ goff, const int* __restrict__ gend,
                                                        u16* __restrict__ gwm, u16* __restrict__ ywm_bf, long n_items) {
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        if (it < n_entries) {
            const long src = (long)(gather[it] >> 6) * 4096, dst = (long)dest[it] * 4096;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                *reinterpret_cast<uint4*>(gwm + dst + u * 512 + lane * 8) = *reinterpret_cast<const uint4*>(dh + src + u * 512 + lane * 8);
        } else {
            const long t = it - n_entries;                       // (group, padding slot 0..255)
            const int g = (int)(t >> 8);
            const long row = (long)gend[g] + (t & 255);
            if (row < goff[g + 1]) {
                const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u) *reinterpret_cast<uint4*>(gwm + row * 4096 + u * 512 + lane * 8) = z;
                *reinterpret_cast<uint4*>(ywm_bf + row * 1024 + lane * 8) = z;
                *reinterpret_cast<uint4*>(ywm_bf + row * 1024 + 512 + lane * 8) = z;
            }
        }
    }
}


// ================================================================================================ linear pairs (sixth identity)
// z_ij = maxpool2(relu(U_i + V_j)) equals z_(i,bg) wherever V_j is the background's and z_(bg,j) wherever U_i is.  When the two
// objects' regions of influence on the 16-grid (D16: the box, +-1 pixel for conv2_1, pooled) are DISJOINT, every pixel is one or the
// other (or both = the all-background value), i.e.  z_ij = z_(i,bg) + z_(bg,j) - z_(bg,bg)  exactly, pixel by pixel - and conv3 is
// linear up to its ReLU, so on the pair's X windows
//     pre_ij = pre_(i,bg) + pre_(bg,j) - pre_(bg,bg)          (pre = conv3 output before bias / ReLU / max-pool)
// in real arithmetic (in f32: three accumulations added instead of one - round-off).  Such a pair needs NO convolution of its own: its
// X windows (rectangles still overlap - conv3's 3x3 and the second pooling widen them - although the inputs do not) are combined from
// pre-activations the per-object window entries produce anyway.  On the benchmark's boxes 12.4 % of the X windows belong to such
// pairs.  Backward = autodiff: the un-pooled gradient of a combined window is ADDED to the un-pooled gradient rows of the two
// per-object entries and SUBTRACTED from the image's background map; the pair contributes no rows to the column GEMMs, no z / dz.
struct PRect { int x0, x1, y0, y1; };
__device__ __forceinline__ void axis_d16(int b0, int b1, int& p0, int& p1) {
    int lo = b0 < 0 ? 0 : b0, hi = b1 > 32 ? 32 : b1;
    if (hi <= lo) { p0 = 0; p1 = 0; return; }
    lo = lo > 0 ? lo - 1 : 0;  hi = hi < 32 ? hi + 1 : 32;
    p0 = lo >> 1;              p1 = (hi + 1) >> 1;
}
// true when the pair's D16 rectangles do not intersect (caller: the X rectangle is not empty)
__device__ __forceinline__ bool d16_disjoint(const int* __restrict__ bi, const int* __restrict__ bj) {
    int ax0, ax1, ay0, ay1, bx0, bx1, by0, by1;
    axis_d16(bi[0], bi[1], ax0, ax1); axis_d16(bi[2], bi[3], ay0, ay1);
    axis_d16(bj[0], bj[1], bx0, bx1); axis_d16(bj[2], bj[3], by0, by1);
    return !(max(ax0, bx0) < min(ax1, bx1) && max(ay0, by0) < min(ay1, by1));
}

// count_all[p] = |X_p|; count_conv[p] = |X_p| unless the pair is linear (then 0); count_lin[p] = |X_p| for linear pairs, else 0;
// pixrect_conv[p] = the pixels the pair's own z / dz are needed at (none for a linear pair)
__global__ __launch_bounds__(256) void shared_count3_kernel(const int* __restrict__ bbox, const int* __restrict__ sub, const int* __restrict__ obj,
                                                            int n_pairs, int* __restrict__ count_all, int* __restrict__ count_conv,
                                                            int* __restrict__ count_lin, int* __restrict__ pixrect_conv) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const int* bi = bbox + 4 * sub[p];
    const int* bj = bbox + 4 * obj[p];
    const WRect x = pair_windows(object_windows(bi), object_windows(bj));
    const int n = (x.x1 - x.x0) * (x.y1 - x.y0);
    const bool lin = n > 0 && d16_disjoint(bi, bj);
    count_all[p] = n;
    count_conv[p] = lin ? 0 : n;
    count_lin[p] = lin ? n : 0;
    pixrect_conv[p] = lin ? 0 : pack_pixel_rect(x);
}

// gather[e] = pair*64 + window for the X windows of the pairs of one class (1: not linear, 2: linear), given that class's prefix counts
__global__ __launch_bounds__(256) void shared_fill_class_kernel(const int* __restrict__ bbox, const int* __restrict__ sub, const int* __restrict__ obj,
                                                                int n_pairs, const int* __restrict__ incl, int* __restrict__ gather, int cls) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const int* bi = bbox + 4 * sub[p];
    const int* bj = bbox + 4 * obj[p];
    const WRect x = pair_windows(object_windows(bi), object_windows(bj));
    if (x.x1 <= x.x0) return;
    const bool lin = d16_disjoint(bi, bj);
    if ((cls == 1 && lin) || (cls == 2 && !lin)) return;
    int e = p ? incl[p - 1] : 0;
    for (int wy = x.y0; wy < x.y1; ++wy)
        for (int wx = x.x0; wx < x.x1; ++wx) gather[e++] = p * 64 + wy * 8 + wx;
}

// index of window (wx, wy) inside the row-major list of a rectangle
__device__ __forceinline__ int rect_local(const WRect& r, int wx, int wy) { return (wy - r.y0) * (r.x1 - r.x0) + (wx - r.x0); }

// Forward of the linear pairs' windows.  raw [(n_pe + 64 n_img) * 4][1024] f32: conv3 pre-activations (no bias) of the per-object
// window entries in list order (pseudo-pair ps: entries incl_all[n_real + ps - 1] - incl_all[n_real - 1] ..) followed by every window
// of every image's background map.  One wavefront per listed window: 4 pixels x 1024 channels, lane = 16 channels.
__global__ __launch_bounds__(256) void windows_linear_fwd_kernel(const int* __restrict__ bbox, const int* __restrict__ sub, const int* __restrict__ obj,
                                                                 const int* __restrict__ obj_img, int n_obj, int n_real,
                                                                 const int* __restrict__ gather_l, const int* __restrict__ n_l,
                                                                 const int* __restrict__ incl_all, const int* __restrict__ dest_all,
                                                                 const float* __restrict__ raw, long n_pe, const float* __restrict__ bias,
                                                                 u16* __restrict__ y, u16* __restrict__ y_bf, unsigned char* __restrict__ am,
                                                                 int* __restrict__ drow_out) {
    const int lane = threadIdx.x & 63;
    const int total = *n_l;
    const int base_ps = n_real ? incl_all[n_real - 1] : 0;
    for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += gridDim.x * 4) {
        const int code = gather_l[t];
        const int p = code >> 6, w = code & 63, wy = w >> 3, wx = w & 7;
        const int i = sub[p], j = obj[p];
        const WRect ri = object_windows(bbox + 4 * i), rj = object_windows(bbox + 4 * j);
        const WRect x = pair_windows(ri, rj);
        const long drow = dest_all[(p ? incl_all[p - 1] : 0) + rect_local(x, wx, wy)];
        if (drow_out && lane == 0) drow_out[t] = (int)drow;          // the backward's background side reads it instead of re-deriving it
        // exclusive prefix count at pair-index k = first list entry of that pair; per-object entries are counted from the first pseudo-pair
        const long ei = (long)(n_real + i > 0 ? incl_all[n_real + i - 1] : 0) - base_ps + rect_local(ri, wx, wy);
        const long ej = (long)incl_all[n_real + n_obj + j - 1] - base_ps + rect_local(rj, wx, wy);
        const long eb = n_pe + (long)obj_img[i] * 64 + w;
        const int c0 = lane * 16;
        float best[16];
        unsigned arg[16];                             // 32-bit in registers (as bytes: sub-dword instruction bloat, profiles/r05_expand_ab.txt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* a = raw + ((ei * 4 + q) << 10) + c0;
            const float* b = raw + ((ej * 4 + q) << 10) + c0;
            const float* g = raw + ((eb * 4 + q) << 10) + c0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 fa = reinterpret_cast<const float4*>(a)[v], fb = reinterpret_cast<const float4*>(b)[v];
                const float4 fg = reinterpret_cast<const float4*>(g)[v];
                const float s[4] = {(fa.x + fb.x) - fg.x, (fa.y + fb.y) - fg.y, (fa.z + fb.z) - fg.z, (fa.w + fb.w) - fg.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (q == 0) { best[4 * v + k] = s[k]; arg[4 * v + k] = 0; }
                    else if (s[k] > best[4 * v + k]) { best[4 * v + k] = s[k]; arg[4 * v + k] = (unsigned)q; }
                }
            }
        }
        uint4 o16[2], ob[2], oa;
        u16* oh = reinterpret_cast<u16*>(o16);
        u16* bh = reinterpret_cast<u16*>(ob);
        unsigned aw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float v = best[k] + bias[c0 + k];
            unsigned a = arg[k];
            if (!(v > 0.f)) { v = 0.f; a = 4; }                 // ReLU killed: no gradient path (same rule as the GEMM's pooled epilogue)
            oh[k] = f32_to_f16_bits(v);
            bh[k] = f32_to_bf16_bits(v);
            aw[k >> 2] |= a << (8 * (k & 3));
        }
        oa = make_uint4(aw[0], aw[1], aw[2], aw[3]);
        uint4* yo = reinterpret_cast<uint4*>(y + drow * 1024 + c0);
        yo[0] = o16[0]; yo[1] = o16[1];
        if (y_bf) { uint4* yb = reinterpret_cast<uint4*>(y_bf + drow * 1024 + c0); yb[0] = ob[0]; yb[1] = ob[1]; }
        if (am) *reinterpret_cast<uint4*>(am + (long)code * 1024 + c0) = oa;
    }
}

// un-pooled gradient of one combined window: acc[q][k] += (code == q) ? dy : 0 for the lane's 16 channels
__device__ __forceinline__ void linear_unpool_acc(float (&acc)[4][16], const u16* __restrict__ dyrow, const unsigned char* __restrict__ amrow,
                                                  int c0) {
    const uint4 g0 = *reinterpret_cast<const uint4*>(dyrow + c0), g1 = *reinterpret_cast<const uint4*>(dyrow + c0 + 8);
    const uint4 cd = *reinterpret_cast<const uint4*>(amrow + c0);
    u16 gh[16];
    *reinterpret_cast<uint4*>(gh) = g0;
    *reinterpret_cast<uint4*>(gh + 8) = g1;
    const unsigned char* ch = reinterpret_cast<const unsigned char*>(&cd);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float v = bf16_bits_to_f32(gh[k]);
        const unsigned code = ch[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][k] += (code == (unsigned)q) ? v : 0.f;
    }
}

// Backward, per-object side: dy3x rows of the per-object entry k (list position e0 + k: pseudo-pair ps, window w) += sum over the LINEAR
// pairs of the object whose X rectangle holds w of the un-pooled gradient of their combined window.  One wavefront per entry; f32
// sums in pair-list order, one bf16 rounding.
__global__ __launch_bounds__(256) void windows_linear_bwd_objects_kernel(const int* __restrict__ bbox, const int* __restrict__ sub,
                                                                         const int* __restrict__ obj, const int* __restrict__ sub_ptr,
                                                                         const int* __restrict__ sub_list, const int* __restrict__ obj_ptr,
                                                                         const int* __restrict__ obj_list, int n_obj, int n_real,
                                                                         const int* __restrict__ gather_c, int e0, int n_pe,
                                                                         const int* __restrict__ incl_all, const int* __restrict__ dest_all,
                                                                         const u16* __restrict__ dy, const unsigned char* __restrict__ am,
                                                                         u16* __restrict__ dy3x) {
    const int lane = threadIdx.x & 63, c0 = lane * 16;
    for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < n_pe; k += gridDim.x * 4) {
        const int code = gather_c[e0 + k];
        const int ps = (code >> 6) - n_real, w = code & 63, wy = w >> 3, wx = w & 7;
        const int role = ps >= n_obj ? 1 : 0, o = ps - role * n_obj;
        const int* ptr = role ? obj_ptr : sub_ptr;
        const int* list = role ? obj_list : sub_list;
        const int* bo = bbox + 4 * o;
        const WRect ro = object_windows(bo);
        float acc[4][16];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[q][c] = 0.f;
        bool any = false;
        // the object's pairs, 64 at a time: lane k tests pair k (linear, and its X rectangle holds w) and looks its gradient row up; a
        // ballot then walks the few that qualify (the serial walk over all ~63 partners was a chain of dependent look-ups: 0.44 ms)
        for (int base = ptr[o]; base < ptr[o + 1]; base += 64) {
            const int it = base + lane;
            int p = 0, drow = 0;
            bool ok = false;
            if (it < ptr[o + 1]) {
                p = list[it];
                const int* bp = bbox + 4 * (role ? sub[p] : obj[p]);
                const WRect rp = object_windows(bp);
                if (in_rect(rp, wx, wy) && d16_disjoint(bo, bp)) {
                    const WRect x = pair_windows(ro, rp);
                    drow = dest_all[(p ? incl_all[p - 1] : 0) + rect_local(x, wx, wy)];
                    ok = true;
                }
            }
            unsigned long long m = __ballot(ok);
            while (m) {
                const int bsel = __ffsll((long long)m) - 1;
                m &= m - 1;
                const long pp = __shfl(p, bsel), dr = __shfl(drow, bsel);
                linear_unpool_acc(acc, dy + dr * 1024, am + (pp * 64 + w) * 1024, c0);
                any = true;
            }
        }
        if (!any) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u16* row = dy3x + ((long)(e0 + k) * 4 + q) * 1024 + c0;
            u16 h[16];
            *reinterpret_cast<uint4*>(h) = *reinterpret_cast<const uint4*>(row);
            *reinterpret_cast<uint4*>(h + 8) = *reinterpret_cast<const uint4*>(row + 8);
#pragma unroll
            for (int c = 0; c < 16; ++c) h[c] = f32_to_bf16_bits(bf16_bits_to_f32(h[c]) + acc[q][c]);
            *reinterpret_cast<uint4*>(row) = *reinterpret_cast<const uint4*>(h);
            *reinterpret_cast<uint4*>(row + 8) = *reinterpret_cast<const uint4*>(h + 8);
        }
    }
}

// Backward, background side: the un-pooled gradient of every combined window is SUBTRACTED from its image's background map
// (dy3_bg [n_img][18][18][1024] bf16, interior already holding the map's own un-pooled gradient) and counts once for conv3's bias.
// order [n_l]: the listed windows sorted by (image, window) (stable: sums in list order); seg [64 n_img + 1] their ranges;
// lin_drow [n_l]: the window-major gradient row of every listed window (written by the forward).
// One wavefront per (image, window, quarter of the channels) - a (image, window) sums ~60 windows on the benchmark, and there are
// only 64 n_img of them; the first version re-derived every window's gradient row through five dependent look-ups (0.88 ms);
// bias_part [64 n_img][1024].
__global__ __launch_bounds__(256) void windows_linear_bwd_bg_kernel(const int* __restrict__ gather_l, const int* __restrict__ lin_drow,
                                                                    const int* __restrict__ order, const int* __restrict__ seg, int n_items,
                                                                    const u16* __restrict__ dy, const unsigned char* __restrict__ am,
                                                                    u16* __restrict__ dy3_bg, float* __restrict__ bias_part) {
    // One WORKGROUP per (image, window, quarter of the channels): its four wavefronts take the segment's windows in interleaved groups
    // of four (wave v: groups v, v + 4, ...) and their partial sums are added in wave order through LDS - a fixed order, so the result
    // does not depend on timing.  (Round 4: one wavefront per job walked the whole segment - 60 windows on average, several hundred
    // for a window inside many boxes - with a three-deep look-up chain per group: 0.43 ms for 30 k windows, all of it latency.)
    __shared__ float red[3][64][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int job = blockIdx.x; job < 4 * n_items; job += gridDim.x) {
        const int it = job >> 2, c0 = (job & 3) * 256 + lane * 4;
        const int b = it >> 6, w = it & 63, wy = w >> 3, wx = w & 7;
        float acc[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[q][c] = 0.f;
        const int s1 = seg[it + 1];
        for (int s0 = seg[it] + 4 * wv; s0 < s1; s0 += 16) {
            uint2 g[4];
            unsigned cd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = s0 + u < s1;
                const int t = order[ok ? s0 + u : s0];
                const long drow = lin_drow[t], code = gather_l[t];
                g[u] = ok ? *reinterpret_cast<const uint2*>(dy + drow * 1024 + c0) : make_uint2(0, 0);
                cd[u] = ok ? *reinterpret_cast<const unsigned*>(am + code * 1024 + c0) : 0x04040404u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u16* gh = reinterpret_cast<const u16*>(&g[u]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = bf16_bits_to_f32(gh[c]);
                    const unsigned k = (cd[u] >> (8 * c)) & 255u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q][c] += (k == (unsigned)q) ? v : 0.f;
                }
            }
        }
        if (wv > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) red[wv - 1][lane][q * 4 + c] = acc[q][c];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[q][c] += red[v][lane][q * 4 + c];
            float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int Y = 2 * wy + (q >> 1) + 1, X = 2 * wx + (q & 1) + 1;
                u16* row = dy3_bg + (((long)b * 18 + Y) * 18 + X) * 1024 + c0;
                uint2 hv = *reinterpret_cast<const uint2*>(row);
                u16* h = reinterpret_cast<u16*>(&hv);
#pragma unroll
                for (int c = 0; c < 4; ++c) { h[c] = f32_to_bf16_bits(bf16_bits_to_f32(h[c]) - acc[q][c]); bs[c] += acc[q][c]; }
                *reinterpret_cast<uint2*>(row) = hv;
            }
            *reinterpret_cast<float4*>(bias_part + (long)it * 1024 + c0) = make_float4(bs[0], bs[1], bs[2], bs[3]);
        }
        __syncthreads();
    }
}


// ================================================================================================ conv2 halves on the objects' own regions
// The same argument one layer down: object o's masked map equals the constant tanh(b1) outside its box, so its conv2 half U_o (V_o)
// equals the background's half at every 32-grid pixel whose 3x3 neighbourhood misses the box.  The pixels that can differ are the
// 2x2-pixel windows of the D16 rectangle (axis_d16: box, +-1 pixel, in units of 2 pixels): conv2_1 is computed on those windows only
// (gemm_nt_pp_kernel<.., ACG> on 32x32 maps, rows scattered back to their window-major place) and the other rows of U_o are copies
// of the background map of the object's image - identical inputs, identical arithmetic, identical bits.
__global__ __launch_bounds__(256) void conv2_regions_count_kernel(const int* __restrict__ bbox, int n_obj, int* __restrict__ count) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_obj) return;
    int x0, x1, y0, y1;
    axis_d16(bbox[4 * o], bbox[4 * o + 1], x0, x1);
    axis_d16(bbox[4 * o + 2], bbox[4 * o + 3], y0, y1);
    count[o] = (x1 > x0 && y1 > y0) ? (x1 - x0) * (y1 - y0) : 0;
}
__global__ __launch_bounds__(256) void conv2_regions_fill_kernel(const int* __restrict__ bbox, int n_obj, const int* __restrict__ incl,
                                                                 int* __restrict__ gather) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_obj) return;
    int x0, x1, y0, y1;
    axis_d16(bbox[4 * o], bbox[4 * o + 1], x0, x1);
    axis_d16(bbox[4 * o + 2], bbox[4 * o + 3], y0, y1);
    if (x1 <= x0 || y1 <= y0) return;
    int e = o ? incl[o - 1] : 0;
    for (int wy = y0; wy < y1; ++wy)
        for (int wx = x0; wx < x1; ++wx) gather[e++] = o * 256 + wy * 16 + wx;
}
// Backward of conv2 on the objects' GRADIENT regions (round 5).  The gradient of object o's conv2 half arrives through the pair
// contraction, which writes a pixel only inside the packed pixel rectangle of o's pseudo-pair - the pixels of the 16-grid within one
// pixel of the windows R_o (pack_pixel_rect(object_windows(box))): every pair of o lives inside it - so dU_o is zero outside that
// rectangle's 2x2-pixel cells of the 32-grid (40 % of the map on the benchmark's boxes).  List of (object, cell) for the conv2
// data gradient: the cells of the rectangle widened by ``dilate`` = 1 cell on every side (a 3x3 transposed convolution reaches one
// pixel further); the images' background objects (o >= n_real) collect gradient everywhere: all 256 cells.  (The WEIGHT gradient over
// the same cells - the TN block with both operands gathered by the list - was built and measured: 0.71-0.80 ms against 0.57 for the
// whole maps although it contracts over 40 % of the rows; the gathered TN staging costs more than the rows it skips.  Not kept.)  ONE workgroup: per-object counts, an exclusive scan in chunks of 1024 objects, fill; *n_out = list length.
__global__ __launch_bounds__(1024) void conv2_bwd_regions_kernel(const int* __restrict__ bbox, int n_real, int n_objx, int dilate,
                                                                 int* __restrict__ gather, int* __restrict__ n_out) {
    __shared__ int wave_tot[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int o0 = 0; o0 < n_objx; o0 += 1024) {
        const int o = o0 + tid;
        int X0 = 0, X1 = 0, Y0 = 0, Y1 = 0;
        if (o < n_real) {
            const WRect r = object_windows(bbox + 4 * o);
            if (r.x1 > r.x0) {
                Y0 = max(2 * r.y0 - 1 - dilate, 0); Y1 = min(2 * r.y1 + 1 + dilate, 16);
                X0 = max(2 * r.x0 - 1 - dilate, 0); X1 = min(2 * r.x1 + 1 + dilate, 16);
            }
        } else if (o < n_objx) { X1 = 16; Y1 = 16; }
        const int cnt = (X1 - X0) * (Y1 - Y0);
        int incl = cnt;                                  // inclusive scan inside the wavefront
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wid; ++w) before += wave_tot[w];
        int e = before + incl - cnt;
        for (int y = Y0; y < Y1; ++y)
            for (int x = X0; x < X1; ++x) gather[e++] = o * 256 + y * 16 + x;
        __syncthreads();
        if (tid == 1023) s_base = before + incl;
        __syncthreads();
    }
    if (tid == 0) *n_out = s_base;
}
// rows of object o outside its region <- the rows of the background object n_obj + obj_img[o]; one wavefront per (object, window): 4 KiB
__global__ __launch_bounds__(256) void conv2_fill_background_kernel(const int* __restrict__ bbox, const int* __restrict__ obj_img, int n_obj,
                                                                    uint4* __restrict__ uv, long n_items) {
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        const int o = (int)(it >> 8), w = (int)(it & 255), wy = w >> 4, wx = w & 15;
        int x0, x1, y0, y1;
        axis_d16(bbox[4 * o], bbox[4 * o + 1], x0, x1);
        axis_d16(bbox[4 * o + 2], bbox[4 * o + 3], y0, y1);
        if (wx >= x0 && wx < x1 && wy >= y0 && wy < y1) continue;               // computed for the object
        const uint4* src = uv + (((long)(n_obj + obj_img[o]) * 256 + w) * 4) * 64;   // a 512-channel f16 row = 64 uint4
        uint4* dst = uv + (((long)o * 256 + w) * 4) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * 64 + lane] = src[q * 64 + lane];
    }
}

static inline int grid_cap(long items, long per_block, int cap) {
    long b = (items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" {

int sgc_shared_windows_count(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, int* count, int* pixel_rect,
                             void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(shared_count_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx, n_pairs, count,
               pixel_rect);
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
    if (sgc_tuning().gather_pp) return launch_gemm_nt_pp_conv_gather<ELEM_F16, EPI_POOL>(p, (hipStream_t)stream);
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

int sgc_shared_windows_assemble_bwd(const int* bbox, const int* sub_idx, const int* obj_idx, const int* sub_ptr, const int* sub_list,
                                    const int* obj_ptr, const int* obj_list, int n_obj, const void* dy, void* dy_obj, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    const long items = 2L * n_obj * 64;
    SGC_LAUNCH(shared_assemble_bwd_kernel, dim3(grid_cap(items, 4, 262144)), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx,
               sub_ptr, sub_list, obj_ptr, obj_list, n_obj, (const u16*)dy, (u16*)dy_obj, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_windows_unpool(const void* dy, const unsigned char* argmax, const int* gather, const int* gather_n, const int* dest,
                       int entries_pad, void* dy3x, float* bias_part, int* n_parts, void* stream) {
    if (entries_pad <= 0) { if (n_parts) *n_parts = 0; return SGC_OK; }
    const int blocks = grid_cap(entries_pad, 4 * 8, 1024);
    if (n_parts) *n_parts = blocks;
    SGC_LAUNCH(windows_unpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, argmax, gather, gather_n, dest,
               entries_pad, (u16*)dy3x, bias_part, 0);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// the same for the entries entry0 .. entry0 + entries_pad - 1 only (dy3x row 0 = entry entry0): the entries in front of them go
// through the sparse forms of both backward GEMMs, which pack their operand from the pooled rows and need no un-pooled ones
int sgc_windows_unpool_from(const void* dy, const unsigned char* argmax, const int* gather, const int* gather_n, const int* dest, int entry0,
                            int entries_pad, void* dy3x, float* bias_part, int* n_parts, void* stream) {
    if (entries_pad <= 0) { if (n_parts) *n_parts = 0; return SGC_OK; }
    if (entry0 < 0) return SGC_ERR_ARG;
    const int blocks = grid_cap(entries_pad, 4 * 8, 1024);
    if (n_parts) *n_parts = blocks;
    SGC_LAUNCH(windows_unpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, argmax, gather + entry0, gather_n,
               dest ? dest + entry0 : nullptr, entries_pad, (u16*)dy3x, bias_part, entry0);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_windows_im2col(const void* z_pad_bf16, const int* gather, const int* gather_n, int entries_pad, void* zcol, void* stream) {
    if (entries_pad <= 0) return SGC_OK;
    const long rows = (long)entries_pad * 4;
    SGC_LAUNCH(windows_im2col_kernel, dim3(grid_cap(rows, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)z_pad_bf16,
               gather, gather_n, rows, (u16*)zcol);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// slabs [splits][1024][9*512] f32 = sum_rows dy3x[row][n] * zcol[row][(tap, c)]   (rows = 4 * entries_pad, a multiple of 64)
int sgc_windows_wgrad(const void* dy3x, const void* zcol, float* slabs, int rows, int splits, int* n_slabs, void* stream) {
    if (rows <= 0) { if (n_slabs) *n_slabs = 0; return SGC_OK; }
    TnParams p{};
    p.A = (const u16*)dy3x; p.B = (const u16*)zcol; p.C = slabs; p.M = 1024; p.N = 9 * 512; p.K = rows;
    p.lda = 1024; p.ldb = 9 * 512; p.ldc = 9 * 512; p.slab_stride = 1024L * 9 * 512;
    return launch_gemm_tn<ELEM_BF16, BMODE_PLAIN>(p, splits, n_slabs, (hipStream_t)stream);
}

// PATCH form of the two calls above: zpatch [entries_pad][16][512] bf16 (entries behind the list: zero rows), then
// slabs [splits][1024][9*512] f32 = sum_rows dy3x[row][n] * zpatch[window of the row][its pixel + tap][c]
int sgc_windows_im2patch(const void* z_pad_bf16, const int* gather, const int* gather_n, int entries_pad, void* zpatch, void* stream) {
    if (entries_pad <= 0) return SGC_OK;
    const long rows = (long)entries_pad * 4;
    SGC_LAUNCH(windows_im2patch_kernel<false>, dim3(grid_cap(rows, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)z_pad_bf16,
               gather, gather_n, rows, (u16*)zpatch, 0);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// the same from the forward's f16 maps (values rounded f16 -> bf16 on the way): no bf16 copy of the real pairs' z is needed
int sgc_windows_im2patch_f16(const void* z_pad_f16, const int* gather, const int* gather_n, int entries_pad, void* zpatch, void* stream) {
    if (entries_pad <= 0) return SGC_OK;
    const long rows = (long)entries_pad * 4;
    SGC_LAUNCH(windows_im2patch_kernel<true>, dim3(grid_cap(rows, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)z_pad_f16,
               gather, gather_n, rows, (u16*)zpatch, 0);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// ... of the entries e0 .. e0 + entries - 1 only (zpatch row 0 = entry e0): the tail of the list behind the windows whose patches the sparse
// weight gradient gathers itself (sgc_windows_wgrad_gather_sparse)
int sgc_windows_im2patch_f16_from(const void* z_pad_f16, const int* gather, const int* gather_n, int e0, int entries, void* zpatch, void* stream) {
    if (entries <= 0) return SGC_OK;
    const long rows = (long)entries * 4;
    SGC_LAUNCH(windows_im2patch_kernel<true>, dim3(grid_cap(rows, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)z_pad_f16,
               gather, gather_n, rows, (u16*)zpatch, e0);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_windows_wgrad_patch(const void* dy3x, const void* zpatch, float* slabs, int rows, int splits, int* n_slabs, void* stream) {
    if (rows <= 0) { if (n_slabs) *n_slabs = 0; return SGC_OK; }
    if (rows & 63) return SGC_ERR_ARG;
    TnParams p{};
    p.A = (const u16*)dy3x; p.B = (const u16*)zpatch; p.C = slabs; p.M = 1024; p.N = 9 * 512; p.K = rows; p.Cin = 512;
    p.lda = 1024; p.ldb = 0; p.ldc = 9 * 512; p.slab_stride = 1024L * 9 * 512;
    if (splits <= 0) splits = tn_auto_splits((p.M / 256) * (p.N / 256), p.K >> 6);
    return launch_gemm_tn_pp<ELEM_BF16, BMODE_PATCH, 0>(p, splits, n_slabs, (hipStream_t)stream);
}

// col [rows][9*512] bf16 = dy3x [rows][1024] * w3col[(tap, c)][1024]^T
int sgc_windows_dgrad_cols(const void* dy3x, const void* w3col, void* col, int rows, void* stream) {
    if (rows <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)dy3x; p.B = (const u16*)w3col; p.C = col; p.M = rows; p.N = 9 * 512; p.K = 1024;
    p.lda = 1024; p.ldb = 1024; p.ldc = 9 * 512;
    return launch_gemm_nt<ELEM_BF16, AMODE_PLAIN, EPI_STORE>(p, (hipStream_t)stream);
}

int sgc_windows_patch_sum2(const void* patch, int n16, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl, int n_pairs,
                           void* dz, void* stream);
int sgc_windows_patch_sum_objects2(const void* patch, int n16, const int* bbox, int n_obj, int n_real, const int* count_incl, void* dz, void* stream);
int sgc_fc1_assemble_ordered(const float* S, const float* owm, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl,
                             const int* dest, int n_obj, const float* bias, int drop_enable, unsigned drop_seed, void* h1, int n_pairs,
                             const float* own_rect_sums, const int* pair_order, void* stream);
int sgc_windows_patch_slots(void) { return PATCH_SLOTS; }
// patch [entries][PATCH_SLOTS][512] bf16: gradient of the 4 x 4 input patch of every listed window (entries = list length, padded freely);
// w3patch: for pp = 4 py + px in order, [512 c_in][combinations x 1024 c_out] bf16 with the combinations (own pixel q, tap t), q + t = pp,
// ordered (qy, ky) major, (qx, kx) minor, q ascending (engine.prep_bwd_weights)
int sgc_windows_dgrad_patches(const void* dy3x, const void* w3patch, void* patch, int entries, void* stream) {
    if (entries <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)dy3x; p.B = (const u16*)w3patch; p.C = patch; p.M = entries; p.N = PATCH_SLOTS * 512; p.K = 4096;
    p.lda = 4 * 1024; p.ldb = 0; p.ldc = PATCH_SLOTS * 512; p.seg_stride = 1024; p.seg_bpad = 0; p.seg_split = 1;
    p.patch_gn = 2;        // XCD patches of 16 M tiles x (slot, both channel halves): 7.37 against 7.50 ms for 4 x 8 (tools/dgrad_patch_microbench.py)
    return launch_gemm_nt_pp_seg<ELEM_BF16>(p, (hipStream_t)stream);
}
int sgc_windows_patch_sum(const void* patch, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl, int n_pairs,
                          void* dz, void* stream) {
    return sgc_windows_patch_sum2(patch, 0, bbox, sub_idx, obj_idx, count_incl, n_pairs, dz, stream);
}
int sgc_windows_patch_sum_objects(const void* patch, const int* bbox, int n_obj, int n_real, const int* count_incl, void* dz, void* stream) {
    return sgc_windows_patch_sum_objects2(patch, 0, bbox, n_obj, n_real, count_incl, dz, stream);
}
// the same with the first n16 entries in the 16-row layout of the sparse data gradient (see patch_sum_rect)
int sgc_windows_patch_sum2(const void* patch, int n16, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl, int n_pairs,
                           void* dz, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(windows_patch_sum_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, (const u16*)patch, bbox, sub_idx, obj_idx,
               count_incl, (u16*)dz, n16);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_windows_patch_sum_objects2(const void* patch, int n16, const int* bbox, int n_obj, int n_real, const int* count_incl, void* dz, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(windows_patch_sum_objects_kernel, dim3(2 * n_obj), dim3(256), 0, (hipStream_t)stream, (const u16*)patch, bbox, n_obj, n_real,
               count_incl, (u16*)dz, n16);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_windows_col2im(const void* col, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl, int n_pairs,
                       void* dz, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(windows_col2im_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, (const u16*)col, bbox, sub_idx, obj_idx,
               count_incl, (u16*)dz);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_pair_contract_windows(const void* dz, const unsigned char* amz, const int* ptr, const int* list, const int* pixel_rect,
                              const int* img_ptr, int role, int n_real_pairs, int n_obj, int n_img, int bg_maps, void* dU_pad,
                              void* stream) {
    if (n_obj <= 0) return SGC_OK;
    const long items = (long)(n_obj + n_img) * 256;
#ifdef SGC_EXPERIMENTS
    static const int pipe = [] { const char* e = getenv("SGC_CONTRACT_PIPE"); return e ? atoi(e) : 1; }();      // A/B: profiles/r05_contract_pipe_ab.txt
#else
    constexpr int pipe = 1;
#endif
    if (pipe)
        SGC_LAUNCH(pair_contract_windows_kernel<true>, dim3(grid_cap(items, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)dz, amz,
               ptr, list, pixel_rect, img_ptr, role, n_real_pairs, n_obj, bg_maps, (u16*)dU_pad, items);
    else
        SGC_LAUNCH(pair_contract_windows_kernel<false>, dim3(grid_cap(items, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)dz, amz,
               ptr, list, pixel_rect, img_ptr, role, n_real_pairs, n_obj, bg_maps, (u16*)dU_pad, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// ---- conv3 forward writing the window-major row space (see "fc1 over shared windows")
// pseudo-pairs: y / y_bf16 row of (pseudo-pair ps, window w) = goff[w] + ps; argmax stays [ps*64 + w]
int sgc_conv3_relu_pool_wm(const void* z_pad, const void* w3r, const float* b3, const int* goff, void* ywm, unsigned char* argmax,
                           void* ywm_bf16, int n_pairs, void* stream) {
    NtParams p{};
    p.A = (const u16*)z_pad; p.B = (const u16*)w3r; p.C = ywm; p.M = n_pairs * 256; p.N = 1024; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 1024; p.lgS = 4; p.Cin = 512; p.bias = b3; p.argmax = argmax; p.C2 = (u16*)ywm_bf16; p.wm_goff = goff;
    return launch_gemm_nt<ELEM_F16, AMODE_CONV, EPI_POOL>(p, (hipStream_t)stream);
}
// X windows: y / y_bf16 of entry e go to row dest[e]; argmax to row gather[e] of the pair-major argmax
int sgc_conv3_relu_pool_windows_wm(const void* z_pad, const void* w3r, const float* b3, const int* gather, const int* gather_n,
                                   const int* dest, int max_entries, void* ywm, unsigned char* argmax, void* ywm_bf16, void* stream) {
    if (max_entries <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)z_pad; p.B = (const u16*)w3r; p.C = ywm; p.M = max_entries * 4; p.N = 1024; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 1024; p.lgS = 4; p.Cin = 512; p.bias = b3; p.argmax = argmax; p.C2 = (u16*)ywm_bf16;
    p.gather = gather; p.gather_n = gather_n; p.dest = dest;
    if (sgc_tuning().gather_pp) return launch_gemm_nt_pp_conv_gather<ELEM_F16, EPI_POOL>(p, (hipStream_t)stream);
    return launch_gemm_nt_cfg<ELEM_F16, AMODE_CONV_GATHER, EPI_POOL, 2, 4, 4, 2>(p, (hipStream_t)stream);
}
// the same, also writing the accumulators (before bias / ReLU / pooling) of the entries e >= raw_first to raw[(e - raw_first)*4 + pixel]
int sgc_conv3_relu_pool_windows_wm_raw(const void* z_pad, const void* w3r, const float* b3, const int* gather, const int* gather_n,
                                       const int* dest, int max_entries, void* ywm, unsigned char* argmax, void* ywm_bf16, float* raw,
                                       int raw_first, void* stream) {
    if (max_entries <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)z_pad; p.B = (const u16*)w3r; p.C = ywm; p.M = max_entries * 4; p.N = 1024; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 1024; p.lgS = 4; p.Cin = 512; p.bias = b3; p.argmax = argmax; p.C2 = (u16*)ywm_bf16;
    p.gather = gather; p.gather_n = gather_n; p.dest = dest; p.raw = raw; p.raw_first = raw_first;
    return launch_gemm_nt_pp_conv_gather<ELEM_F16, EPI_POOL>(p, (hipStream_t)stream);
}
int sgc_fc1_products_pitch(void) { return owm_pitch(); }
// owm [rows][pitch] f32 (columns 0..4095) = ywm [rows][1024] f16 * w1p[:, g*1024 .. +1024]^T, g = tile_group[row / 256]  (rows a multiple of 256)
int sgc_fc1_windows_gemm(const void* ywm, const void* w1p, const int* tile_group, float* owm, int rows, void* stream) {
    if (rows <= 0) return SGC_OK;
    if (rows & 255) return SGC_ERR_ARG;
    NtParams p{};
    p.A = (const u16*)ywm; p.B = (const u16*)w1p; p.C = owm; p.M = rows; p.N = 4096; p.K = 1024;
    p.lda = 1024; p.ldb = 65536; p.ldc = owm_pitch(); p.tile_group = tile_group; p.group_stride = 1024;
    if (sgc_tuning().f32_swap) return launch_gemm_nt_pp<ELEM_F16, EPI_STORE_F32T>(p, (hipStream_t)stream);
    return launch_gemm_nt_pp<ELEM_F16, EPI_STORE_F32>(p, (hipStream_t)stream);
}
// the same with the pair-specific rows (index inside their group >= n_pseudo) written as f16 to oxh [rows][4096] instead of f32 to owm: each
// is one of the ~7 products a pair adds to 13 f32 prefix-sum vectors before its sum is rounded to f16 anyway (model.py:148: h1), and the
// f32 store of those rows was a third of the launch; the per-object rows (2-D prefix sums, differences of large sums) stay f32
int sgc_fc1_windows_gemm_x16(const void* ywm, const void* w1p, const int* tile_group, const int* goff, int n_pseudo, float* owm, void* oxh,
                             int rows, void* stream) {
    if (rows <= 0) return SGC_OK;
    if (rows & 255) return SGC_ERR_ARG;
    NtParams p{};
    p.A = (const u16*)ywm; p.B = (const u16*)w1p; p.C = owm; p.M = rows; p.N = 4096; p.K = 1024;
    p.lda = 1024; p.ldb = 65536; p.ldc = owm_pitch(); p.tile_group = tile_group; p.group_stride = 1024;
    p.Cx = (u16*)oxh; p.x_first = n_pseudo; p.wm_goff = goff;
    return launch_gemm_nt_pp<ELEM_F16, EPI_STORE_F32T>(p, (hipStream_t)stream);
}
int sgc_fc1_integral(const float* owm, const int* goff, int n_pseudo, float* S, void* stream) {
    if (n_pseudo <= 0) return SGC_OK;
    SGC_LAUNCH(fc1_integral_kernel, dim3((unsigned)(((long)n_pseudo * 4096 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, owm, owm_pitch(), goff,
               n_pseudo, S);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_fc1_own_rect_sums(const float* S, const int* bbox, int n_obj, float* own, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(fc1_own_rect_kernel, dim3((unsigned)(((long)n_obj * 4096 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, bbox, n_obj, own);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_fc1_assemble(const float* S, const float* owm, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl,
                     const int* dest, int n_obj, const float* bias, int drop_enable, unsigned drop_seed, void* h1, int n_pairs,
                     const float* own_rect_sums, void* stream) {
    return sgc_fc1_assemble_ordered(S, owm, bbox, sub_idx, obj_idx, count_incl, dest, n_obj, bias, drop_enable, drop_seed, h1, n_pairs,
                                    own_rect_sums, nullptr, stream);
}
// the same rows, assembled in the order of ``pair_order`` (a permutation of the pairs: e.g. sorted by subject; NULL = pair order)
int sgc_fc1_assemble_ordered(const float* S, const float* owm, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl,
                             const int* dest, int n_obj, const float* bias, int drop_enable, unsigned drop_seed, void* h1, int n_pairs,
                             const float* own_rect_sums, const int* pair_order, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(fc1_assemble_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, S, owm, owm_pitch(), bbox, sub_idx, obj_idx, count_incl, dest,
               n_obj, bias, drop_enable, drop_seed, 2.f, (u16*)h1, own_rect_sums, pair_order, (const u16*)nullptr);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// the same with the pairs' X products read from the f16 rows of sgc_fc1_windows_gemm_x16
int sgc_fc1_assemble_x16(const float* S, const void* oxh, const int* bbox, const int* sub_idx, const int* obj_idx, const int* count_incl,
                         const int* dest, int n_obj, const float* bias, int drop_enable, unsigned drop_seed, void* h1, int n_pairs,
                         const float* own_rect_sums, const int* pair_order, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    if (!oxh) return SGC_ERR_ARG;
    SGC_LAUNCH(fc1_assemble_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, S, (const float*)nullptr, owm_pitch(), bbox, sub_idx, obj_idx,
               count_incl, dest, n_obj, bias, drop_enable, drop_seed, 2.f, (u16*)h1, own_rect_sums, pair_order, (const u16*)oxh);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// ---- fc1 backward over the window-major rows
// ---- conv2 halves on the objects' own regions (see above)
int sgc_conv2_regions_count(const int* bbox, int n_obj, int* count, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(conv2_regions_count_kernel, dim3((n_obj + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, n_obj, count);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_conv2_regions_fill(const int* bbox, int n_obj, const int* count_incl, int* gather, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(conv2_regions_fill_kernel, dim3((n_obj + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, n_obj, count_incl, gather);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// uv rows 4*gather[e] + q (window-major rows of [n][1024][512]) = conv3x3(a_pad, w2r) (+ bias) for the listed 2x2-pixel windows
int sgc_conv2_object_regions(const void* a_pad, const void* w2r, const float* bias, const int* gather, const int* gather_n,
                             int max_entries, void* uv, void* stream) {
    if (max_entries <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)a_pad; p.B = (const u16*)w2r; p.C = uv; p.M = max_entries * 4; p.N = 512; p.K = 9 * 128;
    p.ldb = 9 * 128; p.ldc = 512; p.lgS = 5; p.Cin = 128; p.bias = bias; p.gather = gather; p.gather_n = gather_n;
    return launch_gemm_nt_pp_conv_gather<ELEM_F16, EPI_STORE>(p, (hipStream_t)stream);
}
// list [<= n_objx * 256] of (object * 256 + cell) for the conv2 backward over the objects' gradient regions, *n_out its length
int sgc_conv2_bwd_regions(const int* bbox, int n_real, int n_objx, int dilate, int* gather, int* n_out, void* stream) {
    if (n_objx <= 0 || n_real < 0 || n_real > n_objx || dilate < 0) return SGC_ERR_ARG;
    SGC_LAUNCH(conv2_bwd_regions_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, bbox, n_real, n_objx, dilate, gather, n_out);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// da rows 4*gather[e] + q (window-major rows of [n][1024][128] bf16) = the conv2 data gradient (sgc_conv2_dgrad) on the listed cells
// only; rows of unlisted cells are NOT written (the caller zero-fills da: the gradient there is exactly zero)
int sgc_conv2_dgrad_regions(const void* dU_pad, const void* wd2, const int* gather, const int* gather_n, int max_entries, void* da,
                            void* stream) {
    if (max_entries <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)dU_pad; p.B = (const u16*)wd2; p.C = da; p.M = max_entries * 4; p.N = 128; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 128; p.lgS = 5; p.Cin = 512; p.gather = gather; p.gather_n = gather_n;
    return launch_gemm_nt_cfg<ELEM_BF16, AMODE_CONV_GATHER, EPI_STORE, 2, 2, 2, 2>(p, (hipStream_t)stream);
}
int sgc_conv2_fill_background(const int* bbox, const int* obj_img, int n_obj, void* uv, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    const long items = (long)n_obj * 256;
    SGC_LAUNCH(conv2_fill_background_kernel, dim3(grid_cap(items, 4, 131072)), dim3(256), 0, (hipStream_t)stream, bbox, obj_img, n_obj,
               (uint4*)uv, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// ---- linear pairs (see "linear pairs (sixth identity)" above)
int sgc_shared_windows_count3(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, int* count_all, int* count_conv,
                              int* count_linear, int* pixel_rect_conv, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(shared_count3_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx, n_pairs,
               count_all, count_conv, count_linear, pixel_rect_conv);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_shared_windows_fill_class(const int* bbox, const int* sub_idx, const int* obj_idx, int n_pairs, const int* count_incl, int* gather,
                                  int cls, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    if (cls != 1 && cls != 2) return SGC_ERR_ARG;
    SGC_LAUNCH(shared_fill_class_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx, n_pairs,
               count_incl, gather, cls);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// raw [4 * max_entries][1024] f32 = conv3x3(z_pad, w3r) of the listed windows, NO bias / ReLU / pooling (row 4e + q = pixel q of entry e)
int sgc_conv3_windows_raw(const void* z_pad, const void* w3r, const int* gather, const int* gather_n, int max_entries, float* raw,
                          void* stream) {
    if (max_entries <= 0) return SGC_OK;
    NtParams p{};
    p.A = (const u16*)z_pad; p.B = (const u16*)w3r; p.C = raw; p.M = max_entries * 4; p.N = 1024; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 1024; p.lgS = 4; p.Cin = 512; p.gather = gather; p.gather_n = gather_n;
    return launch_gemm_nt_pp_conv_gather<ELEM_F16, EPI_STORE_F32>(p, (hipStream_t)stream);
}
int sgc_windows_linear_forward(const int* bbox, const int* sub_idx, const int* obj_idx, const int* obj_img, int n_obj, int n_real_pairs,
                               const int* gather_linear, const int* n_linear, int max_linear, const int* count_incl_all,
                               const int* dest_all, const float* raw, long n_object_entries, const float* b3, void* ywm, void* ywm_bf16,
                               unsigned char* argmax, int* dest_linear, void* stream) {
    if (max_linear <= 0) return SGC_OK;
    SGC_LAUNCH(windows_linear_fwd_kernel, dim3(grid_cap(max_linear, 4, 65536)), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx,
               obj_img, n_obj, n_real_pairs, gather_linear, n_linear, count_incl_all, dest_all, raw, n_object_entries, b3, (u16*)ywm,
               (u16*)ywm_bf16, argmax, dest_linear);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_windows_linear_backward_objects(const int* bbox, const int* sub_idx, const int* obj_idx, const int* sub_ptr, const int* sub_list,
                                        const int* obj_ptr, const int* obj_list, int n_obj, int n_real_pairs, const int* gather_conv,
                                        int first_object_entry, int n_object_entries, const int* count_incl_all, const int* dest_all,
                                        const void* dywm, const unsigned char* argmax, void* dy3x, void* stream) {
    if (n_object_entries <= 0) return SGC_OK;
    SGC_LAUNCH(windows_linear_bwd_objects_kernel, dim3(grid_cap(n_object_entries, 4, 65536)), dim3(256), 0, (hipStream_t)stream, bbox,
               sub_idx, obj_idx, sub_ptr, sub_list, obj_ptr, obj_list, n_obj, n_real_pairs, gather_conv, first_object_entry,
               n_object_entries, count_incl_all, dest_all, (const u16*)dywm, argmax, (u16*)dy3x);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_windows_linear_backward_bg(const int* gather_linear, const int* dest_linear, const int* order, const int* segments, int n_img,
                                   const void* dywm, const unsigned char* argmax, void* dy3_bg_pad, float* bias_part, void* stream) {
    if (n_img <= 0) return SGC_OK;
    SGC_LAUNCH(windows_linear_bwd_bg_kernel, dim3(grid_cap(256L * n_img, 1, 65536)), dim3(256), 0, (hipStream_t)stream, gather_linear,
               dest_linear, order, segments, 64 * n_img, (const u16*)dywm, argmax, (u16*)dy3_bg_pad, bias_part);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_fc1_gsum(const void* dh1, const int* bbox, const int* sub_idx, const int* obj_idx, const int* sub_ptr, const int* sub_list,
                 const int* obj_ptr, const int* obj_list, const int* goff, int n_obj, void* gwm, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(fc1_gsum_kernel, dim3((unsigned)(2 * n_obj * 4)), dim3(256), 0, (hipStream_t)stream, (const u16*)dh1, bbox, sub_idx, obj_idx,
               sub_ptr, sub_list, obj_ptr, obj_list, goff, n_obj, (u16*)gwm);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_fc1_xrows(const void* dh1, const int* gather, const int* dest, int n_entries, const int* goff, const int* gend, void* gwm,
                  void* ywm_bf16, void* stream) {
    const long items = (long)n_entries + 64L * 256;
    SGC_LAUNCH(fc1_xrows_kernel, dim3(grid_cap(items, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)dh1, gather, dest,
               n_entries, goff, gend, (u16*)gwm, (u16*)ywm_bf16, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// dywm [rows][1024] bf16 = gwm [rows][4096] * (rows g*1024.. of w1pT [65536][4096])^T, g = tile_group[row / 256]
int sgc_fc1_windows_dgrad(const void* gwm, const void* w1pT, const int* tile_group, void* dywm, int rows, void* stream) {
    if (rows <= 0) return SGC_OK;
    if (rows & 255) return SGC_ERR_ARG;
    NtParams p{};
    p.A = (const u16*)gwm; p.B = (const u16*)w1pT; p.C = dywm; p.M = rows; p.N = 1024; p.K = 4096;
    p.lda = 4096; p.ldb = 4096; p.ldc = 1024; p.tile_group = tile_group; p.group_stride = 1024L * 4096;
    p.epi_lds = 1;
    return launch_gemm_nt_pp<ELEM_BF16, EPI_STORE>(p, (hipStream_t)stream);
}
// dw [4096][65536] f32 (columns in (window, channel) order): block g = gwm[group g]^T * ywm_bf16[group g]
int sgc_fc1_windows_wgrad(const void* gwm, const void* ywm_bf16, const int* goff, float* dw, int rows, void* stream) {
    if (rows <= 0) return SGC_OK;
    TnParams p{};
    p.A = (const u16*)gwm; p.B = (const u16*)ywm_bf16; p.C = dw; p.M = 4096; p.N = 1024; p.K = rows;
    p.lda = 4096; p.ldb = 1024; p.ldc = 65536; p.goff = goff;
    p.tiles_m = 16; p.tiles_n = 4; p.ktiles_per_split = 0; p.splits = 64; p.xcd_map = 0; p.xcd_patch = 1;
    if (sgc_tuning().fc1_wgrad_group_xcd) { p.xcd_map = 3; p.xcd_patch = 0; }       // a group's 64 tiles on one XCD (gemm_tn.h)
    auto kern = gemm_tn_pp_kernel<ELEM_BF16, BMODE_PLAIN, 0>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    SGC_LAUNCH(kern, dim3(64 * 64), dim3(512), 8 * 16384, (hipStream_t)stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// ---- second level: pseudo-pairs restricted to the windows of R_o (entries of pair n_real + ps), the rest from per-image background maps
int sgc_shared_objects_count(const int* bbox, int n_obj, int* count, int* pixel_rect, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(shared_count_objects_kernel, dim3((2 * n_obj + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, n_obj, count, pixel_rect);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_shared_objects_fill(const int* bbox, int n_obj, int n_real_pairs, const int* count_incl, int* gather, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(shared_fill_objects_kernel, dim3((2 * n_obj + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, n_obj, n_real_pairs,
               count_incl, gather);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_shared_objects_fill_rows(const int* bbox, const int* obj_img, int n_obj, const int* goff, const void* y_bg, const void* y_bg_bf16,
                                 const unsigned char* argmax_bg, void* ywm, void* ywm_bf16, unsigned char* argmax_ps, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    if ((ywm_bf16 && !y_bg_bf16) || (argmax_ps && !argmax_bg)) return SGC_ERR_ARG;
    const long rows = 2L * n_obj * 64;
    SGC_LAUNCH(shared_fill_object_rows_kernel, dim3(grid_cap(rows, 4, 131072)), dim3(256), 0, (hipStream_t)stream, bbox, obj_img, n_obj, goff,
               (const uint4*)y_bg, (const uint4*)y_bg_bf16, (const uint4*)argmax_bg, (uint4*)ywm, (uint4*)ywm_bf16, (uint4*)argmax_ps, rows);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_shared_objects_bg_grad(const int* bbox, const int* img_ptr, int n_obj, int n_img, const int* goff, const void* dywm, void* dy_bg,
                               void* stream) {
    if (n_img <= 0) return SGC_OK;
    const long items = (long)n_img * 64 * 2;
    SGC_LAUNCH(shared_bg_grad_kernel, dim3(grid_cap(items, 4, 65536)), dim3(256), 0, (hipStream_t)stream, bbox, img_ptr, n_obj, goff,
               (const u16*)dywm, (u16*)dy_bg, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_windows_col2im_objects(const void* col, const int* bbox, int n_obj, int n_real_pairs, const int* count_incl, void* dz, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    SGC_LAUNCH(windows_col2im_objects_kernel, dim3(2 * n_obj), dim3(256), 0, (hipStream_t)stream, (const u16*)col, bbox, n_obj, n_real_pairs,
               count_incl, (u16*)dz);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// The same weight gradient without the im2col buffer: the second operand's rows are gathered from z_pad_bf16 by the window list
// (gemm_tn_pp_kernel<BMODE_GATHER>).  gather must hold VALID windows for all rows / 4 entries (pad behind the list with any
// window of a fully written map: the matching rows of dy3x are zero).
int sgc_windows_wgrad_gather(const void* dy3x, const void* z_pad_bf16, const int* gather, float* slabs, int rows, int splits, int* n_slabs,
                             void* stream) {
    if (rows <= 0) { if (n_slabs) *n_slabs = 0; return SGC_OK; }
    if (rows & 63) return SGC_ERR_ARG;
    TnParams p{};
    p.A = (const u16*)dy3x; p.B = (const u16*)z_pad_bf16; p.C = slabs; p.M = 1024; p.N = 9 * 512; p.K = rows;
    p.lda = 1024; p.ldc = 9 * 512; p.slab_stride = 1024L * 9 * 512; p.lgS = 4; p.Cin = 512; p.gather = gather;
    if (splits <= 0) splits = tn_auto_splits(4 * 18, rows >> 6);
    return launch_gemm_tn_pp<ELEM_BF16, BMODE_GATHER, 0>(p, splits, n_slabs, (hipStream_t)stream);
}

}  // extern "C"
