// Device-side construction of the pair tables of a minibatch (row a2 of the hot path: the reference's nested
// graph_iter / edge_iter / direction loops, train_test.py:174-258) and of the per-pair loss coefficients
// (train_utils.py:64-94,116-157 + the running-sum step weights of train_test.py:219-258), plus the label-column
// gather / scatter of fc2 (model.py:152-168).  All of it is small integer work: the point is that nothing of size
// O(pairs) is built on the host or crosses PCIe - the host uploads O(images + max objects) integers per minibatch.
//
// Pair order (SURVEY 8a'): for g = 1..max_n-1, for e = 0..g-1, direction 1 (subject g, object e) then direction 2
// (subject e, object g); inside a direction-step the images with more than g objects, ascending.  With
// k_g = #{images : n > g} the block of graph_iter g starts at goff[g] = sum_{g'<g} 2 g' k_g' and holds 2g rows of
// k_g pairs; direction-step ordinal t = g(g-1) + row (every g < max_n has k_g >= 1, so no step is empty).
#include "common.h"

struct SceneParams {
    const int* n;          // [B] objects per image
    const int* img_ptr;    // [B+1]
    const int* goff;       // [max_n+1]
    int B, max_n, P, n_obj, pid_ld;
    const int* rel_tri;    // per image cat(relationships[b]) = n(n-1)/2 predicate ids, images concatenated (may be NULL)
    const float* dir_tri;  // same shape: subj_or_obj flags 1 / 0 / -1
    int *sub_idx, *obj_idx, *step, *image, *directed, *raw, *pid, *sub_list, *obj_list;
    int *obj_ptr, *obj_img, *step_ptr;
};

__device__ __forceinline__ int pairs_before_image(const int* __restrict__ n, int image) {
    int s = 0;
    for (int b = 0; b < image; ++b) s += n[b] * (n[b] - 1);
    return s;
}

__global__ __launch_bounds__(256) void scene_pairs_kernel(SceneParams s) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= s.P) return;
    int lo = 1, hi = s.max_n - 1;                       // largest g with goff[g] <= p
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s.goff[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int g = lo;
    const int k = (s.goff[g + 1] - s.goff[g]) / (2 * g);
    const int local = p - s.goff[g];
    const int row = local / k, col = local - row * k;
    const int e = row >> 1;
    const bool first = !(row & 1);
    int image = 0, seen = 0;
    for (int b = 0; b < s.B; ++b) {
        if (s.n[b] > g) {
            if (seen == col) { image = b; break; }
            ++seen;
        }
    }
    const int o0 = s.img_ptr[image], nb = s.n[image];
    const int sl = first ? g : e, ol = first ? e : g;
    const int sub = o0 + sl, obj = o0 + ol;
    s.sub_idx[p] = sub;
    s.obj_idx[p] = obj;
    s.step[p] = g * (g - 1) + row;
    s.image[p] = image;
    s.pid[(long)sub * s.pid_ld + ol] = p;
    const int base = pairs_before_image(s.n, image);
    // pairs of a subject (resp. object) sorted by pair index are its partners in ascending object order: for partner j < a
    // the step is (g = a, e = j), for j > a it is (g = j, e = a) - both increase with j, and every g = j > a block lies
    // behind the g = a block.  The CSR lists of the pair contraction are therefore closed-form.
    s.sub_list[base + sl * (nb - 1) + ol - (ol > sl)] = p;
    s.obj_list[base + ol * (nb - 1) + sl - (sl > ol)] = p;
    if (s.rel_tri) {
        const int idx = (base >> 1) + g * (g - 1) / 2 + e;
        const int r = s.rel_tri[idx];
        const float d = s.dir_tri[idx];
        s.raw[p] = r;
        s.directed[p] = (d == (first ? 1.f : 0.f)) ? r : -1;      // train_utils.py:169-187
    }
}

__global__ __launch_bounds__(256) void scene_objects_kernel(SceneParams s, int T) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)s.n_obj * s.pid_ld) {
        const int o = (int)(i / s.pid_ld), j = (int)(i - (long)o * s.pid_ld);
        int image = 0;
        while (image + 1 < s.B && s.img_ptr[image + 1] <= o) ++image;
        const int local = o - s.img_ptr[image], nb = s.n[image];
        if (j == local || j >= nb) s.pid[i] = -1;
        if (j == 0) {
            s.obj_ptr[o] = pairs_before_image(s.n, image) + local * (nb - 1);
            s.obj_img[o] = image;
            if (o == s.n_obj - 1) s.obj_ptr[s.n_obj] = s.P;
        }
    }
    if (i <= T) {                                        // first pair of every direction-step
        if (i == T) { s.step_ptr[T] = s.P; return; }
        const int t = (int)i;
        int g = (int)((1.0 + sqrt(1.0 + 4.0 * t)) * 0.5);
        while (g * (g - 1) > t) --g;
        while ((g + 1) * g <= t) ++g;
        const int k = (s.goff[g + 1] - s.goff[g]) / (2 * g);
        s.step_ptr[t] = s.goff[g] + (t - g * (g - 1)) * k;
    }
}

// One thread per direction-step (its pairs are contiguous, at most one per image): counts, class-weight sums and the
// per-pair coefficients  loss_i = -a*super[st] - b*rel[t] + c*BCE(conn, y)  in double, sequential pair order - the same
// arithmetic as the host reference implementation engine.loss_coefficients, bit for bit after the cast to f32.
__global__ __launch_bounds__(64) void loss_coefficients_kernel(const int* __restrict__ step_ptr, int T, const int* __restrict__ directed,
                                                               const float* __restrict__ class_weight, int ng, int npos, int hier,
                                                               double lambda_c, double lambda_nc, int* __restrict__ tgt,
                                                               float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc,
                                                               float* __restrict__ cy) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= T) return;
    const int p0 = step_ptr[t], p1 = step_ptr[t + 1];
    const double w = (double)(T - t);
    double n_conn = 0, wsum[3] = {0, 0, 0};
    for (int p = p0; p < p1; ++p) {
        const int d = directed[p];
        if (d >= 0) {
            n_conn += 1;
            const int seg = hier ? (d < ng ? 0 : (d < ng + npos ? 1 : 2)) : 0;
            wsum[seg] += (double)class_weight[d];
        }
    }
    const double n_nc = (double)(p1 - p0) - n_conn;
    for (int p = p0; p < p1; ++p) {
        const int d = directed[p];
        double a = 0, b = 0, c = 0;
        if (d >= 0) {
            const int seg = hier ? (d < ng ? 0 : (d < ng + npos ? 1 : 2)) : 0;
            c = w * lambda_c / n_conn;
            if (hier) a = w / n_conn;
            b = w * (double)class_weight[d] / wsum[seg];
        } else if (n_conn == 0) {
            c = w * lambda_c * lambda_nc / (n_nc > 1 ? n_nc : 1);
        }
        tgt[p] = d;
        ca[p] = (float)a; cb[p] = (float)b; cc[p] = (float)c; cy[p] = d >= 0 ? 1.f : 0.f;
    }
}

// Connectivity statistics of train_one_direction / evaluate_one_direction (train_utils.py:66-87,176-184) over all pairs:
// out[0] not connected, [1] connected, [2] predicted connected (sigmoid >= 0.5), [3] precision numerator (predicted
// connected whose unordered pair has a relation in either direction), [4] recall numerator (connected and predicted).
__global__ __launch_bounds__(256) void connectivity_stats_kernel(const float* __restrict__ conn, const int* __restrict__ directed,
                                                                 const int* __restrict__ raw, const unsigned char* __restrict__ included,
                                                                 int P, unsigned long long* __restrict__ out) {
    __shared__ unsigned int sh[5];
    if (threadIdx.x < 5) sh[threadIdx.x] = 0;
    __syncthreads();
    unsigned int c[5] = {0, 0, 0, 0, 0};
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        if (included && !included[p]) continue;
        const bool connected = directed[p] >= 0;
        // sigmoid(x) >= 0.5 in f32 exactly as torch evaluates it; round(sigmoid) == 1 needs sigmoid > 0.5 (round half to even)
        const float sg = 1.f / (1.f + expf(-conn[p]));
        const bool pred = sg >= 0.5f;
        c[0] += !connected; c[1] += connected; c[2] += pred; c[3] += pred && raw[p] != -1; c[4] += connected && sg > 0.5f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) atomicAdd(&sh[i], c[i]);
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&out[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

// Per-object label vectors: the one-hot / multi-hot columns of fc2 that model.py:152-168 concatenates to the 4096
// trunk features act as per-object additive rows.  One workgroup per object, one thread per fc2 output row.
__global__ __launch_bounds__(512) void label_vectors_kernel(const float* __restrict__ W, int ld, int col0, const long* __restrict__ cats,
                                                            const float* __restrict__ mh, int C, int S, float* __restrict__ lsub,
                                                            float* __restrict__ lobj) {
    const int o = blockIdx.x, r = threadIdx.x;
    const float* w = W + (long)r * ld + col0;
    const int c = (int)cats[o];
    float a = w[c], b = w[C + c];
    if (mh) {
        for (int k = 0; k < S; ++k) {
            const float m = mh[(long)o * S + k];
            if (m != 0.f) { a += m * w[2 * C + k]; b += m * w[2 * C + S + k]; }
        }
    }
    lsub[(long)o * 512 + r] = a;
    lobj[(long)o * 512 + r] = b;
}

// Transpose of label_vectors: gradient of the label columns of fc2.weight.  One workgroup per label column, objects
// visited in ascending order (deterministic), one thread per fc2 row.
__global__ __launch_bounds__(512) void label_grads_kernel(const float* __restrict__ dls, const float* __restrict__ dlo,
                                                          const long* __restrict__ cats_s, const long* __restrict__ cats_o,
                                                          const float* __restrict__ mh_s, const float* __restrict__ mh_o, int n_obj,
                                                          int C, int S, float* __restrict__ gW, int ld, int col0) {
    // Round 5: the objects that carry the column's label are FOUND by all 512 threads at once (thread = object, compacted in ascending
    // order into LDS) and only those rows are added - the first form walked all objects one scalar load and branch after the other in
    // every workgroup (0.13 ms for ~4 rows per column).  Same rows in the same order: same sums.
    __shared__ int s_obj[512];
    __shared__ float s_m[512];
    __shared__ int s_cnt[8];
    const int c = blockIdx.x, r = threadIdx.x, lane = r & 63, wv = r >> 6;
    const bool by_class = c < 2 * C;
    const bool sub = by_class ? c < C : c < 2 * C + S;
    const float* src = sub ? dls : dlo;
    const long* cats = sub ? cats_s : cats_o;
    const float* mh = sub ? mh_s : mh_o;
    const int key = by_class ? (sub ? c : c - C) : (sub ? c - 2 * C : c - 2 * C - S);
    float acc = 0.f;
    for (int base = 0; base < n_obj; base += 512) {
        const int o = base + r;
        float m = 0.f;
        bool ok = false;
        if (o < n_obj) {
            if (by_class) ok = (int)cats[o] == key;
            else { m = mh[(long)o * S + key]; ok = m != 0.f; }
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { off += k < wv ? s_cnt[k] : 0; total += s_cnt[k]; }
        if (ok) {
            const int pos = off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            s_obj[pos] = o;
            s_m[pos] = m;
        }
        __syncthreads();
        if (by_class) {
            for (int t = 0; t < total; ++t) acc += src[(long)s_obj[t] * 512 + r];
        } else {
            for (int t = 0; t < total; ++t) {
                const float mm = s_m[t];
                acc += mm * src[(long)s_obj[t] * 512 + r];
            }
        }
        __syncthreads();
    }
    gW[(long)r * ld + col0 + c] = acc;
}

__global__ __launch_bounds__(256) void slab_sum_ld_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols,
                                                          long ld_out, int slabs) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= slabs; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(long)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < slabs; ++k) s += in[(long)k * n + i];
        const long r = i / cols;
        out[r * ld_out + (i - r * cols)] = s;
    }
}

// Zero fill with 16-byte stores (workspace allocation: the zero halos of the padded activation / gradient tensors are written
// once, when a buffer is created; the kernels only ever write interiors).
__global__ __launch_bounds__(256) void fill_zero_kernel(uint4* __restrict__ p, long n16, unsigned char* __restrict__ tail, int ntail) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

// Two-level form of bucket_place (round 5): the list is cut into segments of BP_SEG entries and a workgroup owns one (key, segment):
// pass 1 counts the key's entries per segment, pass 2 (mode 1 only) turns the per-key totals into the keys' start offsets, pass 3 places
// with rank = entries of the key in earlier segments + rank inside the segment - the same ranks as the one-level kernel, which let ONE
// workgroup per key walk the whole list (64 workgroups x 56 iterations for the 228 k window entries of the benchmark: 0.09 ms of mostly
// idle chip, twice per step).  No assumption about the order of the list.
constexpr int BP_SEG = 8192;
__device__ __forceinline__ int bp_key(const int* __restrict__ codes, const int* __restrict__ sub_idx, const int* __restrict__ obj_img,
                                      int img_key, int e) {
    const int c = codes[e];
    return img_key ? obj_img[sub_idx[c >> 6]] * 64 + (c & 63) : (c & 63);
}
__global__ __launch_bounds__(1024) void bucket_count_kernel(const int* __restrict__ codes, int n, const int* __restrict__ sub_idx,
                                                            const int* __restrict__ obj_img, int img_key, int S, int* __restrict__ cnt) {
    __shared__ int s_c;
    const int k = blockIdx.x, sg = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int e0 = sg * BP_SEG, e1 = min(n, e0 + BP_SEG);
    int c = 0;
    for (int e = e0 + tid; e < e1; e += 1024) c += bp_key(codes, sub_idx, obj_img, img_key, e) == k ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (tid == 0) s_c = 0;
    __syncthreads();
    if (lane == 0 && c) atomicAdd(&s_c, c);                       // integer LDS atomics: order-independent
    __syncthreads();
    if (tid == 0) cnt[k * S + sg] = s_c;
}
// seg[k] = number of entries with a smaller key, seg[n_keys] = n (one workgroup; keys in chunks of 1024)
__global__ __launch_bounds__(1024) void bucket_starts_kernel(const int* __restrict__ cnt, int n_keys, int S, int n, int* __restrict__ seg) {
    __shared__ int wave_tot[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n_keys; k0 += 1024) {
        const int k = k0 + tid;
        int t = 0;
        if (k < n_keys)
            for (int sg = 0; sg < S; ++sg) t += cnt[k * S + sg];
        int v = t;                                                // inclusive scan inside the wavefront
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(v, d);
            if (lane >= d) v += u;
        }
        if (lane == 63) wave_tot[wid] = v;
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wid; ++w) off += wave_tot[w];
        if (k < n_keys) seg[k] = off + v - t;
        __syncthreads();
        if (tid == 1023) s_base = off + v;
        __syncthreads();
    }
    if (tid == 0) seg[n_keys] = n;
}
__global__ __launch_bounds__(1024) void bucket_place2_kernel(const int* __restrict__ codes, int n, const int* __restrict__ sub_idx,
                                                             const int* __restrict__ obj_img, int img_key, int S,
                                                             const int* __restrict__ cnt, const int* __restrict__ start, int* __restrict__ out,
                                                             int mode) {
    __shared__ int wave_tot[4][16];
    const int k = blockIdx.x, sg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (cnt[k * S + sg] == 0) return;                             // uniform
    int before = start[k];
    for (int s2 = 0; s2 < sg; ++s2) before += cnt[k * S + s2];
    const int e0 = sg * BP_SEG, e1 = min(n, e0 + BP_SEG);
    for (int it = e0; it < e1; it += 4096) {
        unsigned long long bal[4];
        bool m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = it + j * 1024 + tid;
            m[j] = e < e1 && bp_key(codes, sub_idx, obj_img, img_key, e) == k;
            bal[j] = __ballot(m[j]);
            if (lane == 0) wave_tot[j][wid] = __popcll(bal[j]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int off = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int t = wave_tot[j][w];
                off += w < wid ? t : 0;
                tot += t;
            }
            if (m[j]) {
                const int rank = before + off + __popcll(bal[j] & ((1ull << lane) - 1ull));
                const int e = it + j * 1024 + tid;
                if (mode == 0) out[e] = rank; else out[rank] = e;
            }
            before += tot;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
// ---- stable bucket placement of a window list (the row plan of conv3 / fc1 over shared windows, csrc/kernels_shared.hip) ----------
// codes[e] = pair*64 + window (list order = pair order).  key(e) = window (img_key = 0) or image(pair)*64 + window (img_key = 1:
// image = obj_img[sub_idx[pair]]).  One workgroup per key scans the list with ballots: rank_k(e) = number of earlier entries with
// the same key - what a stable sort by key gives, without the sort (torch.sort + searchsorted + gathers were a dozen small launches
// per step).  mode 0: out[e] = base[key] + rank (destination row of every entry, window-major row space);
// mode 1: out[start_k + rank] = e and seg[k] = start_k = number of entries with a smaller key (seg[n_keys] = n): the list ordered by
// key, stable.  No atomics; every output is written exactly once.
__global__ __launch_bounds__(1024) void bucket_place_kernel(const int* __restrict__ codes, int n, const int* __restrict__ sub_idx,
                                                            const int* __restrict__ obj_img, int img_key, int n_keys,
                                                            const int* __restrict__ base, int* __restrict__ out, int* __restrict__ seg, int mode) {
    __shared__ int wave_tot[4][16];
    __shared__ int s_less;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto key_of = [&](int e) {
        const int c = codes[e];
        return img_key ? obj_img[sub_idx[c >> 6]] * 64 + (c & 63) : (c & 63);
    };
    int start = 0;
    if (mode == 1) {                      // entries with a smaller key: a first pass over the list
        int less = 0;
        for (int e = tid; e < n; e += 1024) less += key_of(e) < k ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) less += __shfl_down(less, o, 64);
        if (tid == 0) s_less = 0;
        __syncthreads();
        if (lane == 0 && less) atomicAdd(&s_less, less);          // integer LDS atomics: order-independent
        __syncthreads();
        start = s_less;
        if (tid == 0) {
            seg[k] = start;
            if (k == n_keys - 1) seg[n_keys] = n;
        }
    } else {
        start = base[k];
    }
    int running = 0;
    for (int it = 0; it < n; it += 4096) {
        unsigned long long bal[4];
        bool m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = it + j * 1024 + tid;
            m[j] = e < n && key_of(e) == k;
            bal[j] = __ballot(m[j]);
            if (lane == 0) wave_tot[j][wid] = __popcll(bal[j]);
        }
        __syncthreads();
        int before = running;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int off = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int t = wave_tot[j][w];
                off += w < wid ? t : 0;
                tot += t;
            }
            if (m[j]) {
                const int rank = before + off + __popcll(bal[j] & ((1ull << lane) - 1ull));
                const int e = it + j * 1024 + tid;
                if (mode == 0) out[e] = start + rank; else out[start + rank] = e;
            }
            before += tot;
        }
        running = before;
        __syncthreads();
    }
}

// dest[i] = goff[code & 63] + (code >> 6) - P for the per-object entries of the window list (a pseudo-pair's own windows ARE per-object rows)
__global__ void window_rows_objects_kernel(const int* __restrict__ codes, int n, const int* __restrict__ goff, int P, int* __restrict__ dest) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = codes[i];
    dest[i] = goff[c & 63] + (c >> 6) - P;
}
// rows of the CONV list (pairs that convolve their own windows + per-object entries): the same (pair, window) sits at
// first(pair, all) + its rank inside the pair's rectangle in the list of ALL X windows
__global__ void window_rows_conv_kernel(const int* __restrict__ codes_conv, int n, const int* __restrict__ incl_conv, const int* __restrict__ incl_all,
                                        const int* __restrict__ dest_all, int* __restrict__ dest_conv) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int pair = codes_conv[e] >> 6;
    const int fc = pair ? incl_conv[pair - 1] : 0, fa = pair ? incl_all[pair - 1] : 0;
    dest_conv[e] = dest_all[e - fc + fa];
}

// inclusive prefix sums of every row of an int32 [rows][n] matrix: one 1024-thread workgroup per row walks it in chunks of 4096
// (torch.cumsum over 3 x 33 k counts took 77 us: its innermost-dim scan runs on a handful of workgroups)
__global__ __launch_bounds__(1024) void scan_rows_kernel(const int* __restrict__ in, int* __restrict__ out, int n) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int* src = in + (long)blockIdx.x * n;
    int* dst = out + (long)blockIdx.x * n;
    int running = 0;
    for (int base = 0; base < n; base += 4096) {
        const int e = base + tid * 4;
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (e + j < n) ? src[e + j] : 0;
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        int s = v[3];                                     // inclusive scan of the threads' sums inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane == 63) wave_tot[wid] = s;
        __syncthreads();
        int off = running;
#pragma unroll
        for (int w = 0; w < 16; ++w) off += w < wid ? wave_tot[w] : 0;
        int total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) total += wave_tot[w];
        const int before = off + s - v[3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e + j < n) dst[e + j] = before + v[j];
        running += total;
        __syncthreads();
    }
}

// Pair lists of the pseudo-pairs (o, background of o's image), (background, o) and - with_bg - the all-background pair of every image,
// and the window codes of the background maps' 64 windows (the tail of the pair index space: P + 2 n_obj + image): a dozen
// arange / add / cat launches per step otherwise.
__global__ void pseudo_pair_tables_kernel(const int* __restrict__ obj_img, int n_obj, int n_img, int n_pairs, int with_bg, int* __restrict__ ps_sub,
                                          int* __restrict__ ps_obj, int* __restrict__ bg_codes, int* __restrict__ bg_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_obj) {
        const int bg = n_obj + obj_img[i];
        ps_sub[i] = i; ps_obj[i] = bg;
        ps_sub[n_obj + i] = bg; ps_obj[n_obj + i] = i;
    }
    if (with_bg && i < n_img) ps_sub[2 * n_obj + i] = ps_obj[2 * n_obj + i] = n_obj + i;
    if (bg_codes && i < 64 * n_img) bg_codes[i] = (n_pairs + 2 * n_obj + (i >> 6)) * 64 + (i & 63);
    if (bg_n && i == 0) *bg_n = 64 * n_img;
}

extern "C" {

int sgc_fill_zero(void* ptr, long nbytes, void* stream) {
    if (nbytes <= 0) return SGC_OK;
    if (((uintptr_t)ptr & 15) != 0) return SGC_ERR_ARG;
    const long n16 = nbytes >> 4;
    const long blocks = (n16 + 255) / 256 < 16384 ? ((n16 + 255) / 256 > 0 ? (n16 + 255) / 256 : 1) : 16384;
    SGC_LAUNCH(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)ptr, n16,
               (unsigned char*)ptr + (n16 << 4), (int)(nbytes & 15));
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_scene_tables(const int* n_per_img, const int* img_ptr, const int* goff, int n_img, int max_n, int n_pairs, int n_obj, int pid_ld,
                     const int* rel_tri, const float* dir_tri, int* sub_idx, int* obj_idx, int* step, int* image, int* directed,
                     int* raw, int* pid, int* obj_ptr, int* sub_list, int* obj_list, int* obj_img, int* step_ptr, void* stream) {
    if (n_img <= 0 || n_obj <= 0 || pid_ld < (max_n > 1 ? max_n : 1)) return SGC_ERR_ARG;
    if ((rel_tri == nullptr) != (dir_tri == nullptr)) return SGC_ERR_ARG;
    if (rel_tri && (!directed || !raw)) return SGC_ERR_ARG;
    SceneParams s{n_per_img, img_ptr, goff, n_img, max_n, n_pairs, n_obj, pid_ld, rel_tri, dir_tri, sub_idx, obj_idx, step, image,
                  directed, raw, pid, sub_list, obj_list, obj_ptr, obj_img, step_ptr};
    const int T = max_n > 1 ? max_n * (max_n - 1) : 0;
    const long work = (long)n_obj * pid_ld > T + 1 ? (long)n_obj * pid_ld : T + 1;
    SGC_LAUNCH(scene_objects_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s, T);
    SGC_CHECK_LAUNCH();
    if (n_pairs > 0) {
        SGC_LAUNCH(scene_pairs_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, s);
        SGC_CHECK_LAUNCH();
    }
    return SGC_OK;
}

int sgc_loss_coefficients(const int* step_ptr, int n_steps, const int* directed, const float* class_weight, int ng, int np, int hier,
                          double lambda_connectivity, double lambda_not_connected, int* tgt, float* coef_a, float* coef_b, float* coef_c,
                          float* conn_y, void* stream) {
    if (n_steps <= 0) return SGC_OK;
    SGC_LAUNCH(loss_coefficients_kernel, dim3((n_steps + 63) / 64), dim3(64), 0, (hipStream_t)stream, step_ptr, n_steps, directed,
               class_weight, ng, np, hier, lambda_connectivity, lambda_not_connected, tgt, coef_a, coef_b, coef_c, conn_y);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_connectivity_stats(const float* conn, const int* directed, const int* raw, const unsigned char* included, int n_pairs,
                           unsigned long long* out5, void* stream) {
    if (hipMemsetAsync(out5, 0, 5 * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) return SGC_ERR_LAUNCH;
    if (n_pairs <= 0) return SGC_OK;
    const int blocks = (n_pairs + 255) / 256 < 256 ? (n_pairs + 255) / 256 : 256;
    SGC_LAUNCH(connectivity_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, conn, directed, raw, included, n_pairs, out5);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_label_vectors(const float* fc2_weight, int ld, int col0, const long* cats, const float* super_multihot, int n_obj, int C, int S,
                      float* lsub, float* lobj, void* stream) {
    if (n_obj <= 0) return SGC_OK;
    if (super_multihot == nullptr) S = 0;
    SGC_LAUNCH(label_vectors_kernel, dim3(n_obj), dim3(512), 0, (hipStream_t)stream, fc2_weight, ld, col0, cats, super_multihot, C, S, lsub,
               lobj);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_label_grads(const float* dlsub, const float* dlobj, const long* cats_sub, const long* cats_obj, const float* mh_sub,
                    const float* mh_obj, int n_obj, int C, int S, float* grad_fc2_weight, int ld, int col0, void* stream) {
    if (mh_sub == nullptr || mh_obj == nullptr) S = 0;
    const int cols = 2 * C + 2 * S;
    if (cols <= 0) return SGC_OK;
    SGC_LAUNCH(label_grads_kernel, dim3(cols), dim3(512), 0, (hipStream_t)stream, dlsub, dlobj, cats_sub, cats_obj, mh_sub, mh_obj, n_obj,
               C, S, grad_fc2_weight, ld, col0);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_slab_sum_ld(const float* in, float* out, int rows, int cols, long ld_out, int slabs, void* stream) {
    if (rows <= 0 || cols <= 0) return SGC_OK;
    const long n = (long)rows * cols;
    const long blocks = (n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536;
    SGC_LAUNCH(slab_sum_ld_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols, ld_out, slabs);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_bucket_place(const int* codes, int n, const int* sub_idx, const int* obj_img, int img_key, int n_keys, const int* base, int* out,
                     int* seg, int mode, void* stream) {
    if (n_keys <= 0 || (mode != 0 && mode != 1) || (mode == 0 && base == nullptr) || (mode == 1 && seg == nullptr)) return SGC_ERR_ARG;
    if (img_key && (sub_idx == nullptr || obj_img == nullptr)) return SGC_ERR_ARG;
    SGC_LAUNCH(bucket_place_kernel, dim3(n_keys), dim3(1024), 0, (hipStream_t)stream, codes, n < 0 ? 0 : n, sub_idx, obj_img, img_key, n_keys, base,
               out, seg, mode);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// the same result through the two-level kernels; scratch: int[n_keys * S] with S = ceil(n / 8192) segments (sgc_bucket_place_scratch_ints)
long sgc_bucket_place_scratch_ints(int n, int n_keys) { return (long)n_keys * (((n > 0 ? n : 1) + BP_SEG - 1) / BP_SEG); }
int sgc_bucket_place_seg(const int* codes, int n, const int* sub_idx, const int* obj_img, int img_key, int n_keys, const int* base, int* out,
                         int* seg, int mode, int* scratch, long scratch_ints, void* stream) {
    if (n_keys <= 0 || (mode != 0 && mode != 1) || (mode == 0 && base == nullptr) || (mode == 1 && seg == nullptr)) return SGC_ERR_ARG;
    if (img_key && (sub_idx == nullptr || obj_img == nullptr)) return SGC_ERR_ARG;
    if (n <= 0) {
        if (mode == 1) SGC_LAUNCH(bucket_starts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scratch, n_keys, 0, 0, seg);
        SGC_CHECK_LAUNCH();
        return SGC_OK;
    }
    const int S = (n + BP_SEG - 1) / BP_SEG;
    if (scratch == nullptr || scratch_ints < (long)n_keys * S) return SGC_ERR_ARG;
    SGC_LAUNCH(bucket_count_kernel, dim3(n_keys, S), dim3(1024), 0, (hipStream_t)stream, codes, n, sub_idx, obj_img, img_key, S, scratch);
    if (mode == 1) SGC_LAUNCH(bucket_starts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scratch, n_keys, S, n, seg);
    SGC_LAUNCH(bucket_place2_kernel, dim3(n_keys, S), dim3(1024), 0, (hipStream_t)stream, codes, n, sub_idx, obj_img, img_key, S, scratch,
               mode == 1 ? seg : base, out, mode);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_window_rows_objects(const int* codes, int n, const int* goff, int n_pairs, int* dest, void* stream) {
    if (n <= 0) return SGC_OK;
    SGC_LAUNCH(window_rows_objects_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, codes, n, goff, n_pairs, dest);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_window_rows_conv(const int* codes_conv, int n, const int* incl_conv, const int* incl_all, const int* dest_all, int* dest_conv,
                         void* stream) {
    if (n <= 0) return SGC_OK;
    SGC_LAUNCH(window_rows_conv_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, codes_conv, n, incl_conv, incl_all, dest_all,
               dest_conv);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_scan_rows(const int* in, int* out, int rows, int n, void* stream) {
    if (rows <= 0 || n <= 0) return SGC_OK;
    SGC_LAUNCH(scan_rows_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, in, out, n);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_pseudo_pair_tables(const int* obj_img, int n_obj, int n_img, int n_pairs, int with_bg, int* ps_sub, int* ps_obj, int* bg_codes,
                           int* bg_n, void* stream) {
    const int n = n_obj > 64 * n_img ? n_obj : 64 * n_img;
    if (n <= 0) return SGC_OK;
    SGC_LAUNCH(pseudo_pair_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obj_img, n_obj, n_img, n_pairs, with_bg,
               ps_sub, ps_obj, bg_codes, bg_n);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
