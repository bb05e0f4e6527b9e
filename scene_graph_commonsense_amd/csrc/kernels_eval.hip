// Evaluation-side kernels: overlap filter (K10) and per-image stable top-K ranking (K9).
#include "common.h"

// iou_mask of testing(): reference train_test.py:403-408 computes sum(mask_g | mask_e) / sum(mask_g & mask_e),
// maps inf -> 0 and keeps ratio > 0, i.e. "the two rectangles share at least one grid cell"
// (0/0 = NaN and x/0 = inf both end up False).
__global__ void overlap_filter_kernel(const int* __restrict__ bbox, const int* __restrict__ sub, const int* __restrict__ obj,
                                      unsigned char* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int* a = bbox + 4 * sub[i];
    const int* b = bbox + 4 * obj[i];
    const int w = min(a[1], b[1]) - max(a[0], b[0]);
    const int h = min(a[3], b[3]) - max(a[2], b[2]);
    const bool a_ok = a[1] > a[0] && a[3] > a[2], b_ok = b[1] > b[0] && b[3] > b[2];
    out[i] = (a_ok && b_ok && w > 0 && h > 0) ? 1 : 0;
}

__device__ __forceinline__ unsigned order_key(float f) {      // ascending uint order == ascending float order
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One workgroup per image: radix-select the K-th largest confidence (4 x 8-bit LDS histograms), gather everything
// above it plus the earliest ties (ordered compaction with wavefront ballots), bitonic-sort the <=128 survivors
// by (confidence desc, index asc).  Equivalent to a stable descending argsort truncated to K
// (reference evaluator.py:304,315-316 uses an unstable argsort; ties are resolved by append order here).
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ conf, const int* __restrict__ seg_ptr, int K,
                                                   int* __restrict__ out_idx, int* __restrict__ out_count) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_mask;
    __shared__ int s_remaining, s_cnt, s_base, wave_cnt[4];
    __shared__ unsigned long long keys[128];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = seg_ptr[img], n = seg_ptr[img + 1] - b;
    const int k = min(K, n);
    const float* c = conf + b;
    if (tid < 128) keys[tid] = ~0ull;
    if (tid == 0) { s_prefix = 0; s_mask = 0; s_remaining = k; s_cnt = 0; out_count[img] = k; }
    __syncthreads();
    if (k > 0) {
        for (int pass = 3; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, mask = s_mask;
            for (int i = tid; i < n; i += 256) {
                const unsigned key = order_key(c[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, rem = s_remaining;
                for (int bk = 255; bk >= 0; --bk) {
                    const int hv = (int)hist[bk];
                    if (cum + hv >= rem) {
                        s_prefix = prefix | ((unsigned)bk << (8 * pass));
                        s_mask = mask | (0xffu << (8 * pass));
                        s_remaining = rem - cum;
                        break;
                    }
                    cum += hv;
                }
            }
            __syncthreads();
        }
        const unsigned thr = s_prefix;
        for (int i = tid; i < n; i += 256) {
            const unsigned key = order_key(c[i]);
            if (key > thr) {
                const int pos = atomicAdd(&s_cnt, 1);
                keys[pos] = ((unsigned long long)(~key) << 32) | (unsigned)i;
            }
        }
        __syncthreads();
        if (tid == 0) s_base = s_cnt;
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 256) {
            if (s_base >= k) break;
            const int i = c0 + tid;
            const bool flag = i < n && order_key(c[i]) == thr;
            const unsigned long long bal = __ballot(flag);
            const int within = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[wv] = __popcll(bal);
            __syncthreads();
            int off = s_base + within;
            for (int w2 = 0; w2 < wv; ++w2) off += wave_cnt[w2];
            if (flag && off < k) keys[off] = ((unsigned long long)(~thr) << 32) | (unsigned)i;
            __syncthreads();
            if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
        // bitonic sort of 128 keys, ascending ( = confidence descending, index ascending)
        for (int size = 2; size <= 128; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                if (tid < 64) {
                    const int lo = 2 * tid - (tid & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const unsigned long long a = keys[lo], bb = keys[hi];
                    if ((a > bb) == up) { keys[lo] = bb; keys[hi] = a; }
                }
            }
        }
        __syncthreads();
    }
    for (int j = tid; j < K; j += 256) out_idx[(long)img * K + j] = (j < k) ? (int)(keys[j] & 0xffffffffull) : -1;
}

// Commonsense filter of eval_cs / train_cs (reference evaluator.py:189-194,261-266): a candidate whose
// (subject class, predicate, object class) triplet is in the "violated" set or is not in the "aligned" set gets
// confidence -inf.  The Python-set membership tests become two bit lookups in C*R*C-bit device bitmaps.
__global__ void commonsense_filter_kernel(const long* __restrict__ scat, const long* __restrict__ pred, const long* __restrict__ ocat,
                                          float* __restrict__ conf, int n, const unsigned* __restrict__ aligned,
                                          const unsigned* __restrict__ violated, int C, int R) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long s = scat[i], r = pred[i], o = ocat[i];
    bool keep = false;
    if (s >= 0 && s < C && o >= 0 && o < C && r >= 0 && r < R) {
        const long bit = (s * R + r) * C + o;
        const bool in_yes = (aligned[bit >> 5] >> (bit & 31)) & 1u;
        const bool in_no = (violated[bit >> 5] >> (bit & 31)) & 1u;
        keep = in_yes && !in_no;
    }
    if (!keep) conf[i] = -INFINITY;
}

// train_cs (reference train_utils.py:36-50): per candidate (pair, super-category) flags "triplet not in the aligned set"
// (weak penalty) and "triplet in the violated set" (strong penalty).  cand_pred [n_pairs][n_cand] int32.
__global__ void commonsense_flags_kernel(const long* __restrict__ scat, const long* __restrict__ ocat, const int* __restrict__ cand_pred,
                                         int n_pairs, int n_cand, const unsigned* __restrict__ aligned,
                                         const unsigned* __restrict__ violated, int C, int R, float* __restrict__ weak,
                                         float* __restrict__ strong) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs * n_cand) return;
    const int pr = i / n_cand;
    const long s = scat[pr], o = ocat[pr], r = cand_pred[i];
    bool in_yes = false, in_no = false;
    if (s >= 0 && s < C && o >= 0 && o < C && r >= 0 && r < R) {
        const long bit = (s * R + r) * C + o;
        in_yes = (aligned[bit >> 5] >> (bit & 31)) & 1u;
        in_no = (violated[bit >> 5] >> (bit & 31)) & 1u;
    }
    weak[i] = in_yes ? 0.f : 1.f;
    strong[i] = in_no ? 1.f : 0.f;
}

// Recall@K hit test (reference evaluator.py:306-356, Top-3 variant :720-760): for every connected ground-truth triple, the rank
// j of the first of its image's K ranked candidates with matching subject / object labels, both grid IoUs >= the threshold AND a
// matching predicate (a candidate whose labels and boxes match but whose predicate differs does not end the scan), or K.
// One wavefront per triple: lane l tests candidates l, l+64, ...; a ballot gives the first matching rank.
struct RecallParams {
    const long* c_scat; const long* c_ocat; const long* c_pred;      // candidates (flat); c_pred [n][n_pred]
    const float* c_sbox; const float* c_obox;                         // [n][4] (x0,x1,y0,y1) as stored (int() truncation applied here)
    const int* keep_pos; const int* keep_cnt;                         // [n_img][K] flat candidate positions in ranked order, [n_img]
    const int* t_row; const long* t_rel; const long* t_scat; const long* t_ocat; const float* t_sbox; const float* t_obox;
    const unsigned char* equiv; int n_equiv;                          // label equivalence table (SGDET/SGCLS, utils.py:355-373) or NULL
    int n_t, K, n_pred, F; double iou_thresh;
    int* hit;
};

__device__ __forceinline__ int slice_clip(float v, int F) {           // int(v) then Python slice clipping on an axis of length F
    int i = (int)v;
    return i < 0 ? max(i + F, 0) : min(i, F);
}
__device__ __forceinline__ bool grid_iou_ok(const float* a, const float* b, int F, double thr) {
    const int ax0 = slice_clip(a[0], F), ax1 = slice_clip(a[1], F), ay0 = slice_clip(a[2], F), ay1 = slice_clip(a[3], F);
    const int bx0 = slice_clip(b[0], F), bx1 = slice_clip(b[1], F), by0 = slice_clip(b[2], F), by1 = slice_clip(b[3], F);
    const int aa = max(ax1 - ax0, 0) * max(ay1 - ay0, 0), ab = max(bx1 - bx0, 0) * max(by1 - by0, 0);
    int inter = max(min(ax1, bx1) - max(ax0, bx0), 0) * max(min(ay1, by1) - max(ay0, by0), 0);
    if (aa <= 0 || ab <= 0) inter = 0;
    const int uni = aa + ab - inter;
    const double iou = uni > 0 ? (double)inter / (double)uni : 0.0;
    return iou >= thr;
}

__global__ __launch_bounds__(256) void recall_hits_kernel(const RecallParams p) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= p.n_t) return;
    const int row = p.t_row[t], cnt = p.keep_cnt[row];
    const long rel = p.t_rel[t], sc = p.t_scat[t], oc = p.t_ocat[t];
    int hit = p.K;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        bool ok = false;
        if (j < cnt) {
            const long c = p.keep_pos[(long)row * p.K + j];
            const long cs = p.c_scat[c], co = p.c_ocat[c];
            bool label = (cs == sc) && (co == oc);
            if (p.equiv && !label) {
                const bool in = sc >= 0 && sc < p.n_equiv && oc >= 0 && oc < p.n_equiv && cs >= 0 && cs < p.n_equiv && co >= 0 && co < p.n_equiv;
                label = in && p.equiv[sc * p.n_equiv + cs] && p.equiv[oc * p.n_equiv + co];
            }
            if (label) {
                bool pred = false;
                for (int k = 0; k < p.n_pred; ++k) pred |= p.c_pred[c * p.n_pred + k] == rel;
                ok = pred && grid_iou_ok(p.t_sbox + 4L * t, p.c_sbox + 4 * c, p.F, p.iou_thresh) &&
                     grid_iou_ok(p.t_obox + 4L * t, p.c_obox + 4 * c, p.F, p.iou_thresh);
            }
        }
        const unsigned long long m = __ballot(ok);
        if (m) { hit = j0 + __ffsll((long long)m) - 1; break; }
    }
    if (lane == 0) p.hit[t] = hit;
}

extern "C" {

int sgc_recall_hits(const long* c_scat, const long* c_ocat, const long* c_pred, int n_pred, const float* c_sbox, const float* c_obox,
                    const int* keep_pos, const int* keep_cnt, int K, const int* t_row, const long* t_rel, const long* t_scat,
                    const long* t_ocat, const float* t_sbox, const float* t_obox, int n_targets, const unsigned char* equiv, int n_equiv,
                    int feature_size, double iou_thresh, int* hit, void* stream) {
    if (n_targets <= 0) return SGC_OK;
    if (n_pred < 1 || K < 1) return SGC_ERR_ARG;
    RecallParams p{c_scat, c_ocat, c_pred, c_sbox, c_obox, keep_pos, keep_cnt, t_row, t_rel, t_scat, t_ocat, t_sbox, t_obox, equiv, n_equiv,
                   n_targets, K, n_pred, feature_size, iou_thresh, hit};
    SGC_LAUNCH(recall_hits_kernel, dim3((n_targets + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_commonsense_flags(const long* scat, const long* ocat, const int* cand_pred, int n_pairs, int n_cand, const unsigned* aligned,
                          const unsigned* violated, int C, int R, float* weak, float* strong, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(commonsense_flags_kernel, dim3((n_pairs * n_cand + 255) / 256), dim3(256), 0, (hipStream_t)stream, scat, ocat, cand_pred,
               n_pairs, n_cand, aligned, violated, C, R, weak, strong);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_commonsense_filter(const long* scat, const long* pred, const long* ocat, float* conf, int n, const unsigned* aligned,
                           const unsigned* violated, int C, int R, void* stream) {
    if (n <= 0) return SGC_OK;
    SGC_LAUNCH(commonsense_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, scat, pred, ocat, conf, n,
               aligned, violated, C, R);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_overlap_filter(const int* bbox, const int* sub_idx, const int* obj_idx, unsigned char* out, int n_pairs, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    SGC_LAUNCH(overlap_filter_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, sub_idx, obj_idx,
                       out, n_pairs);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_topk_per_image(const float* conf, const int* seg_ptr, int n_img, int K, int* out_idx, int* out_count, void* stream) {
    if (K < 1 || K > 128) return SGC_ERR_ARG;
    if (n_img <= 0) return SGC_OK;
    SGC_LAUNCH(topk_kernel, dim3(n_img), dim3(256), 0, (hipStream_t)stream, conf, seg_ptr, K, out_idx, out_count);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
