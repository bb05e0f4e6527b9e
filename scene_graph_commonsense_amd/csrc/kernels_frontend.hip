// SGDET / SGCLS object front-end: DETR decoder outputs -> the object list that feeds the pair path.
// Reference: evaluate.py:311-366 (= :545-589) and utils.py:58-74,377-425.  All three kernels are latency-bound bookkeeping on
// <= 100 queries x <= ~600 classes per image: one wavefront per query / one workgroup per image, no host round trips.
#include "common.h"
#include <math.h>

// --------------------------------------------------------------------------------------------------------------------
// K_F1  one wavefront per (image, query): softmax over the C1 logits (evaluate.py:311), arg-max -> has-object test (:312),
// top-k probabilities and classes (:313-316), DETR (alphabetical) -> dataset (frequency) class index (:320-322), box
// cxcywh in [0,1] -> (x0,x1,y0,y1) on the feature grid with clamp (:326-332).
// cand_cat [B][Q][k]: mapped class, or -1 when the query has no object or the mapped class == num_classes (:323,340-344).
constexpr int F1_MAX_PER_LANE = 16;           // C1 <= 1024
__global__ __launch_bounds__(256) void detr_candidates_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                              const int* __restrict__ alp2fre, int n_query, int C1,
                                                              int num_classes, int topk, float feature_size,
                                                              int* __restrict__ cand_cat, float* __restrict__ cand_conf,
                                                              float* __restrict__ cand_box) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= n_query) return;
    const float* x = logits + (long)qi * C1;
    float v[F1_MAX_PER_LANE];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < F1_MAX_PER_LANE; ++t) {
        const int c = lane + 64 * t;
        v[t] = c < C1 ? x[c] : -INFINITY;
        m = fmaxf(m, v[t]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < F1_MAX_PER_LANE; ++t) {
        const int c = lane + 64 * t;
        v[t] = c < C1 ? expf(v[t] - m) : -1.f;       // un-normalised probability; -1 marks padding / already selected
        if (c < C1) s += v[t];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    bool has_object = false;
    for (int r = 0; r < topk; ++r) {
        // largest probability; equal values resolve to the lower class index (torch.argmax: first maximum)
        float bv = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < F1_MAX_PER_LANE; ++t) {
            const float p = v[t] >= 0.f ? v[t] / s : -1.f;
            const int c = lane + 64 * t;
            if (p > bv || (p == bv && p >= 0.f && c < bi)) { bv = p; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (r == 0) has_object = bi < num_classes;
#pragma unroll
        for (int t = 0; t < F1_MAX_PER_LANE; ++t)
            if (lane + 64 * t == bi) v[t] = -1.f;
        if (lane == 0) {
            const int mapped = alp2fre[bi];
            cand_cat[(long)qi * topk + r] = (has_object && mapped != num_classes) ? mapped : -1;
            cand_conf[(long)qi * topk + r] = bv;
        }
    }
    if (lane == 0) {
        const float cx = boxes[(long)qi * 4], cy = boxes[(long)qi * 4 + 1], w = boxes[(long)qi * 4 + 2], h = boxes[(long)qi * 4 + 3];
        const float x0 = fminf(fmaxf(cx - w / 2, 0.f), 1.f), x1 = fminf(fmaxf(cx + w / 2, 0.f), 1.f);
        const float y0 = fminf(fmaxf(cy - h / 2, 0.f), 1.f), y1 = fminf(fmaxf(cy + h / 2, 0.f), 1.f);
        cand_box[(long)qi * 4] = x0 * feature_size; cand_box[(long)qi * 4 + 1] = x1 * feature_size;
        cand_box[(long)qi * 4 + 2] = y0 * feature_size; cand_box[(long)qi * 4 + 3] = y1 * feature_size;
    }
}

// --------------------------------------------------------------------------------------------------------------------
// K_F2  one workgroup per image: per-class greedy NMS (evaluate.py:347-366; torchvision.ops.nms 0.15.2 semantics).  The
// S = Q*k candidate slots are sorted by (class ascending [torch.unique order], score descending, slot ascending [stable
// sort]) with a bitonic network on 64-bit keys, then suppressed greedily inside each class segment.  Output: the kept slot
// indices in exactly the order the reference concatenates them.
constexpr int NMS_MAX = 512;                 // candidate slots per image (100 queries x up to 5 categories)
__global__ __launch_bounds__(NMS_MAX) void nms_per_class_kernel(const int* __restrict__ cand_cat, const float* __restrict__ cand_conf,
                                                                const float* __restrict__ cand_box, int Q, int topk,
                                                                double iou_threshold, int* __restrict__ out_slot,
                                                                int* __restrict__ out_count) {
    __shared__ unsigned long long key[NMS_MAX];
    __shared__ float bx[NMS_MAX][4];
    __shared__ int cls[NMS_MAX];
    __shared__ unsigned char supp[NMS_MAX];
    __shared__ int n_valid, n_keep;
    const int b = blockIdx.x, t = threadIdx.x, S = Q * topk;
    if (t == 0) { n_valid = 0; n_keep = 0; }
    unsigned long long k = ~0ull;
    if (t < S) {
        const int c = cand_cat[(long)b * S + t];
        if (c >= 0) {
            const unsigned sb = __float_as_uint(cand_conf[(long)b * S + t]);       // probabilities are >= 0: bit order = value order
            k = ((unsigned long long)c << 42) | ((unsigned long long)(0xffffffffu - sb) << 10) | (unsigned long long)t;
        }
    }
    key[t] = k;
    __syncthreads();
    for (int size = 2; size <= NMS_MAX; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int p = t ^ stride;
            if (p > t) {
                const unsigned long long a = key[t], c2 = key[p];
                const bool up = (t & size) == 0;
                if ((a > c2) == up) { key[t] = c2; key[p] = a; }
            }
            __syncthreads();
        }
    const unsigned long long mine = key[t];
    const int slot = (int)(mine & 0x3ff);
    const bool valid = mine != ~0ull;
    if (valid) {
        atomicAdd(&n_valid, 1);
        cls[t] = (int)(mine >> 42);
        const float* bp = cand_box + ((long)b * Q + slot / topk) * 4;       // (x0,x1,y0,y1); every category of a query shares its box
        bx[t][0] = bp[0]; bx[t][1] = bp[2]; bx[t][2] = bp[1]; bx[t][3] = bp[3];     // -> (x1,y1,x2,y2) as handed to nms (:349)
    }
    supp[t] = 0;
    __syncthreads();
    const int n = n_valid;
    for (int i = 0; i < n; ++i) {
        if (!supp[i] && t > i && t < n && !supp[t] && cls[t] == cls[i]) {
            const float ix1 = bx[i][0], iy1 = bx[i][1], ix2 = bx[i][2], iy2 = bx[i][3];
            const float iarea = (ix2 - ix1) * (iy2 - iy1);
            const float jarea = (bx[t][2] - bx[t][0]) * (bx[t][3] - bx[t][1]);
            const float w = fmaxf(0.f, fminf(ix2, bx[t][2]) - fmaxf(ix1, bx[t][0]));
            const float h = fmaxf(0.f, fminf(iy2, bx[t][3]) - fmaxf(iy1, bx[t][1]));
            const float inter = w * h;
            const float ovr = inter / (iarea + jarea - inter);
            if ((double)ovr > iou_threshold) supp[t] = 1;       // float IoU against the double threshold, as torchvision compares
        }
        __syncthreads();
    }
    // ordered compaction of the survivors
    if (t == 0) {
        int c = 0;
        for (int i = 0; i < n; ++i)
            if (!supp[i]) out_slot[(long)b * S + c++] = (int)(key[i] & 0x3ff);
        for (int i = c; i < S; ++i) out_slot[(long)b * S + i] = -1;
        out_count[b] = c;
    }
}

// --------------------------------------------------------------------------------------------------------------------
// K_F3  SGCLS label matching (utils.py:377-425): for every ground-truth box the two predicted boxes with the largest grid IoU
// (utils.py:58-74: rasterised masks with int() truncation; value = float(intersection) / float(union) in double, stored as
// f32).  Equal IoUs resolve to the lower prediction index (torch.topk leaves that order unspecified).
__global__ __launch_bounds__(256) void match_top2_kernel(const float* __restrict__ pred_box, const int* __restrict__ pred_ptr,
                                                         const float* __restrict__ tgt_box, const int* __restrict__ tgt_ptr,
                                                         int feature_size, int* __restrict__ top_idx, float* __restrict__ top_iou) {
    const int b = blockIdx.y;
    const int t0 = tgt_ptr[b], nt = tgt_ptr[b + 1] - t0, p0 = pred_ptr[b], np = pred_ptr[b + 1] - p0;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nt) return;
    auto clip = [&](float f) { int v = (int)f; return v < 0 ? 0 : (v > feature_size ? feature_size : v); };
    const float* bt = tgt_box + (long)(t0 + k) * 4;
    const int ta0 = clip(bt[0]), ta1 = clip(bt[1]), tb0 = clip(bt[2]), tb1 = clip(bt[3]);
    const int tarea = max(ta1 - ta0, 0) * max(tb1 - tb0, 0);
    float v0 = -1.f, v1 = -1.f;
    int i0 = -1, i1 = -1;
    for (int j = 0; j < np; ++j) {
        const float* bp = pred_box + (long)(p0 + j) * 4;
        const int a0 = clip(bp[0]), a1 = clip(bp[1]), b0 = clip(bp[2]), b1 = clip(bp[3]);
        const int parea = max(a1 - a0, 0) * max(b1 - b0, 0);
        int inter = 0;
        if (parea > 0 && tarea > 0) inter = max(min(a1, ta1) - max(a0, ta0), 0) * max(min(b1, tb1) - max(b0, tb0), 0);
        const int uni = parea + tarea - inter;
        const float v = uni == 0 ? 0.f : (float)((double)inter / (double)uni);
        if (v > v0) { v1 = v0; i1 = i0; v0 = v; i0 = j; }
        else if (v > v1) { v1 = v; i1 = j; }
    }
    top_idx[(long)(t0 + k) * 2] = i0; top_idx[(long)(t0 + k) * 2 + 1] = i1;
    top_iou[(long)(t0 + k) * 2] = v0; top_iou[(long)(t0 + k) * 2 + 1] = v1;
}

extern "C" {

int sgc_detr_candidates(const float* logits, const float* boxes, const int* alp2fre, int n_img, int n_query, int C1, int num_classes,
                        int topk, float feature_size, int* cand_cat, float* cand_conf, float* cand_box, void* stream) {
    if (C1 < 2 || C1 > 64 * F1_MAX_PER_LANE || topk < 1 || topk > C1 || num_classes > C1) return SGC_ERR_ARG;
    const int nq = n_img * n_query;
    if (nq <= 0) return SGC_OK;
    SGC_LAUNCH(detr_candidates_kernel, dim3((nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, boxes, alp2fre, nq, C1,
               num_classes, topk, feature_size, cand_cat, cand_conf, cand_box);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_nms_per_class(const int* cand_cat, const float* cand_conf, const float* cand_box, int n_img, int n_query, int topk,
                      double iou_threshold, int* out_slot, int* out_count, void* stream) {
    if (n_query * topk > NMS_MAX || n_query < 1 || topk < 1) return SGC_ERR_ARG;
    if (n_img <= 0) return SGC_OK;
    SGC_LAUNCH(nms_per_class_kernel, dim3(n_img), dim3(NMS_MAX), 0, (hipStream_t)stream, cand_cat, cand_conf, cand_box, n_query, topk,
               iou_threshold, out_slot, out_count);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_match_boxes_top2(const float* pred_box, const int* pred_ptr, const float* tgt_box, const int* tgt_ptr, int n_img, int max_tgt,
                         int feature_size, int* top_idx, float* top_iou, void* stream) {
    if (feature_size < 1) return SGC_ERR_ARG;
    if (n_img <= 0 || max_tgt <= 0) return SGC_OK;
    SGC_LAUNCH(match_top2_kernel, dim3((max_tgt + 255) / 256, n_img), dim3(256), 0, (hipStream_t)stream, pred_box, pred_ptr, tgt_box,
               tgt_ptr, feature_size, top_idx, top_iou);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
