// Backward-path kernels of the pairwise relation head (gfx950) and their C-ABI launchers.
// Gradient tensors are bf16 (range of f32; the reference's running-sum loss quirk spreads per-pair loss
// weights over >3 decades), accumulation is f32.  Layouts as in kernels_fwd.hip, plus
//   dy3_pad  [n_pair][18][18][1024]  grad wrt conv3 pre-pool output, zero border
//   dz       [n_pair*256][512]       grad wrt z, rows window-major over the 16x16 map
//   dU_pad   [n_obj][34][34][512]    grad wrt U (or V), zero border (halo for the conv2 dgrad)
#include "common.h"
#include "gemm_nt.h"
#include "gemm_tn.h"
#include "gemm_tn_sp.h"

// ------------------------------------------------------------------------------------------------ head + loss
// Per pair: dL/dlogits of the hierarchical NLL / BCE loss (reference train_utils.py:64-94,116-157 with the
// per-step weights of train_test.py:219-258 folded into the coefficients by the host), the pair's loss
// value, and dL/d(fc2 pre-activation) through the head weights and the ReLU/dropout mask.
struct HeadBwdParams {
    const float* rel; const float* sup; const float* conn; const float* p;
    const int* tgt; const float* coef_a; const float* coef_b; const float* coef_c; const float* conn_y;
    const float* W;            // [64][512] rows as in the forward head
    int n_pairs, ng, np, ns, hier; float invT1, invT2, invT3; float drop_scale;
    float* dl; float* loss; u16* dpre;
    const float* dp_extra;     // optional [n_pairs][512]: extra dL/d(hidden) (supervised-contrastive term)
    const float* cs_coef;      // optional [n_pairs][3 or 1]: commonsense penalty coefficient per candidate (train_cs)
    const float* cand_conf; const int* cand_pred;   // forward candidates (max log-prob / logit and argmax per segment)
    // upstream mode (g_rel != NULL): no loss here - the caller's autograd hands over dL/d(outputs) of the per-step forward()
    const float* g_rel; const float* g_sup; const float* g_conn;
};

__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(512) void head_loss_bwd_kernel(const HeadBwdParams hp) {
    // 4 or 8 wavefronts per workgroup (one weight image in LDS per CU); a pair's work is a chain of small dependent loads and 54 LDS rows,
    // so the wavefronts in flight are what hides it (round 5: 8 when there are pairs for them, 0.18 -> see profiles/r05_small_kernels.txt)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Wl = reinterpret_cast<float*>(smem);          // [64][512]
    float* dls = Wl + 64 * 512;                          // [waves][64]
    const int nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < 64 * 512; i += blockDim.x) Wl[i] = hp.W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int R = hp.ng + hp.np + hp.ns;
    float* dlw = dls + w * 64;
    for (int pr = blockIdx.x * nw + w; pr < hp.n_pairs; pr += gridDim.x * nw) {
        const bool up = hp.g_rel != nullptr;
        const int t = up ? -1 : hp.tgt[pr];
        const float a = up ? 0.f : hp.coef_a[pr], b = up ? 0.f : hp.coef_b[pr], cc = up ? 0.f : hp.coef_c[pr], y = up ? 0.f : hp.conn_y[pr];
        const float cn = up ? 0.f : hp.conn[pr];
        float dl = 0.f, lossv = 0.f;
        if (cc != 0.f) lossv += cc * (y > 0.5f ? softplusf(-cn) : softplusf(cn));
        const float sig = 1.f / (1.f + expf(-cn));
        if (hp.g_rel) {
            // backward of model.py:176-184 for arbitrary upstream gradients: rel_k = log_softmax(l_k / T_k) + sup_k,
            // sup = log_softmax(s).  d l_r = (G_r - softmax_r * sum_seg G) / T_k;  d s_j = Gs_j - exp(sup_j) * sum Gs with
            // Gs_k = G_sup_k + sum_{r in seg k} G_r;  flat head: the relation output IS the logit.
            const float gr = lane < R ? hp.g_rel[(long)pr * R + lane] : 0.f;
            if (hp.hier) {
                const int seg = lane < hp.ng ? 0 : (lane < hp.ng + hp.np ? 1 : 2);
                float segsum[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) segsum[k] = wave_sum_f((lane < R && seg == k) ? gr : 0.f);
                float gs[3], gtot = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) { gs[k] = segsum[k] + (hp.g_sup ? hp.g_sup[(long)pr * 3 + k] : 0.f); gtot += gs[k]; }
                if (lane < R) {
                    const float invT = seg == 0 ? hp.invT1 : (seg == 1 ? hp.invT2 : hp.invT3);
                    dl = invT * (gr - expf(hp.rel[(long)pr * R + lane] - hp.sup[(long)pr * 3 + seg]) * segsum[seg]);
                } else if (lane < R + 3) {
                    const int j = lane - R;
                    dl = (j == 0 ? gs[0] : (j == 1 ? gs[1] : gs[2])) - expf(hp.sup[(long)pr * 3 + j]) * gtot;
                } else if (lane == R + 3) {
                    dl = hp.g_conn ? hp.g_conn[pr] : 0.f;
                }
            } else {
                if (lane < R) dl = gr;
                else if (lane == R) dl = hp.g_conn ? hp.g_conn[pr] : 0.f;
            }
            lossv = 0.f;
        } else if (hp.hier) {
            if (lane == R + 3) dl = cc * (sig - y);
            if (t >= 0) {
                const int st = t < hp.ng ? 0 : (t < hp.ng + hp.np ? 1 : 2);
                const float invT = st == 0 ? hp.invT1 : (st == 1 ? hp.invT2 : hp.invT3);
                const float sst = hp.sup[(long)pr * 3 + st];
                if (lane < R) {
                    const int seg = lane < hp.ng ? 0 : (lane < hp.ng + hp.np ? 1 : 2);
                    if (seg == st) dl = b * invT * (expf(hp.rel[(long)pr * R + lane] - sst) - (lane == t ? 1.f : 0.f));
                } else if (lane < R + 3) {
                    dl = (a + b) * (expf(hp.sup[(long)pr * 3 + lane - R]) - ((lane - R) == st ? 1.f : 0.f));
                }
                lossv += -a * sst - b * hp.rel[(long)pr * R + t];
            }
            if (hp.cs_coef) {        // penalty kappa_s * max_r softmax(x_s / T_s): d/dx_r = kappa * pmax * ([r == argmax] - p_r) / T
                const int seg = lane < hp.ng ? 0 : (lane < hp.ng + hp.np ? 1 : 2);
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const float kap = hp.cs_coef[(long)pr * 3 + s2];
                    if (kap == 0.f) continue;
                    const float ss = hp.sup[(long)pr * 3 + s2];
                    const float pmax = expf(hp.cand_conf[(long)pr * 3 + s2] - ss);
                    const float invT = s2 == 0 ? hp.invT1 : (s2 == 1 ? hp.invT2 : hp.invT3);
                    if (lane < R && seg == s2)
                        dl += kap * invT * pmax * ((lane == hp.cand_pred[(long)pr * 3 + s2] ? 1.f : 0.f) - expf(hp.rel[(long)pr * R + lane] - ss));
                    lossv += kap * pmax;
                }
            }
        } else {
            if (lane == R) dl = cc * (sig - y);
            if (t >= 0) {                                   // weighted cross-entropy on raw logits
                const float x = lane < R ? hp.rel[(long)pr * R + lane] : -INFINITY;
                const float m = wave_max_f(x);
                const float e = lane < R ? expf(x - m) : 0.f;
                const float s = wave_sum_f(e);
                if (lane < R) dl = b * (e / s - (lane == t ? 1.f : 0.f));
                const float xt = hp.rel[(long)pr * R + t];
                lossv += -b * (xt - m - logf(s));
            }
            if (hp.cs_coef && hp.cs_coef[pr] != 0.f) {
                const float kap = hp.cs_coef[pr];
                const float x = lane < R ? hp.rel[(long)pr * R + lane] : -INFINITY;
                const float m = wave_max_f(x);
                const float e = lane < R ? expf(x - m) : 0.f;
                const float s = wave_sum_f(e);
                const float pmax = 1.f / s;                       // exp(max - m) / s with max == m
                if (lane < R) dl += kap * pmax * ((lane == hp.cand_pred[pr] ? 1.f : 0.f) - e / s);
                lossv += kap * pmax;
            }
        }
        hp.dl[(long)pr * 64 + lane] = dl;
        if (lane == 0 && hp.loss) hp.loss[pr] = lossv;
        dlw[lane] = dl;
        __builtin_amdgcn_wave_barrier();
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const int nrow = hp.hier ? R + 4 : R + 1;
        for (int r = 0; r < nrow; ++r) {
            const float d = dlw[r];
            if (d != 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(d, Wl[r * 512 + lane + 64 * j], acc[j]);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            const float pv = hp.p[(long)pr * 512 + k];
            const float g = acc[j] + (hp.dp_extra ? hp.dp_extra[(long)pr * 512 + k] : 0.f);
            hp.dpre[(long)pr * 512 + k] = f32_to_bf16_bits(pv > 0.f ? g * hp.drop_scale : 0.f);
        }
    }
}

// dW[r][k] partial = sum over a chunk of pairs dl[p][r] * p[p][k];  column 512 holds the bias gradient.
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dl, const float* __restrict__ p,
                                                         float* __restrict__ part, int n_pairs, int chunk) {
    __shared__ float dls[64];
    const int k0 = threadIdx.x, k1 = threadIdx.x + 256;
    float a0[64], a1[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    float ab = 0.f;
    const int pb = blockIdx.x * chunk;
    const int pe = min(n_pairs, pb + chunk);
    for (int pr = pb; pr < pe; ++pr) {
        __syncthreads();
        if (threadIdx.x < 64) dls[threadIdx.x] = dl[(long)pr * 64 + threadIdx.x];
        __syncthreads();
        const float v0 = p[(long)pr * 512 + k0], v1 = p[(long)pr * 512 + k1];
        if (threadIdx.x < 64) ab += dls[threadIdx.x];
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const float d = dls[r];
            a0[r] = fmaf(d, v0, a0[r]);
            a1[r] = fmaf(d, v1, a1[r]);
        }
    }
    float* out = part + (long)blockIdx.x * 64 * 513;
#pragma unroll
    for (int r = 0; r < 64; ++r) { out[r * 513 + k0] = a0[r]; out[r * 513 + k1] = a1[r]; }
    if (threadIdx.x < 64) out[threadIdx.x * 513 + 512] = ab;
}

// ------------------------------------------------------------------------------------------------ supervised contrastive
// SupConLossHierar (reference sup_contrast/losses.py:85-181, contrast_mode 'all') on F [2M][512] f32: rows 0..M-1 are the
// hidden vectors of the connected pairs, rows M..2M-1 those of the augmented view; label of row i is labels[i mod M].
// Positives: same label, i != j.  Denominator: every j != i whose parent super-category (label < 15 / < 26 / else, the
// reference's hard-coded boundaries) equals that of i.  One workgroup per anchor row: dots against all rows, row max,
// masked exp-sum, loss_i = -mean_pos(log_prob), and the row of G = dL/d(logits/temperature... before the max shift).
__global__ __launch_bounds__(256) void supcon_rows_kernel(const float* __restrict__ Fm, const int* __restrict__ labels, int M,
                                                          float inv_temp, float* __restrict__ G, float* __restrict__ loss_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* fi = reinterpret_cast<float*>(smem);          // [512]
    float* row = fi + 512;                               // [2M]
    __shared__ float red[256];
    const int n = 2 * M, i = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < 512; k += 256) fi[k] = Fm[(long)i * 512 + k];
    __syncthreads();
    const int li = labels[i % M];
    const int pi = (li >= 15) + (li >= 26);
    float lmax = -INFINITY;
    for (int j = tid; j < n; j += 256) {
        const float* fj = Fm + (long)j * 512;
        float d = 0.f;
        for (int k = 0; k < 512; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(fj + k);
            d = fmaf(fi[k], v.x, d); d = fmaf(fi[k + 1], v.y, d); d = fmaf(fi[k + 2], v.z, d); d = fmaf(fi[k + 3], v.w, d);
        }
        d *= inv_temp;
        row[j] = d;
        lmax = fmaxf(lmax, d);
    }
    red[tid] = lmax;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) { if (tid < sft) red[tid] = fmaxf(red[tid], red[tid + sft]); __syncthreads(); }
    const float m = red[0];
    __syncthreads();
    float s_exp = 0.f, s_mask = 0.f, s_pos = 0.f;
    for (int j = tid; j < n; j += 256) {
        if (j == i) continue;
        const int lj = labels[j % M];
        const int pj = (lj >= 15) + (lj >= 26);
        const float l = row[j] - m;
        if (pj == pi) s_exp += expf(l);
        if (lj == li) { s_mask += 1.f; s_pos += l; }
    }
    __shared__ float r2[3][256];
    r2[0][tid] = s_exp; r2[1][tid] = s_mask; r2[2][tid] = s_pos;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if (tid < sft) { r2[0][tid] += r2[0][tid + sft]; r2[1][tid] += r2[1][tid + sft]; r2[2][tid] += r2[2][tid + sft]; }
        __syncthreads();
    }
    const float S = r2[0][0] + 1e-7f, nm = r2[1][0], npos = r2[1][0] + 1e-7f, sp = r2[2][0];
    if (tid == 0) loss_rows[i] = -(sp - nm * logf(S)) / npos;
    const float wsum = nm / npos, invn = 1.f / (float)n;
    for (int j = tid; j < n; j += 256) {
        float g = 0.f;
        if (j != i) {
            const int lj = labels[j % M];
            const int pj = (lj >= 15) + (lj >= 26);
            const float e = (pj == pi) ? expf(row[j] - m) / S : 0.f;
            g = -invn * ((lj == li ? 1.f / npos : 0.f) - wsum * e);
        }
        G[(long)i * n + j] = g;
    }
}

// dF[i][:] = scale * sum_j (G[i][j] + G[j][i]) * F[j][:]   (logits = F F^T / temperature is symmetric in its two factors)
__global__ __launch_bounds__(256) void supcon_dfeat_kernel(const float* __restrict__ Fm, const float* __restrict__ G, int n, float scale,
                                                           float* __restrict__ dF) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* c = reinterpret_cast<float*>(smem);           // [n]
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < n; j += 256) c[j] = G[(long)i * n + j] + G[(long)j * n + i];
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
    for (int j = 0; j < n; ++j) {
        const float cj = c[j];
        a0 = fmaf(cj, Fm[(long)j * 512 + tid], a0);
        a1 = fmaf(cj, Fm[(long)j * 512 + 256 + tid], a1);
    }
    dF[(long)i * 512 + tid] = a0 * scale;
    dF[(long)i * 512 + 256 + tid] = a1 * scale;
}

// ------------------------------------------------------------------------------------------------ small reductions
// out[n] (+)= sum_s in[s][n]
__global__ void slab_sum_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int slabs, int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = accumulate ? out[i] : 0.f;
        int k = 0;
        for (; k + 8 <= slabs; k += 8) {          // eight loads in flight, added in slab order (same sums as the plain loop)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(long)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < slabs; ++k) s += in[(long)k * n + i];
        out[i] = s;
    }
}

// SGD with momentum and weight decay in one pass (torch.optim.SGD semantics, dampening 0, no Nesterov; train_test.py:99-100):
//   g' = g + wd * w;  buf = first ? g' : momentum * buf + g';  w -= lr * buf.      torch's foreach path makes three passes.
__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ buf,
                                                           long n, float lr, float momentum, float wd, int first) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 wv = reinterpret_cast<float4*>(w)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
        float* wp = reinterpret_cast<float*>(&wv);
        const float* gp = reinterpret_cast<const float*>(&gv);
        float* bp = reinterpret_cast<float*>(&bv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] + wd * wp[k];
            bp[k] = first ? gg : momentum * bp[k] + gg;
            wp[k] = wp[k] - lr * bp[k];
        }
        reinterpret_cast<float4*>(buf)[i] = bv;
        reinterpret_cast<float4*>(w)[i] = wv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                 // tail (n not a multiple of 4)
        const long i = (n4 << 2) + threadIdx.x;
        const float gg = g[i] + wd * w[i];
        const float b = first ? gg : momentum * buf[i] + gg;
        buf[i] = b;
        w[i] = w[i] - lr * b;
    }
}

__global__ __launch_bounds__(256) void sgd_momentum_scalar_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                                  float* __restrict__ buf, long n, float lr, float momentum, float wd,
                                                                  int first) {       // views that are not 16-byte aligned (small tensors)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gg = g[i] + wd * w[i];
        const float b = first ? gg : momentum * buf[i] + gg;
        buf[i] = b;
        w[i] = w[i] - lr * b;
    }
}

// column sums of a 16-bit matrix: part[blockIdx.y][cols] = sum over this block's rows.
// Block = 32 column chunks (8 columns, 16 B each) x 8 row lanes; row lanes are reduced through LDS.
template <int ELEM>
__global__ __launch_bounds__(256) void colsum_kernel(const u16* __restrict__ X, float* __restrict__ part, long rows, int cols,
                                                     long rows_per_block) {
    __shared__ float red[8][32][9];
    const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cc) * 8;
    const long r0 = blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (c < cols) {
        for (long r = r0 + rl; r < r1; r += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(X + r * cols + c);
            const u16* h = reinterpret_cast<const u16*>(&v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += from_elem<ELEM>(h[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rl][cc][k] = acc[k];
    __syncthreads();
    if (rl == 0 && c < cols) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s2 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s2 += red[q][cc][k];
            part[(long)blockIdx.y * cols + c + k] = s2;
        }
    }
}

// out[o][:] = sum over pairs in list[ptr[o]..ptr[o+1]) of X[pair][:]  (label-column gradients of fc2)
__global__ __launch_bounds__(64) void segment_sum_rows_kernel(const u16* __restrict__ X, const int* __restrict__ ptr,
                                                              const int* __restrict__ list, float* __restrict__ out, int cols) {
    const int o = blockIdx.x, c = threadIdx.x * 8;
    if (c >= cols) return;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int i = ptr[o]; i < ptr[o + 1]; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(X + (long)list[i] * cols + c);
        const u16* h = reinterpret_cast<const u16*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += bf16_bits_to_f32(h[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out[(long)o * cols + c + k] = acc[k];
}

__global__ void convert_f16_bf16_kernel(const u16* __restrict__ in, u16* __restrict__ out, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const uint4 v = reinterpret_cast<const uint4*>(in)[i];
        const u16* h = reinterpret_cast<const u16*>(&v);
        uint4 o;
        u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
        for (int k = 0; k < 8; ++k) oh[k] = f32_to_bf16_bits(f16_bits_to_f32(h[k]));
        reinterpret_cast<uint4*>(out)[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------ layout transforms
// 64x64 tile transpose with type conversion: dst[base_d + j*ds_j + i] = convert(src[base_s + i*ss_i + j]), i,j < 64,
// one tile per workgroup, tiles enumerated by (blockIdx.x, blockIdx.y) with independent source/destination strides.
// Used to derive the fc1 compute copies from the f32 master ([n][c*64+w] -> f16 [n][w*1024+c] and bf16 [w*1024+c][n])
// and to bring the fc1 weight gradient back to the reference layout.  OUT: 0 f16, 1 bf16, 2 f32.
template <int OUT>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, void* __restrict__ dst, long sa_s, long sb_s,
                                                             long ss_i, long sa_d, long sb_d, long ds_j) {
    __shared__ float tile[64][65];
    const long bs = blockIdx.x * sa_s + blockIdx.y * sb_s;
    const long bd = blockIdx.x * sa_d + blockIdx.y * sb_d;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = r * 4 + w;
        tile[i][lane] = src[bs + i * ss_i + lane];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = r * 4 + w;
        const float v = tile[lane][j];
        const long o = bd + j * ds_j + lane;
        if constexpr (OUT == 0) reinterpret_cast<u16*>(dst)[o] = f32_to_f16_bits(v);
        else if constexpr (OUT == 1) reinterpret_cast<u16*>(dst)[o] = f32_to_bf16_bits(v);
        else reinterpret_cast<float*>(dst)[o] = v;
    }
}

// ------------------------------------------------------------------------------------------------ un-pool (conv3 output)
// dy3_pad[p][2py+dy+1][2px+dx+1][c] = (argmax[p][W][c] == dy*2+dx) ? dy[p][W][c] : 0 ; db3[c] += routed dy
__global__ __launch_bounds__(256) void unpool_kernel(const u16* __restrict__ dy, const unsigned char* __restrict__ am,
                                                     u16* __restrict__ dy3, float* __restrict__ dbias, long n_win) {
    const int ch = threadIdx.x & 127;                    // 128 chunks of 8 channels
    const int sub = threadIdx.x >> 7;                    // 2 windows per block iteration
    float bs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bs[k] = 0.f;
    for (long wdx = (long)blockIdx.x * 2 + sub; wdx < n_win; wdx += (long)gridDim.x * 2) {
        const long p = wdx >> 6;
        const int W = (int)(wdx & 63), py = W >> 3, px = W & 7;
        const uint4 g = *reinterpret_cast<const uint4*>(dy + wdx * 1024 + ch * 8);
        const uint2 a = *reinterpret_cast<const uint2*>(am + wdx * 1024 + ch * 8);
        const u16* gh = reinterpret_cast<const u16*>(&g);
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(&a);
        uint4 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u16* oh = reinterpret_cast<u16*>(&o[q]);
#pragma unroll
            for (int k = 0; k < 8; ++k) oh[k] = (ab[k] == q) ? gh[k] : (u16)0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (ab[k] < 4) bs[k] += bf16_bits_to_f32(gh[k]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int Y = 2 * py + (q >> 1) + 1, X = 2 * px + (q & 1) + 1;
            *reinterpret_cast<uint4*>(dy3 + ((p * 18 + Y) * 18 + X) * 1024 + ch * 8) = o[q];
        }
    }
    if (dbias) {                                         // per-block partial: dbias[blockIdx.x][1024]
        __shared__ float red[2][1024];
#pragma unroll
        for (int k = 0; k < 8; ++k) red[sub][ch * 8 + k] = bs[k];
        __syncthreads();
        for (int c = threadIdx.x; c < 1024; c += 256) dbias[(long)blockIdx.x * 1024 + c] = red[0][c] + red[1][c];
    }
}

// ------------------------------------------------------------------------------------------------ pair expansion (training)
// Forward expansion that also records which of the four positions won (0..3, 4 = none positive): the
// routing mask of relu+maxpool for the contraction below.
__global__ __launch_bounds__(256) void pair_expand_train_kernel(const u16* __restrict__ U, const u16* __restrict__ V,
                                                                const int* __restrict__ sub, const int* __restrict__ obj,
                                                                u16* __restrict__ z, u16* __restrict__ zb,
                                                                unsigned char* __restrict__ amz, long n_items) {
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        const int p = (int)(it >> 8), W = (int)(it & 255);
        const u16* up = U + ((long)sub[p] * 1024 + 4 * W) * 512 + lane * 8;
        const u16* vp = V + ((long)obj[p] * 1024 + 4 * W) * 512 + lane * 8;
        float best[8];
        unsigned arg[8];                                // 32-bit: as bytes the compiler spends ~145 sub-dword instructions per item on them
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = 0.f; arg[k] = 4; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4*>(up + q * 512);
            const uint4 b = *reinterpret_cast<const uint4*>(vp + q * 512);
            const u16* ah = reinterpret_cast<const u16*>(&a);
            const u16* bh = reinterpret_cast<const u16*>(&b);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float s = f16_bits_to_f32(ah[k]) + f16_bits_to_f32(bh[k]);
                if (s > best[k]) { best[k] = s; arg[k] = (unsigned)q; }
            }
        }
        const int Y = W >> 4, X = W & 15;
        const long zo = (((long)p * 18 + Y + 1) * 18 + X + 1) * 512 + lane * 8;
        if (z) {                                               // f16 copy: conv3 forward operand
            uint4 o;
            u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
            for (int k = 0; k < 8; ++k) oh[k] = f32_to_f16_bits(best[k]);
            *reinterpret_cast<uint4*>(z + zo) = o;
        }
        if (zb) {                                              // bf16 copy: conv3 weight-gradient operand
            uint4 o;
            u16* oh = reinterpret_cast<u16*>(&o);
#pragma unroll
            for (int k = 0; k < 8; ++k) oh[k] = f32_to_bf16_bits(best[k]);
            *reinterpret_cast<uint4*>(zb + zo) = o;
        }
        if (amz) {                                             // two 4-bit routing codes per byte: channel 2k low, 2k+1 high
            unsigned ao = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) ao |= arg[k] << (4 * k);
            *reinterpret_cast<unsigned*>(amz + it * 256 + lane * 4) = ao;
        }
    }
}

// ------------------------------------------------------------------------------------------------ pair contraction
// dU_pad[o][2Y+dy+1][2X+dx+1][c] = sum over pairs p in list(o):  (code(amz[p][W], c) == dy*2+dx) ? dz[p][m3(W)][c] : 0
// amz [P][256][256]: two 4-bit codes per byte (channel 2k in the low nibble, 2k+1 in the high one)
// (the transpose of the expansion: dU_i = sum_j g_ij, dV_j = sum_i g_ij; one wavefront owns one (object, window),
//  so the segmented reduction needs no atomics.)
__global__ __launch_bounds__(256) void pair_contract_kernel(const u16* __restrict__ dz, const unsigned char* __restrict__ amz,
                                                            const int* __restrict__ ptr, const int* __restrict__ list,
                                                            u16* __restrict__ dU, long n_items) {
    const int lane = threadIdx.x & 63;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += (long)gridDim.x * 4) {
        const int o = (int)(it >> 8), W = (int)(it & 255);
        const int Y = W >> 4, X = W & 15;
        const int m3 = 4 * ((Y >> 1) * 8 + (X >> 1)) + (Y & 1) * 2 + (X & 1);
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[q][k] = 0.f;
        const int i0 = ptr[o], i1 = ptr[o + 1];
        auto add = [&](const uint4& g, unsigned a) __attribute__((always_inline)) {
            const u16* gh = reinterpret_cast<const u16*>(&g);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = bf16_bits_to_f32(gh[k]);
                const unsigned code = (a >> (4 * k)) & 15u;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q][k] += (code == (unsigned)q) ? v : 0.f;
            }
        };
        // Four pairs per trip: their 1 KiB gradient rows and 256 B routing rows are requested before the first is consumed
        // (one row in flight per wavefront left the kernel latency-bound at 57 % of HBM: ~40 KiB in flight per CU against the
        // ~64 KiB the 8 TB/s x 2 us product asks for).  The additions keep the list order, so the sums are bit-identical.
        int i = i0;
        for (; i + 4 <= i1; i += 4) {
            long p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = list[i + u];
            uint4 g[4];
            unsigned a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = *reinterpret_cast<const uint4*>(dz + (p[u] * 256 + m3) * 512 + lane * 8);
                a[u] = *reinterpret_cast<const unsigned*>(amz + (p[u] * 256 + W) * 256 + lane * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(g[u], a[u]);
        }
        for (; i < i1; ++i) {
            const long p = list[i];
            const uint4 g = *reinterpret_cast<const uint4*>(dz + (p * 256 + m3) * 512 + lane * 8);
            const unsigned a = *reinterpret_cast<const unsigned*>(amz + (p * 256 + W) * 256 + lane * 4);
            add(g, a);
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

// ------------------------------------------------------------------------------------------------ mask backward
// dA[img][pix][c] = sum_{o in img, pix in box_o} da[o][m(pix)][c];  dcst_part[img][block][c] = sum over the block's 16 pixels of
// sum_{o, pix not in box_o} da[...]: per-block partials in a fixed order (no float atomics: the conv1 bias gradient is
// reproducible bit for bit), reduced afterwards by sgc_slab_sum.
__global__ __launch_bounds__(256) void mask_objects_bwd_kernel(const u16* __restrict__ da, const int* __restrict__ img_ptr,
                                                               const int* __restrict__ bbox, float* __restrict__ dA,
                                                               float* __restrict__ dcst_part, int F, int D) {
    __shared__ float outs[256][9];
    const int cpp = D / 8;                                 // 16 chunks
    const int img = blockIdx.y;
    const int item = blockIdx.x * 256 + threadIdx.x;       // (pixel, chunk)
    const int pix = item / cpp, ch = item - pix * cpp;
    const int y = pix / F, x = pix - y * F;
    const int m = 4 * ((y >> 1) * (F >> 1) + (x >> 1)) + (y & 1) * 2 + (x & 1);
    float in[8], out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { in[k] = 0.f; out[k] = 0.f; }
    if (pix < F * F) {
        for (int o = img_ptr[img]; o < img_ptr[img + 1]; ++o) {
            const uint4 v = *reinterpret_cast<const uint4*>(da + ((long)o * F * F + m) * D + ch * 8);
            const u16* h = reinterpret_cast<const u16*>(&v);
            const int* b = bbox + 4 * o;
            const bool inside = x >= b[0] && x < b[1] && y >= b[2] && y < b[3];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f = bf16_bits_to_f32(h[k]);
                if (inside) in[k] += f; else out[k] += f;
            }
        }
        float* dst = dA + ((long)img * F * F + pix) * D + ch * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = in[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) outs[threadIdx.x][k] = out[k];
    __syncthreads();
    if (threadIdx.x < D) {                                 // channel c = chunk*8 + k: the block's 256/cpp pixels in order
        const int c8 = threadIdx.x >> 3, k = threadIdx.x & 7;
        float acc = 0.f;
        for (int pl = 0; pl < 256 / cpp; ++pl) acc += outs[pl * cpp + c8][k];
        dcst_part[((long)img * gridDim.x + blockIdx.x) * D + threadIdx.x] = acc;
    }
}

// dpre = dA * (1 - a^2)  (tanh backward), 16-bit out for the conv1 weight-gradient GEMM
__global__ void tanh_bwd_kernel(const float* __restrict__ dA, const u16* __restrict__ a, u16* __restrict__ dpre, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = f16_bits_to_f32(a[i]);
        dpre[i] = f32_to_bf16_bits(dA[i] * (1.f - t * t));
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
static inline int grid_for(long items, long per_block, int cap) {
    long b = (items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

static int launch_head_bwd(const HeadBwdParams& hp, hipStream_t stream) {
    const int lds = (64 * 512 + 8 * 64) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head_loss_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int waves = hp.n_pairs >= 256 * 8 * 2 ? 8 : 4;
    SGC_LAUNCH(head_loss_bwd_kernel, dim3(grid_for(hp.n_pairs, waves, 256)), dim3(64 * waves), lds, stream, hp);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// ---- weight layouts: strided gather + cast (f32 master -> one 16-bit compute copy per launch) ---------------------------------
// dst element (i0..i4) of shape dims (row-major walk of the OUTPUT, so stores coalesce) at dst_off + sum i_k * dstride[k] gets
// cast(src[src_off + sum i_k * sstride[k]]); strides in elements, any sign (a flipped 3x3 tap is stride -1 from offset 8).
// Replaces the torch view / permute / flip / cat / cast chains that rebuilt the conv2 / conv3 / fc2 layouts after every optimizer
// step (~55 small launches per step); the layouts themselves are unchanged (tests/test_gemm_gpu.py holds both forms bit for bit).
struct PermuteCastParams {
    const float* src; void* dst;
    long src_off, dst_off, total;
    int dims[5];
    long sstride[5], dstride[5];
};
template <int KIND>   // 0 f16, 1 bf16, 2 f32
__global__ void permute_cast_kernel(const PermuteCastParams p) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < p.total; i += (long)gridDim.x * blockDim.x) {
        long r = i, so = p.src_off, d_o = p.dst_off;
#pragma unroll
        for (int k = 4; k >= 0; --k) {
            const long q = r / p.dims[k];
            const int idx = (int)(r - q * p.dims[k]);
            r = q;
            so += idx * p.sstride[k];
            d_o += idx * p.dstride[k];
        }
        const float v = p.src[so];
        if constexpr (KIND == 0) reinterpret_cast<u16*>(p.dst)[d_o] = f32_to_f16_bits(v);
        else if constexpr (KIND == 1) reinterpret_cast<u16*>(p.dst)[d_o] = f32_to_bf16_bits(v);
        else reinterpret_cast<float*>(p.dst)[d_o] = v;
    }
}
// Up to 48 [rows][cols] blocks of ONE source tensor in one launch: block s goes to dst_off[s] with row pitch dst_ld[s] from
// src_off[s] (row / column strides shared).  The patch form of the conv3 data gradient stacks 36 transposed tap matrices.
struct SegmentCastParams {
    const float* src; void* dst;
    int rows, cols, n_seg;
    long s_row, s_col;
    long dst_off[48], dst_ld[48], src_off[48];
};
template <int KIND>
__global__ void segment_cast_kernel(const SegmentCastParams p) {
    const int s = blockIdx.y;
    const long n = (long)p.rows * p.cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / p.cols), c = (int)(i - (long)r * p.cols);
        const float v = p.src[p.src_off[s] + r * p.s_row + c * p.s_col];
        const long d_o = p.dst_off[s] + r * p.dst_ld[s] + c;
        if constexpr (KIND == 0) reinterpret_cast<u16*>(p.dst)[d_o] = f32_to_f16_bits(v);
        else if constexpr (KIND == 1) reinterpret_cast<u16*>(p.dst)[d_o] = f32_to_bf16_bits(v);
        else reinterpret_cast<float*>(p.dst)[d_o] = v;
    }
}

// SGD step of fc1.weight fused with the two within-row layout passes around it (the step's largest HBM-bound chain: 277 M
// parameters, 97 % of them here).  The fc1 weight-gradient GEMM leaves dW in GEMM order [n][window*1024 + channel]; the reference's
// parameter (and its momentum buffer) is [n][channel*64 + window] (model.py:118, the flatten of [1024, 8, 8]); the forward's f16 compute
// copy is again [n][window*1024 + channel].  Unfused: transpose_cast<2> (gradient -> reference order) + sgd_momentum_kernel +
// transpose_cast<0> (f16 copy) = 10.7 GB of traffic; here one pass reads gradient, weight and momentum once and writes weight,
// momentum and the f16 copy once (6.4 GB).  One workgroup = one row n x one block of 64 channels x 64 windows: the gradient tile
// goes through LDS transposed, the update runs in the parameter's own (contiguous) order with the SAME expressions as
// sgd_momentum_kernel (bit-identical weights: tests/test_gemm_gpu.py), the new weights go back through the same LDS tile for the
// f16 rows.  w1p == nullptr: no compute copy (sharded data parallelism: the copy is made after the all-gather).
__global__ __launch_bounds__(256) void sgd_fc1_fused_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ buf,
                                                            float lr, float momentum, float wd, int first, u16* __restrict__ w1p) {
    __shared__ float tile[64][65];
    const long row = (long)blockIdx.x * 65536;
    const int cb = blockIdx.y;
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const float* gs = g + row + cb * 64;                  // [window][64 channels of the block], window stride 1024
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int win = r * 4 + wq;
        tile[win][lane] = gs[(long)win * 1024 + lane];
    }
    __syncthreads();
    float* wp = w + row + cb * 4096;                      // [64 channels][64 windows] contiguous
    float* bp = buf + row + cb * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = r * 4 + wq, i = c * 64 + lane;      // lane = window
        const float wi = wp[i];
        const float gg = tile[lane][c] + wd * wi;
        const float b = first ? gg : momentum * bp[i] + gg;
        const float wn = wi - lr * b;
        bp[i] = b;
        wp[i] = wn;
        tile[lane][c] = wn;                               // the element this thread read: no hazard
    }
    if (w1p == nullptr) return;
    __syncthreads();
    u16* os = w1p + row + cb * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int win = r * 4 + wq;
        os[(long)win * 1024 + lane] = f32_to_f16_bits(tile[win][lane]);
    }
}

// Up to 32 SMALL tensors in one launch (blockIdx.y = tensor): the relation head has 22 parameter tensors, 18 of them below a million
// elements (biases, head rows, conv1) - one launch each cost more in launch gaps than in work.  Same update, scalar accesses.
struct SgdMultiParams {
    float* w[32]; const float* g[32]; float* m[32];
    long n[32];
    float lr, momentum, weight_decay;
    unsigned first_mask;
};
__global__ void sgd_momentum_multi_kernel(const SgdMultiParams p) {
    const int t = blockIdx.y;
    float* __restrict__ w = p.w[t];
    const float* __restrict__ g = p.g[t];
    float* __restrict__ m = p.m[t];
    const bool first = (p.first_mask >> t) & 1u;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < p.n[t]; i += (long)gridDim.x * blockDim.x) {
        const float wi = w[i];
        const float d = g[i] + p.weight_decay * wi;
        const float b = first ? d : p.momentum * m[i] + d;
        m[i] = b;
        w[i] = wi - p.lr * b;
    }
}

extern "C" {

int sgc_head_loss_bwd(const float* rel, const float* sup, const float* conn, const float* p, const int* tgt,
                      const float* coef_a, const float* coef_b, const float* coef_c, const float* conn_y,
                      const float* W, int n_pairs, int ng, int np, int ns, int hier, float T1, float T2, float T3,
                      float drop_scale, float* dl, float* loss, void* dpre, const float* dp_extra, const float* cs_coef,
                      const float* cand_conf, const int* cand_pred, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    HeadBwdParams hp{rel, sup, conn, p, tgt, coef_a, coef_b, coef_c, conn_y, W, n_pairs, ng, np, ns, hier,
                     1.f / T1, 1.f / T2, 1.f / T3, drop_scale, dl, loss, (u16*)dpre, dp_extra, cs_coef, cand_conf, cand_pred,
                     nullptr, nullptr, nullptr};
    return launch_head_bwd(hp, (hipStream_t)stream);
}

int sgc_head_bwd_upstream(const float* rel, const float* sup, const float* p, const float* g_rel, const float* g_sup, const float* g_conn,
                          const float* g_hidden, const float* W, int n_pairs, int ng, int np, int ns, int hier, float T1, float T2,
                          float T3, float drop_scale, float* dl, void* dpre, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    if (!g_rel || !dl || !dpre) return SGC_ERR_ARG;
    HeadBwdParams hp{rel, sup, nullptr, p, nullptr, nullptr, nullptr, nullptr, nullptr, W, n_pairs, ng, np, ns, hier,
                     1.f / T1, 1.f / T2, 1.f / T3, drop_scale, dl, nullptr, (u16*)dpre, g_hidden, nullptr, nullptr, nullptr,
                     g_rel, g_sup, g_conn};
    return launch_head_bwd(hp, (hipStream_t)stream);
}

// part: [n_blocks][64][513] f32, n_blocks = ceil(n_pairs / chunk)
int sgc_head_wgrad(const float* dl, const float* p, float* part, int n_pairs, int chunk, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    const int nb = (n_pairs + chunk - 1) / chunk;
    SGC_LAUNCH(head_wgrad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dl, p, part, n_pairs, chunk);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// SupConLossHierar forward + gradient.  F [2M][512] f32, labels [M]; G [2M][2M] and loss_rows [2M] are outputs/scratch;
// dF [2M][512] = grad_scale * dL/dF with L = mean(loss_rows).  2M <= 16384.
int sgc_supcon_hierar(const float* F, const int* labels, int M, float temperature, float grad_scale, float* G, float* loss_rows,
                      float* dF, void* stream) {
    if (M <= 0) return SGC_OK;
    const int n = 2 * M;
    if (n > 16384) return SGC_ERR_ARG;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(supcon_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (512 + 16384) * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(supcon_dfeat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              16384 * 4);
    SGC_LAUNCH(supcon_rows_kernel, dim3(n), dim3(256), (512 + n) * 4, (hipStream_t)stream, F, labels, M, 1.f / temperature, G, loss_rows);
    SGC_CHECK_LAUNCH();
    SGC_LAUNCH(supcon_dfeat_kernel, dim3(n), dim3(256), n * 4, (hipStream_t)stream, F, G, n, grad_scale / temperature, dF);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_slab_sum(const float* in, float* out, long n, int slabs, int accumulate, void* stream) {
    if (n <= 0) return SGC_OK;
    SGC_LAUNCH(slab_sum_kernel, dim3(grid_for(n, 256, 65536)), dim3(256), 0, (hipStream_t)stream, in, out, n, slabs,
                       accumulate);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_sgd_momentum_multi(int n_tensors, float* const* w, const float* const* g, float* const* momentum_buf, const long* n, float lr,
                           float momentum, float weight_decay, unsigned first_mask, void* stream) {
    if (n_tensors <= 0) return SGC_OK;
    if (n_tensors > 32) return SGC_ERR_ARG;
    SgdMultiParams p{};
    long nmax = 0;
    for (int t = 0; t < n_tensors; ++t) {
        p.w[t] = w[t]; p.g[t] = g[t]; p.m[t] = momentum_buf[t]; p.n[t] = n[t];
        nmax = n[t] > nmax ? n[t] : nmax;
    }
    p.lr = lr; p.momentum = momentum; p.weight_decay = weight_decay; p.first_mask = first_mask;
    const unsigned bx = (unsigned)((nmax + 255) / 256 < 1024 ? ((nmax + 255) / 256 > 0 ? (nmax + 255) / 256 : 1) : 1024);
    SGC_LAUNCH(sgd_momentum_multi_kernel, dim3(bx, n_tensors), dim3(256), 0, (hipStream_t)stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_sgd_fc1_fused(float* w, const float* g_gemm_order, float* momentum_buf, int rows, float lr, float momentum, float weight_decay,
                      int first_step, void* w1p_f16, void* stream) {
    if (rows <= 0) return SGC_OK;
    if ((((uintptr_t)w | (uintptr_t)g_gemm_order | (uintptr_t)momentum_buf) & 3) != 0) return SGC_ERR_ARG;
    SGC_LAUNCH(sgd_fc1_fused_kernel, dim3((unsigned)rows, 16), dim3(256), 0, (hipStream_t)stream, w, g_gemm_order, momentum_buf, lr, momentum,
               weight_decay, first_step, (u16*)w1p_f16);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_sgd_momentum_step(float* w, const float* g, float* momentum_buf, long n, float lr, float momentum, float weight_decay,
                          int first_step, void* stream) {
    if (n <= 0) return SGC_OK;
    if ((((uintptr_t)w | (uintptr_t)g | (uintptr_t)momentum_buf) & 15) != 0)       // a view that is not 16-byte aligned: scalar accesses
        SGC_LAUNCH(sgd_momentum_scalar_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream, w, g, momentum_buf, n,
                   lr, momentum, weight_decay, first_step);
    else
        SGC_LAUNCH(sgd_momentum_kernel, dim3(grid_for(n, 1024, 16384)), dim3(256), 0, (hipStream_t)stream, w, g, momentum_buf, n, lr,
                   momentum, weight_decay, first_step);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// part [row_blocks][cols] f32; returns row_blocks through *n_parts
int sgc_colsum(int elem, const void* X, float* part, long rows, int cols, int row_blocks, void* stream) {
    if (cols % 8 || row_blocks < 1) return SGC_ERR_ARG;
    if (rows <= 0) return SGC_OK;
    const long rpb = (rows + row_blocks - 1) / row_blocks;
    dim3 grid((cols / 8 + 31) / 32, row_blocks);
    if (elem == ELEM_F16)
        SGC_LAUNCH(colsum_kernel<ELEM_F16>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)X, part, rows, cols, rpb);
    else
        SGC_LAUNCH(colsum_kernel<ELEM_BF16>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)X, part, rows, cols, rpb);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_segment_sum_rows(const void* X, const int* ptr, const int* list, float* out, int n_seg, int cols, void* stream) {
    if (cols != 512) return SGC_ERR_ARG;
    if (n_seg <= 0) return SGC_OK;
    SGC_LAUNCH(segment_sum_rows_kernel, dim3(n_seg), dim3(64), 0, (hipStream_t)stream, (const u16*)X, ptr, list, out, cols);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_transpose_cast(const float* src, void* dst, int out_kind, int na, int nb, long sa_s, long sb_s, long ss_i, long sa_d,
                       long sb_d, long ds_j, void* stream) {
    if (na <= 0 || nb <= 0) return SGC_OK;
    dim3 grid(na, nb);
    if (out_kind == 0) SGC_LAUNCH(transpose_cast_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, sa_s, sb_s, ss_i, sa_d, sb_d, ds_j);
    else if (out_kind == 1) SGC_LAUNCH(transpose_cast_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, sa_s, sb_s, ss_i, sa_d, sb_d, ds_j);
    else if (out_kind == 2) SGC_LAUNCH(transpose_cast_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, sa_s, sb_s, ss_i, sa_d, sb_d, ds_j);
    else return SGC_ERR_ARG;
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_permute_cast(const float* src, void* dst, int out_kind, int ndim, const int* dims, const long* src_strides, const long* dst_strides,
                     long src_off, long dst_off, void* stream) {
    if (ndim < 1 || ndim > 5 || out_kind < 0 || out_kind > 2) return SGC_ERR_ARG;
    PermuteCastParams p{};
    p.src = src; p.dst = dst; p.src_off = src_off; p.dst_off = dst_off; p.total = 1;
    for (int k = 0; k < 5; ++k) { p.dims[k] = 1; p.sstride[k] = 0; p.dstride[k] = 0; }
    for (int k = 0; k < ndim; ++k) {                     // right-aligned: the last given dimension is the fastest
        const int d = 5 - ndim + k;
        if (dims[k] <= 0) return SGC_OK;
        p.dims[d] = dims[k]; p.sstride[d] = src_strides[k]; p.dstride[d] = dst_strides[k];
        p.total *= dims[k];
    }
    const long blocks = (p.total + 255) / 256 < 4096 ? (p.total + 255) / 256 : 4096;
    if (out_kind == 0) SGC_LAUNCH(permute_cast_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else if (out_kind == 1) SGC_LAUNCH(permute_cast_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else SGC_LAUNCH(permute_cast_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_segment_cast(const float* src, void* dst, int out_kind, int rows, int cols, long src_row_stride, long src_col_stride, int n_seg,
                     const long* dst_off, const long* dst_ld, const long* src_off, void* stream) {
    if (n_seg < 0 || n_seg > 48 || out_kind < 0 || out_kind > 2) return SGC_ERR_ARG;
    if (n_seg == 0 || rows <= 0 || cols <= 0) return SGC_OK;
    SegmentCastParams p{};
    p.src = src; p.dst = dst; p.rows = rows; p.cols = cols; p.n_seg = n_seg; p.s_row = src_row_stride; p.s_col = src_col_stride;
    for (int s = 0; s < n_seg; ++s) { p.dst_off[s] = dst_off[s]; p.dst_ld[s] = dst_ld[s]; p.src_off[s] = src_off[s]; }
    const long n = (long)rows * cols;
    const unsigned bx = (unsigned)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    if (out_kind == 0) SGC_LAUNCH(segment_cast_kernel<0>, dim3(bx, n_seg), dim3(256), 0, (hipStream_t)stream, p);
    else if (out_kind == 1) SGC_LAUNCH(segment_cast_kernel<1>, dim3(bx, n_seg), dim3(256), 0, (hipStream_t)stream, p);
    else SGC_LAUNCH(segment_cast_kernel<2>, dim3(bx, n_seg), dim3(256), 0, (hipStream_t)stream, p);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_convert_f16_bf16(const void* in, void* out, long n, void* stream) {
    if (n % 8) return SGC_ERR_ARG;
    if (n <= 0) return SGC_OK;
    SGC_LAUNCH(convert_f16_bf16_kernel, dim3(grid_for(n / 8, 256, 65536)), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)in, (u16*)out, n / 8);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

// ---- fc2
// dh1 [n_pairs][4096] bf16 = (dpre [n_pairs][512] * w2mT[4096][512]^T) masked by h1 > 0 (x drop_scale)
int sgc_fc2_dgrad(const void* dpre, const void* w2mT, const void* h1, void* dh1, int n_pairs, float drop_scale, void* stream) {
    NtParams p{};
    p.A = (const u16*)dpre; p.B = (const u16*)w2mT; p.C = dh1; p.M = n_pairs; p.N = 4096; p.K = 512;
    p.lda = 512; p.ldb = 512; p.ldc = 4096; p.mask_src = (const u16*)h1; p.scale = drop_scale;
    return launch_gemm_nt<ELEM_BF16, AMODE_PLAIN, EPI_RELUMASK>(p, (hipStream_t)stream);
}
// slabs [splits][512][4096] f32 = dpre^T * h1_bf16 over n_rows (multiple of 64, zero padded)
int sgc_fc2_wgrad(const void* dpre, const void* h1_bf16, float* slabs, int n_rows, int splits, int* n_slabs, void* stream) {
    TnParams p{};
    p.A = (const u16*)dpre; p.B = (const u16*)h1_bf16; p.C = slabs; p.M = 512; p.N = 4096; p.K = n_rows;
    p.lda = 512; p.ldb = 4096; p.ldc = 4096; p.slab_stride = 512L * 4096;
    return launch_gemm_tn<ELEM_BF16, BMODE_PLAIN>(p, splits, n_slabs, (hipStream_t)stream);
}
// ---- fc1
// dy [n_pairs][K] bf16 = dh1 [n_pairs][4096] * w1pT[K][4096]^T
int sgc_fc1_dgrad(const void* dh1, const void* w1pT, void* dy, int n_pairs, int K, void* stream) {
    NtParams p{};
    p.A = (const u16*)dh1; p.B = (const u16*)w1pT; p.C = dy; p.M = n_pairs; p.N = K; p.K = 4096;
    p.lda = 4096; p.ldb = 4096; p.ldc = K;
    return launch_gemm_nt<ELEM_BF16, AMODE_PLAIN, EPI_STORE>(p, (hipStream_t)stream);
}
// dW1p [4096][K] f32 = dh1^T * y_bf16 over n_rows (multiple of 64, zero padded); no split (16384 tiles)
int sgc_fc1_wgrad(const void* dh1, const void* y_bf16, float* dw, int n_rows, int K, void* stream) {
    TnParams p{};
    p.A = (const u16*)dh1; p.B = (const u16*)y_bf16; p.C = dw; p.M = 4096; p.N = K; p.K = n_rows;
    p.lda = 4096; p.ldb = K; p.ldc = K; p.slab_stride = 0;
    return launch_gemm_tn<ELEM_BF16, BMODE_PLAIN>(p, 1, nullptr, (hipStream_t)stream);
}
// ---- conv3
// dbias_part: [*n_parts][1024] f32 per-block partial sums of the routed gradient (conv3 bias gradient)
int sgc_unpool_relu_bwd(const void* dy, const unsigned char* argmax, void* dy3_pad, float* dbias_part, int* n_parts,
                        int n_pairs, void* stream) {
    if (n_pairs <= 0) { if (n_parts) *n_parts = 0; return SGC_OK; }
    const long n_win = (long)n_pairs * 64;
    const int blocks = grid_for(n_win, 2, 2048);
    if (n_parts) *n_parts = blocks;
    SGC_LAUNCH(unpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, argmax,
                       (u16*)dy3_pad, dbias_part, n_win);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// The same un-pooling, also emitting the packed 2:4 operand of sgc_conv3_wgrad_sparse in the one pass over dy / argmax
int sgc_unpool_relu_bwd_pack(const void* dy, const unsigned char* argmax, void* dy3_pad, float* dbias_part, int* n_parts,
                             void* pack_ac, void* pack_ic, int n_pairs, void* stream) {
    if (n_pairs <= 0) { if (n_parts) *n_parts = 0; return SGC_OK; }
    return launch_unpool_pack((const u16*)dy, argmax, (u16*)dy3_pad, dbias_part, n_parts, (u16*)pack_ac, (unsigned*)pack_ic,
                              n_pairs, (hipStream_t)stream);
}
// dz [n_pairs*256][512] bf16 = conv3x3(dy3_pad, wd3[512][9][1024])
int sgc_conv3_dgrad(const void* dy3_pad, const void* wd3, void* dz, int n_pairs, void* stream) {
    NtParams p{};
    p.A = (const u16*)dy3_pad; p.B = (const u16*)wd3; p.C = dz; p.M = n_pairs * 256; p.N = 512; p.K = 9 * 1024;
    p.ldb = 9 * 1024; p.ldc = 512; p.lgS = 4; p.Cin = 1024;
    return launch_gemm_nt<ELEM_BF16, AMODE_CONV, EPI_STORE>(p, (hipStream_t)stream);
}
// The same data gradient straight from the POOLED gradient dy [n_pairs*64][1024] bf16 and the forward routing byte: the block
// un-pools into its LDS patch (conv16_halo_pp_kernel<.., ASRC = 1>), no un-pooled tensor exists in memory.
int sgc_conv3_dgrad_pooled(const void* dy, const unsigned char* argmax, const void* wd3, void* dz, int n_pairs, void* stream) {
    if (!dy || !argmax) return SGC_ERR_ARG;
    NtParams p{};
    p.A = (const u16*)dy; p.Apool = (const u16*)dy; p.Acode = argmax; p.B = (const u16*)wd3; p.C = dz; p.M = n_pairs * 256; p.N = 512;
    p.K = 9 * 1024; p.ldb = 9 * 1024; p.ldc = 512; p.lgS = 4; p.Cin = 1024;
    return launch_conv16_halo_pp<ELEM_BF16, EPI_STORE>(p, (hipStream_t)stream);
}
// slabs [splits][1024][9*512] f32 = sum_pix dy3[pix][n] * z_pad_bf16[pix+tap][c]
int sgc_conv3_wgrad(const void* dy3_pad, const void* z_pad_bf16, float* slabs, int n_pairs, int splits, int* n_slabs, void* stream) {
    TnParams p{};
    p.A = (const u16*)dy3_pad; p.B = (const u16*)z_pad_bf16; p.C = slabs; p.M = 1024; p.N = 9 * 512; p.K = n_pairs * 256;
    p.ldc = 9 * 512; p.slab_stride = 1024L * 9 * 512; p.lgS = 4; p.Cin = 512; p.CinA = 1024;
    return launch_gemm_tn<ELEM_BF16, BMODE_CONV, 1>(p, splits, n_slabs, (hipStream_t)stream);
}
// The same gradient on the sparse matrix cores (gemm_tn_sp.h): dy [n_pairs*64][1024] bf16 is the POOLED gradient (fc1 data
// gradient, not yet un-pooled), argmax the conv3 forward routing byte (0..3, 4 = ReLU killed); pack_ac (n_pairs*4*1024*64 B)
// and pack_ic (n_pairs*4*1024*8 B) are scratch for the packed 2:4 operand.
int sgc_conv3_wgrad_sparse(const void* dy, const unsigned char* argmax, const void* z_pad_bf16, void* pack_ac, void* pack_ic,
                           float* slabs, int n_pairs, int splits, int* n_slabs, void* stream) {
    if (n_pairs <= 0) { if (n_slabs) *n_slabs = 0; return SGC_OK; }
    if (dy) {                                              // dy == NULL: pack_ac / pack_ic already hold the operand (sgc_unpool_relu_bwd_pack)
        int rc = launch_sparse_pack((const u16*)dy, argmax, (u16*)pack_ac, (unsigned*)pack_ic, n_pairs, (hipStream_t)stream);
        if (rc != SGC_OK) return rc;
    }
    TnParams p{};
    p.A = nullptr; p.B = (const u16*)z_pad_bf16; p.C = slabs; p.M = 1024; p.N = 9 * 512; p.K = n_pairs * 256;
    p.ldc = 9 * 512; p.slab_stride = 1024L * 9 * 512; p.lgS = 4; p.Cin = 512; p.CinA = 1024;
    return launch_gemm_tn_sp(p, (const u16*)pack_ac, (const unsigned*)pack_ic, splits, n_slabs, (hipStream_t)stream);
}
// The conv3 weight gradient over a window LIST on the sparse matrix cores (csrc/kernels_shared.hip: the pair-specific windows):
// the first n_entries (a multiple of 16) listed windows, whose un-pooled gradient has one non-zero per window and channel (the real
// pairs' windows - the per-object entries behind them collect sums of several windows and stay on the dense block).  dywm: pooled
// gradient rows (window-major row space, row dest[e]); argmax: routing bytes at gather[e]; zpatch: the windows' 4 x 4 input patches
// (sgc_windows_im2patch*); pack_ac (n_entries/16 * 64 KiB) / pack_ic (n_entries/16 * 8 KiB): scratch for the packed operand.
int sgc_windows_wgrad_patch_sparse(const void* dywm, const unsigned char* argmax, const int* gather, const int* dest, int n_entries,
                                   const void* zpatch, void* pack_ac, void* pack_ic, float* slabs, int splits, int* n_slabs, void* stream) {
    if (n_entries <= 0) { if (n_slabs) *n_slabs = 0; return SGC_OK; }
    if (n_entries & 15) return SGC_ERR_ARG;
    const int n_tiles = n_entries / 16;
    SGC_LAUNCH(windows_sparse_pack_kernel, dim3((unsigned)n_tiles, 8), dim3(256), 0, (hipStream_t)stream, (const u16*)dywm, argmax, gather, dest,
               n_entries, (u16*)pack_ac, (unsigned*)pack_ic, n_tiles);
    SGC_CHECK_LAUNCH();
    TnParams p{};
    p.A = nullptr; p.B = (const u16*)zpatch; p.C = slabs; p.M = 1024; p.N = 9 * 512; p.K = n_entries * 4;
    p.ldc = 9 * 512; p.slab_stride = 1024L * 9 * 512; p.lgS = 4; p.Cin = 512; p.CinA = 1024;
    return launch_gemm_tn_sp<1>(p, (const u16*)pack_ac, (const unsigned*)pack_ic, splits, n_slabs, (hipStream_t)stream);
}
// The same product with the second operand read straight from the forward's f16 maps z_pad_f16 [pairs][18][18][512] through the window
// list (gemm_tn_sp_kernel<2>: per-window bases by scalar loads one K tile ahead, f16 -> bf16 in registers): no patch copy of the
// n_entries windows.  Same bits as sgc_windows_im2patch_f16 + sgc_windows_wgrad_patch_sparse (same conversion, same products, same order).
// Every one of the n_entries list entries must be a window of a map whose 4 x 4 patch rows are written (the real pairs' windows are).
int sgc_windows_wgrad_gather_sparse(const void* dywm, const unsigned char* argmax, const int* gather, const int* dest, int n_entries,
                                    const void* z_pad_f16, void* pack_ac, void* pack_ic, float* slabs, int splits, int* n_slabs, void* stream) {
    if (n_entries <= 0) { if (n_slabs) *n_slabs = 0; return SGC_OK; }
    if (n_entries & 15) return SGC_ERR_ARG;
    const int n_tiles = n_entries / 16;
    SGC_LAUNCH(windows_sparse_pack_kernel, dim3((unsigned)n_tiles, 8), dim3(256), 0, (hipStream_t)stream, (const u16*)dywm, argmax, gather, dest,
               n_entries, (u16*)pack_ac, (unsigned*)pack_ic, n_tiles);
    SGC_CHECK_LAUNCH();
    TnParams p{};
    p.A = nullptr; p.B = (const u16*)z_pad_f16; p.C = slabs; p.M = 1024; p.N = 9 * 512; p.K = n_entries * 4;
    p.ldc = 9 * 512; p.slab_stride = 1024L * 9 * 512; p.lgS = 4; p.Cin = 512; p.CinA = 1024; p.gather = gather;
    return launch_gemm_tn_sp<2>(p, (const u16*)pack_ac, (const unsigned*)pack_ic, splits, n_slabs, (hipStream_t)stream);
}
// ---- expansion / contraction
int sgc_pair_expand_train(const void* U, const void* V, const int* sub_idx, const int* obj_idx, void* z_pad_f16, void* z_pad_bf16,
                          unsigned char* amz, int n_pairs, void* stream) {
    if (n_pairs <= 0) return SGC_OK;
    const long items = (long)n_pairs * 256;
    const int blocks = grid_for(items, 4, 262144);
    SGC_LAUNCH(pair_expand_train_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)U, (const u16*)V, sub_idx,
               obj_idx, (u16*)z_pad_f16, (u16*)z_pad_bf16, amz, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_pair_contract(const void* dz, const unsigned char* amz, const int* ptr, const int* list, void* dU_pad, int n_obj,
                      void* stream) {
    if (n_obj <= 0) return SGC_OK;
    const long items = (long)n_obj * 256;
    SGC_LAUNCH(pair_contract_kernel, dim3(grid_for(items, 4, 262144)), dim3(256), 0, (hipStream_t)stream, (const u16*)dz,
                       amz, ptr, list, (u16*)dU_pad, items);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// ---- conv2 (per role)
// da [n_obj*1024][128] bf16 = conv3x3(dU_pad, wd2[128][9][512])
int sgc_conv2_dgrad(const void* dU_pad, const void* wd2, void* da, int n_obj, void* stream) {
    NtParams p{};
    p.A = (const u16*)dU_pad; p.B = (const u16*)wd2; p.C = da; p.M = n_obj * 1024; p.N = 128; p.K = 9 * 512;
    p.ldb = 9 * 512; p.ldc = 128; p.lgS = 5; p.Cin = 512;
    return launch_gemm_nt<ELEM_BF16, AMODE_CONV, EPI_STORE>(p, (hipStream_t)stream);
}
// slabs [splits][512][9*128] f32 = sum_pix dU[pix][n] * a_pad_bf16[pix+tap][c]
int sgc_conv2_wgrad(const void* dU_pad, const void* a_pad_bf16, float* slabs, int n_obj, int splits, int* n_slabs, void* stream) {
    TnParams p{};
    p.A = (const u16*)dU_pad; p.B = (const u16*)a_pad_bf16; p.C = slabs; p.M = 512; p.N = 9 * 128; p.K = n_obj * 1024;
    p.ldc = 9 * 128; p.slab_stride = 512L * 9 * 128; p.lgS = 5; p.Cin = 128; p.CinA = 512;
    return launch_gemm_tn<ELEM_BF16, BMODE_CONV, 1>(p, splits, n_slabs, (hipStream_t)stream);
}
// ---- masks / conv1
int sgc_object_masked_maps_bwd(const void* da, const int* img_ptr, const int* bbox, float* dA, float* dcst_part, int* n_parts,
                               int n_img, int F, int D, void* stream) {
    if (D != 128) return SGC_ERR_ARG;
    const int items = F * F * (D / 8);
    const int blocks = (items + 255) / 256;
    if (n_parts) *n_parts = blocks * (n_img > 0 ? n_img : 0);
    if (n_img <= 0) return SGC_OK;
    SGC_LAUNCH(mask_objects_bwd_kernel, dim3(blocks, n_img), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)da, img_ptr, bbox, dA, dcst_part, F, D);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
int sgc_tanh_bwd(const float* dA, const void* a_img, void* dpre, long n, void* stream) {
    if (n <= 0) return SGC_OK;
    SGC_LAUNCH(tanh_bwd_kernel, dim3(grid_for(n, 256, 65536)), dim3(256), 0, (hipStream_t)stream, dA, (const u16*)a_img,
                       (u16*)dpre, n);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}
// slabs [splits][128][XC] f32 = dpre^T * x_bf16 over n_rows image pixels
int sgc_conv1_wgrad(const void* dpre, const void* x_bf16, float* slabs, int n_rows, int XC, int splits, int* n_slabs, void* stream) {
    TnParams p{};
    p.A = (const u16*)dpre; p.B = (const u16*)x_bf16; p.C = slabs; p.M = 128; p.N = XC; p.K = n_rows;
    p.lda = 128; p.ldb = XC; p.ldc = XC; p.slab_stride = 128L * XC;
    return launch_gemm_tn<ELEM_BF16, BMODE_PLAIN>(p, splits, n_slabs, (hipStream_t)stream);
}

}  // extern "C"
