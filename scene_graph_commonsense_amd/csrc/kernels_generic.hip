// Generic-size trunk of the relation classifier (reference model.py:138-150, :110-111: ``BayesianRelationClassifier(args, input_dim,
// feature_size, ...)`` accepts ANY input_dim / feature_size).  The tiled MFMA kernels of this library are specialised to the sizes every
// shipped configuration of the reference uses (input_dim = 128, feature_size = 32, main.py:49-85); these kernels serve every OTHER size:
// plain f32 HIP, one thread per output element, the reference's per-pair graph literally (no sharing identities), so that a module
// constructed with non-default sizes still runs on the GPU through the same C-ABI and drop-in classes instead of raising.  They are
// not tuned (small-shape unit configurations: input_dim = 16, feature_size = 8 is 1.2 MFLOP per pair); what follows the trunk - fc2 with
// the label gather, the Bayesian head, loss and their backward - does not depend on the two sizes and runs on the ordinary kernels.
//
// Layouts (all f32, channels-last): a [P][F*F][2C] (tanh(conv1) of both roles, concatenated as model.py:141), z [P][(F/2)^2][4C] and
// y [P][(F/4)^2][8C] (after ReLU + 2x2 max-pool) with one routing byte per element (dy*2+dx of the first maximum, 4 = no positive value:
// the ReLU kills the window's gradient - torch's relu'(0) = 0 with max_pool2d's first-index tie rule gives exactly that).
// Inputs are read where they lie: ``feat`` / ``depth`` NCHW with per-image strides (a minibatch's [B,2C,F,F] + [B,1,F,F], or the
// pre-masked [b,2C+1,F,F] crops of the per-step forward() with depth = channel 2C), masked by the pair side's box on the fly
// (train_test.py:164-169,194-195: feature * mask, depth * mask; boxes already slice-normalised).
#include "common.h"

namespace {

__device__ __forceinline__ float masked_input(const float* __restrict__ feat, const float* __restrict__ depth, long sf, long sd, int img,
                                              const int* __restrict__ box, int k, int pix, int C2, int F) {
    const int y = pix / F, x = pix - y * F;
    if (x < box[0] || x >= box[1] || y < box[2] || y >= box[3]) return 0.f;
    return k < C2 ? feat[img * sf + (long)k * F * F + pix] : depth[img * sd + pix];
}

// a[p][pix][side*C + c] = tanh(b[side][c] + sum_k W[side][c][k] * x_side[p][k][pix])
__global__ void generic_conv1_tanh_kernel(const float* __restrict__ feat, const float* __restrict__ depth, long sf, long sd,
                                          const int* __restrict__ img, const int* __restrict__ box, const float* __restrict__ w,
                                          const float* __restrict__ b, int P, int C, int F, float* __restrict__ a) {
    const long n = (long)P * F * F * 2 * C;
    const int C2 = 2 * C, K = 2 * C + 1;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % (2 * C));
        const long r = i / (2 * C);
        const int pix = (int)(r % (F * F)), p = (int)(r / (F * F));
        const int side = ch / C, c = ch - side * C;
        const int im = img[side * P + p];
        const int* bx = box + ((long)side * P + p) * 4;
        const float* wr = w + ((long)side * C + c) * K;
        float acc = b[side * C + c];
        for (int k = 0; k < K; ++k) acc += wr[k] * masked_input(feat, depth, sf, sd, im, bx, k, pix, C2, F);
        a[i] = tanhf(acc);
    }
}

// in [P][S][S][Cin] -> out [P][S/2][S/2][Cout] = maxpool2(relu(conv3x3(in, W [Cout][Cin][3][3], pad 1) + b)), code: routing byte
__global__ void generic_conv3x3_relu_pool_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b, int P,
                                                 int S, int Cin, int Cout, float* __restrict__ out, unsigned char* __restrict__ code) {
    const int H = S / 2;
    const long n = (long)P * H * H * Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const long r = i / Cout;
        const int win = (int)(r % (H * H)), p = (int)(r / (H * H));
        const int wy = win / H, wx = win - wy * H;
        float best = 0.f;
        int arg = 4;
        for (int q = 0; q < 4; ++q) {
            const int y = 2 * wy + (q >> 1), x = 2 * wx + (q & 1);
            float acc = b[co];
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= S) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    if (xx < 0 || xx >= S) continue;
                    const float* ip = in + (((long)p * S + yy) * S + xx) * Cin;
                    const float* wp = w + ((long)co * Cin * 3 + ky) * 3 + kx;          // W[co][ci][ky][kx], stride 9 over ci
                    for (int ci = 0; ci < Cin; ++ci) acc += ip[ci] * wp[(long)ci * 9];
                }
            }
            if (acc > best) { best = acc; arg = q; }
        }
        out[i] = best;
        code[i] = (unsigned char)arg;
    }
}

// dpre [P][S][S][C] from the pooled gradient [P][S/2][S/2][C] and the routing bytes
__global__ void generic_unpool_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ code, int P, int S, int C,
                                      float* __restrict__ dpre) {
    const int H = S / 2;
    const long n = (long)P * H * H * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;
        const int win = (int)(r % (H * H)), p = (int)(r / (H * H));
        const int wy = win / H, wx = win - wy * H;
        const int a = code[i];
        const float g = dout[i];
        for (int q = 0; q < 4; ++q)
            dpre[(((long)p * S + 2 * wy + (q >> 1)) * S + 2 * wx + (q & 1)) * C + c] = (q == a) ? g : 0.f;
    }
}

// din[p][y][x][ci] = sum_{ky,kx,co} dpre[p][y-ky+1][x-kx+1][co] * W[co][ci][ky][kx]
__global__ void generic_conv3x3_bwd_data_kernel(const float* __restrict__ dpre, const float* __restrict__ w, int P, int S, int Cin, int Cout,
                                                float* __restrict__ din) {
    const long n = (long)P * S * S * Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const long r = i / Cin;
        const int pix = (int)(r % (S * S)), p = (int)(r / (S * S));
        const int y = pix / S, x = pix - y * S;
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - ky + 1;
            if (yy < 0 || yy >= S) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x - kx + 1;
                if (xx < 0 || xx >= S) continue;
                const float* gp = dpre + (((long)p * S + yy) * S + xx) * Cout;
                const float* wp = w + ((long)ci * 3 + ky) * 3 + kx;                    // + co * Cin * 9
                for (int co = 0; co < Cout; ++co) acc += gp[co] * wp[(long)co * Cin * 9];
            }
        }
        din[i] = acc;
    }
}

// dW[co][ci][ky][kx] = sum_{p,y,x} dpre[p][y][x][co] * in[p][y+ky-1][x+kx-1][ci]; one thread per weight, fixed summation order
__global__ void generic_conv3x3_bwd_weight_kernel(const float* __restrict__ dpre, const float* __restrict__ in, int P, int S, int Cin, int Cout,
                                                  float* __restrict__ dw) {
    const long n = (long)Cout * Cin * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % 9);
        const long r = i / 9;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const int ky = t / 3, kx = t - ky * 3;
        float acc = 0.f;
        for (int p = 0; p < P; ++p)
            for (int y = 0; y < S; ++y) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= S) continue;
                for (int x = 0; x < S; ++x) {
                    const int xx = x + kx - 1;
                    if (xx < 0 || xx >= S) continue;
                    acc += dpre[(((long)p * S + y) * S + x) * Cout + co] * in[(((long)p * S + yy) * S + xx) * Cin + ci];
                }
            }
        dw[i] = acc;
    }
}

// column sums of a [rows][C] f32 matrix: one thread per column
__global__ void generic_colsum_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (long r = 0; r < rows; ++r) acc += x[r * C + c];
    out[c] = acc;
}

// h1[p][n] = dropout(relu(b[n] + sum_k W1[n][k] * yflat[p][k])), k = c * Q + pix (NCHW flatten, model.py:148), y [P][Q][C8]
__global__ void generic_fc1_relu_kernel(const float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ b, int P, int Q, int C8,
                                        int dropout, uint32_t seed, u16* __restrict__ h1) {
    const long n = (long)P * 4096;
    const int K = Q * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i & 4095), p = (int)(i >> 12);
        const float* wr = w + (long)o * K;
        const float* yp = y + (long)p * K;
        float acc = b[o];
        for (int c = 0; c < C8; ++c)
            for (int q = 0; q < Q; ++q) acc += wr[c * Q + q] * yp[q * C8 + c];
        acc = fmaxf(acc, 0.f);
        if (dropout) acc = dropout_keep(seed, (uint32_t)i) ? 2.f * acc : 0.f;
        h1[i] = f32_to_f16_bits(acc);
    }
}

// dy[p][q][c] = sum_n dh1[p][n] * W1[n][c*Q+q]   (dh1: bf16 gradient wrt fc1's pre-activation, ReLU / dropout already applied)
__global__ void generic_fc1_bwd_data_kernel(const u16* __restrict__ dh1, const float* __restrict__ w, int P, int Q, int C8, float* __restrict__ dy) {
    const long n = (long)P * Q * C8;
    const int K = Q * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        const long r = i / C8;
        const int q = (int)(r % Q), p = (int)(r / Q);
        const u16* g = dh1 + (long)p * 4096;
        const float* wc = w + c * Q + q;
        float acc = 0.f;
        for (int o = 0; o < 4096; ++o) acc += bf16_bits_to_f32(g[o]) * wc[(long)o * K];
        dy[i] = acc;
    }
}

// dW1[n][c*Q+q] = sum_p dh1[p][n] * y[p][q][c];  db1[n] = sum_p dh1[p][n] (threads with k == 0)
__global__ void generic_fc1_bwd_weight_kernel(const u16* __restrict__ dh1, const float* __restrict__ y, int P, int Q, int C8, float* __restrict__ dw,
                                              float* __restrict__ db) {
    const int K = Q * C8;
    const long n = (long)4096 * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), o = (int)(i / K);
        const int c = k / Q, q = k - c * Q;
        float acc = 0.f, sb = 0.f;
        for (int p = 0; p < P; ++p) {
            const float g = bf16_bits_to_f32(dh1[(long)p * 4096 + o]);
            acc += g * y[((long)p * Q + q) * C8 + c];
            sb += g;
        }
        dw[i] = acc;
        if (k == 0) db[o] = sb;
    }
}

// da (gradient wrt a = tanh(pre1)) -> dpre1 in place
__global__ void generic_tanh_bwd_kernel(const float* __restrict__ a, float* __restrict__ da, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float t = a[i];
        da[i] *= 1.f - t * t;
    }
}

// dW[side][c][k] = sum_{p,pix} dpre1[p][pix][side*C+c] * x_side[p][k][pix];  db[side][c] = sum dpre1 (threads with k == 0)
__global__ void generic_conv1_bwd_weight_kernel(const float* __restrict__ feat, const float* __restrict__ depth, long sf, long sd,
                                                const int* __restrict__ img, const int* __restrict__ box, const float* __restrict__ dpre, int P, int C,
                                                int F, float* __restrict__ dw, float* __restrict__ db) {
    const int K = 2 * C + 1, C2 = 2 * C;
    const long n = (long)2 * C * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long r = i / K;
        const int c = (int)(r % C), side = (int)(r / C);
        float acc = 0.f, sb = 0.f;
        for (int p = 0; p < P; ++p) {
            const int im = img[side * P + p];
            const int* bx = box + ((long)side * P + p) * 4;
            for (int pix = 0; pix < F * F; ++pix) {
                const float g = dpre[((long)p * F * F + pix) * C2 + side * C + c];
                acc += g * masked_input(feat, depth, sf, sd, im, bx, k, pix, C2, F);
                sb += g;
            }
        }
        dw[i] = acc;
        if (k == 0) db[side * C + c] = sb;
    }
}

inline int grid_for(long n) { return (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256 < 1 ? 1 : (n + 255) / 256); }

}  // namespace

extern "C" {

int sgc_generic_conv1_tanh(const float* feat, const float* depth, long stride_feat, long stride_depth, const int* img, const int* box,
                           const float* w1, const float* b1, int n_pairs, int C, int F, float* a, void* stream) {
    if (n_pairs < 0 || C <= 0 || F <= 0) return SGC_ERR_ARG;
    if (n_pairs == 0) return SGC_OK;
    SGC_LAUNCH(generic_conv1_tanh_kernel, dim3(grid_for((long)n_pairs * F * F * 2 * C)), dim3(256), 0, (hipStream_t)stream, feat, depth,
               stride_feat, stride_depth, img, box, w1, b1, n_pairs, C, F, a);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_generic_conv3x3_relu_pool(const float* in, const float* w, const float* b, int n_pairs, int S, int Cin, int Cout, float* out,
                                  unsigned char* code, void* stream) {
    if (n_pairs < 0 || S <= 0 || (S & 1) || Cin <= 0 || Cout <= 0) return SGC_ERR_ARG;
    if (n_pairs == 0) return SGC_OK;
    SGC_LAUNCH(generic_conv3x3_relu_pool_kernel, dim3(grid_for((long)n_pairs * (S / 2) * (S / 2) * Cout)), dim3(256), 0, (hipStream_t)stream, in, w, b,
               n_pairs, S, Cin, Cout, out, code);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_generic_conv3x3_bwd(const float* in, const float* w, const float* dout, const unsigned char* code, int n_pairs, int S, int Cin, int Cout,
                            float* dpre, float* din, float* dw, float* db, void* stream) {
    if (n_pairs <= 0 || S <= 0 || (S & 1) || Cin <= 0 || Cout <= 0) return SGC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    SGC_LAUNCH(generic_unpool_kernel, dim3(grid_for((long)n_pairs * (S / 2) * (S / 2) * Cout)), dim3(256), 0, st, dout, code, n_pairs, S, Cout, dpre);
    SGC_CHECK_LAUNCH();
    if (din) {
        SGC_LAUNCH(generic_conv3x3_bwd_data_kernel, dim3(grid_for((long)n_pairs * S * S * Cin)), dim3(256), 0, st, dpre, w, n_pairs, S, Cin, Cout, din);
        SGC_CHECK_LAUNCH();
    }
    SGC_LAUNCH(generic_conv3x3_bwd_weight_kernel, dim3(grid_for((long)Cout * Cin * 9)), dim3(256), 0, st, dpre, in, n_pairs, S, Cin, Cout, dw);
    SGC_CHECK_LAUNCH();
    SGC_LAUNCH(generic_colsum_kernel, dim3((Cout + 63) / 64), dim3(64), 0, st, dpre, (long)n_pairs * S * S, Cout, db);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_generic_fc1_relu(const float* y, const float* w1, const float* b1, int n_pairs, int Q, int C8, int dropout, unsigned seed, void* h1,
                         void* stream) {
    if (n_pairs < 0 || Q <= 0 || C8 <= 0) return SGC_ERR_ARG;
    if (n_pairs == 0) return SGC_OK;
    SGC_LAUNCH(generic_fc1_relu_kernel, dim3(grid_for((long)n_pairs * 4096)), dim3(256), 0, (hipStream_t)stream, y, w1, b1, n_pairs, Q, C8, dropout,
               (uint32_t)seed, (u16*)h1);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_generic_fc1_bwd(const void* dh1, const float* y, const float* w1, int n_pairs, int Q, int C8, float* dy, float* dw1, float* db1,
                        void* stream) {
    if (n_pairs <= 0 || Q <= 0 || C8 <= 0) return SGC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    SGC_LAUNCH(generic_fc1_bwd_data_kernel, dim3(grid_for((long)n_pairs * Q * C8)), dim3(256), 0, st, (const u16*)dh1, w1, n_pairs, Q, C8, dy);
    SGC_CHECK_LAUNCH();
    SGC_LAUNCH(generic_fc1_bwd_weight_kernel, dim3(grid_for((long)4096 * Q * C8)), dim3(256), 0, st, (const u16*)dh1, y, n_pairs, Q, C8, dw1, db1);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

int sgc_generic_conv1_bwd(const float* feat, const float* depth, long stride_feat, long stride_depth, const int* img, const int* box, const float* a,
                          float* da, int n_pairs, int C, int F, float* dw1, float* db1, void* stream) {
    if (n_pairs <= 0 || C <= 0 || F <= 0) return SGC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)n_pairs * F * F * 2 * C;
    SGC_LAUNCH(generic_tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, da, n);
    SGC_CHECK_LAUNCH();
    SGC_LAUNCH(generic_conv1_bwd_weight_kernel, dim3(grid_for((long)2 * C * (2 * C + 1))), dim3(256), 0, st, feat, depth, stride_feat, stride_depth, img,
               box, da, n_pairs, C, F, dw1, db1);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
