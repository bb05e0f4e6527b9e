// Test hooks: raw access to the two GEMM engines (used by tests/test_gemm_gpu.py only).
#include "gemm_nt.h"
#include "gemm_tn.h"

extern "C" {

int sgc_dbg_gemm_nt(int elem, const void* A, const void* B, void* C, int M, int N, int K,
                    long lda, long ldb, long ldc, const float* bias, void* stream) {
    NtParams p{};
    p.A = (const u16*)A; p.B = (const u16*)B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.bias = bias;
    if (elem == ELEM_F16) return launch_gemm_nt<ELEM_F16, AMODE_PLAIN, EPI_STORE>(p, (hipStream_t)stream);
    return launch_gemm_nt<ELEM_BF16, AMODE_PLAIN, EPI_STORE>(p, (hipStream_t)stream);
}

// ablation hook for tools/gemm_microbench.py (bf16, 256x256 block): abl as documented at gemm_nt_kernel
int sgc_dbg_gemm_nt_abl(int abl, const void* A, const void* B, void* C, int M, int N, int K, void* stream) {
    NtParams p{};
    p.A = (const u16*)A; p.B = (const u16*)B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = K; p.ldb = K; p.ldc = N;
    if ((N % 256) || (K % 64)) return SGC_ERR_ARG;
    switch (abl) {
        case 0: return launch_gemm_nt_cfg<ELEM_BF16, AMODE_PLAIN, EPI_STORE, 2, 4, 4, 2, 0>(p, (hipStream_t)stream);
        case 1: return launch_gemm_nt_cfg<ELEM_BF16, AMODE_PLAIN, EPI_STORE, 2, 4, 4, 2, 1>(p, (hipStream_t)stream);
        case 2: return launch_gemm_nt_cfg<ELEM_BF16, AMODE_PLAIN, EPI_STORE, 2, 4, 4, 2, 2>(p, (hipStream_t)stream);
        case 3: return launch_gemm_nt_cfg<ELEM_BF16, AMODE_PLAIN, EPI_STORE, 2, 4, 4, 2, 3>(p, (hipStream_t)stream);
        case 5: return launch_gemm_nt_cfg<ELEM_BF16, AMODE_PLAIN, EPI_STORE, 2, 4, 4, 2, 5>(p, (hipStream_t)stream);
        case 4: return launch_gemm_nt_ring<ELEM_BF16, AMODE_PLAIN, EPI_STORE>(p, (hipStream_t)stream);
        case 6: p.epi_lds = 1; return launch_gemm_nt_pp<ELEM_BF16, EPI_STORE, 0>(p, (hipStream_t)stream);
        case 7: p.epi_lds = 1; return launch_gemm_nt_pp<ELEM_BF16, EPI_STORE, 1>(p, (hipStream_t)stream);
        case 8: return launch_gemm_nt_w4<ELEM_BF16, EPI_STORE, 0>(p, (hipStream_t)stream);
        case 9: return launch_gemm_nt_w4<ELEM_BF16, EPI_STORE, 1>(p, (hipStream_t)stream);
        case 10: p.epi_lds = 1; return launch_gemm_nt_pp1<ELEM_BF16, EPI_STORE>(p, (hipStream_t)stream);
    }
    return SGC_ERR_ARG;
}

// A: zero-padded channels-last images [n_img][S+2][S+2][Cin]; B: [N][Cin/64][9][64]; C: [n_img*S*S][N] window-major rows
int sgc_dbg_conv_nt(int elem, const void* A, const void* B, void* C, int n_img, int lgS, int Cin, int N,
                    const float* bias, void* stream) {
    NtParams p{};
    p.A = (const u16*)A; p.B = (const u16*)B; p.C = C; p.M = n_img << (2 * lgS); p.N = N; p.K = 9 * Cin;
    p.ldb = 9L * Cin; p.ldc = N; p.lgS = lgS; p.Cin = Cin; p.bias = bias;
    if (elem == ELEM_F16) return launch_gemm_nt<ELEM_F16, AMODE_CONV, EPI_STORE>(p, (hipStream_t)stream);
    return launch_gemm_nt<ELEM_BF16, AMODE_CONV, EPI_STORE>(p, (hipStream_t)stream);
}

// C slabs [splits][M][N] f32; returns number of slabs written through *slabs
int sgc_dbg_gemm_tn(int elem, const void* A, const void* B, float* C, int M, int N, int K,
                    long lda, long ldb, int splits, int* slabs, void* stream) {
    TnParams p{};
    p.A = (const u16*)A; p.B = (const u16*)B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = N; p.slab_stride = (long)M * N;
    int r = (elem == ELEM_F16) ? launch_gemm_tn<ELEM_F16, BMODE_PLAIN>(p, splits, slabs, (hipStream_t)stream)
                               : launch_gemm_tn<ELEM_BF16, BMODE_PLAIN>(p, splits, slabs, (hipStream_t)stream);
    return r;
}

// A: [n_img*S*S][M] rows window-major; B: padded images [n_img][S+2][S+2][Cin]; C slabs [splits][M][9*Cin]
int sgc_dbg_conv_tn(int elem, const void* A, const void* B, float* C, int M, int n_img, int lgS, int Cin,
                    int splits, int* slabs, void* stream) {
    TnParams p{};
    p.A = (const u16*)A; p.B = (const u16*)B; p.C = C; p.M = M; p.N = 9 * Cin; p.K = n_img << (2 * lgS);
    p.lda = M; p.ldc = 9L * Cin; p.slab_stride = (long)M * 9 * Cin; p.lgS = lgS; p.Cin = Cin;
    int r = (elem == ELEM_F16) ? launch_gemm_tn<ELEM_F16, BMODE_CONV>(p, splits, slabs, (hipStream_t)stream)
                               : launch_gemm_tn<ELEM_BF16, BMODE_CONV>(p, splits, slabs, (hipStream_t)stream);
    return r;
}

// ds_read_b64_tr_b16 semantics probe: LDS filled with its own element index; every lane reads at byte addr[lane]
__global__ void tr_probe_kernel(const int* addr, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)((char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int sgc_dbg_tr_probe(const int* addr, short* out, void* stream) {
    SGC_LAUNCH(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
