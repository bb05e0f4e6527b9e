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
        case 6: p.epi_lds = 1; return launch_gemm_nt_pp<ELEM_BF16, EPI_STORE, 0>(p, (hipStream_t)stream);
        case 7: p.epi_lds = 1; return launch_gemm_nt_pp<ELEM_BF16, EPI_STORE, 1>(p, (hipStream_t)stream);
#ifdef SGC_EXPERIMENTS      // rejected main-loop variants, built with SGC_EXPERIMENTS=1 only
        case 4: return launch_gemm_nt_ring<ELEM_BF16, AMODE_PLAIN, EPI_STORE>(p, (hipStream_t)stream);
        case 8: return launch_gemm_nt_w4<ELEM_BF16, EPI_STORE, 0>(p, (hipStream_t)stream);
        case 9: return launch_gemm_nt_w4<ELEM_BF16, EPI_STORE, 1>(p, (hipStream_t)stream);
        case 10: p.epi_lds = 1; return launch_gemm_nt_pp1<ELEM_BF16, EPI_STORE>(p, (hipStream_t)stream);
#endif
    }
    return SGC_ERR_ARG;
}

// tools/stride_microbench.py: the ping-pong NT block with operand row pitches as parameters (do power-of-two pitches cost cache channels?)
int sgc_dbg_gemm_nt_ld(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, void* stream) {
    NtParams p{};
    p.A = (const u16*)A; p.B = (const u16*)B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = N; p.epi_lds = 1;
    if ((N % 256) || (K % 64) || (lda % 8) || (ldb % 8)) return SGC_ERR_ARG;
    return launch_gemm_nt_pp<ELEM_BF16, EPI_STORE, 0>(p, (hipStream_t)stream);
}

// tools/fc1_windows_microbench.py: the grouped fc1 product over window-major rows (sgc_fc1_windows_gemm) with the weight layout and the
// epilogue as parameters.  mode 0: f32 tile in the MFMA's C layout (4-byte stores), 1: transposed tile (16-byte stores), 2: the same
// without its stores (C ignored), 3: f16 output through the LDS-staged epilogue.
int sgc_dbg_fc1_windows_gemm(const void* ywm, const void* w, const int* tile_group, void* owm, int rows, long ldb, long group_stride,
                             long ldc, int mode, int stagger, int phases, unsigned long long* clk, void* stream) {
    if (rows <= 0 || (rows & 255)) return SGC_ERR_ARG;
    NtParams p{};
    p.A = (const u16*)ywm; p.B = (const u16*)w; p.C = mode == 2 ? nullptr : owm; p.M = rows; p.N = 4096; p.K = 1024;
    p.lda = 1024; p.ldb = ldb; p.ldc = ldc; p.tile_group = tile_group; p.group_stride = group_stride;
    p.stagger = stagger; p.stagger_phases = phases; p.clk = clk;
    switch (mode) {
        case 0: return launch_gemm_nt_pp<ELEM_F16, EPI_STORE_F32>(p, (hipStream_t)stream);
        case 1: case 2: return launch_gemm_nt_pp<ELEM_F16, EPI_STORE_F32T>(p, (hipStream_t)stream);
        case 3: p.epi_lds = 1; return launch_gemm_nt_pp<ELEM_F16, EPI_STORE>(p, (hipStream_t)stream);
        case 11: p.nt_store = 8; return launch_gemm_nt_pp<ELEM_F16, EPI_STORE_F32T>(p, (hipStream_t)stream);      // tile-contiguous output
        case 4: case 5: case 6: case 7: p.nt_store = mode - 3; return launch_gemm_nt_pp<ELEM_F16, EPI_STORE_F32T>(p, (hipStream_t)stream);   // 1 with nt / sc1 / sc0 sc1 / sc0 sc1 nt
    }
    return SGC_ERR_ARG;
}

// tools/dgrad_patch_microbench.py: sgc_windows_dgrad_patches with the layouts as parameters (lda / seg_stride: elements between the
// windows / between the own pixels of a window in dy3x; bpad: padding of B_pp's rows; split: 1 = the centre pixels in two slots of two combinations)
int sgc_dbg_dgrad_patches(const void* dy3x, const void* w3patch, void* patch, int entries, long lda, long seg_stride, int bpad, int split,
                          void* stream) {
    NtParams p{};
    p.A = (const u16*)dy3x; p.B = (const u16*)w3patch; p.C = patch; p.M = entries; p.N = (split ? 20 : 16) * 512; p.K = 4096;
    p.lda = lda; p.ldb = 0; p.ldc = p.N; p.seg_stride = seg_stride; p.seg_bpad = bpad & 0xffff; p.patch_gn = bpad >> 16; p.seg_split = split;      // (bpad: low 16 bits = padding, high bits = patch width in virtual N tiles, 0 = default)
    return launch_gemm_nt_pp_seg<ELEM_BF16>(p, (hipStream_t)stream);
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

// v_smfmac_f32_32x32x32_bf16 semantics probe: every lane supplies its 8 compressed A values, 16 B values and the index word
__global__ void smfmac_probe_kernel(const u16* a, const u16* b, const int* idx, float* c, int abid) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef __bf16 bf16x16_t __attribute__((ext_vector_type(16)));
    const int l = threadIdx.x;
    bf16x8_t av; bf16x16_t bv;
    for (int i = 0; i < 8; ++i) { u16 t = a[l * 8 + i]; av[i] = __builtin_bit_cast(__bf16, t); }
    for (int i = 0; i < 16; ++i) { u16 t = b[l * 16 + i]; bv[i] = __builtin_bit_cast(__bf16, t); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (abid == 0) acc = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(av, bv, acc, idx[l], 0, 0);
    else if (abid == 1) acc = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(av, bv, acc, idx[l], 0, 1);
    else acc = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(av, bv, acc, idx[l], 1, 0);
    for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}
int sgc_dbg_smfmac_probe(const void* a, const void* b, const int* idx, float* c, int abid, void* stream) {
    SGC_LAUNCH(smfmac_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const u16*)a, (const u16*)b, idx, c, abid);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
// MFMA issue-rate probe: 8 independent accumulators, `iters` rounds of 8 instructions per wave; mode 0 dense 32x32x16, 1 sparse 32x32x32
template <int mode>
__global__ __launch_bounds__(512, 2) void mfma_rate_kernel(float* out, int iters) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef __bf16 bf16x16_t __attribute__((ext_vector_type(16)));
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a; bf16x16_t b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x * 0.001f + i);
    for (int i = 0; i < 16; ++i) b[i] = (__bf16)(float)(threadIdx.x * 0.002f + i);
    const bf16x8_t b8 = __builtin_shufflevector(b, b, 0, 1, 2, 3, 4, 5, 6, 7);
    const int idx = 0x4444;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b8, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, acc[i], idx, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}
extern "C" {
int sgc_dbg_mfma_rate(float* out, int blocks, int iters, int mode, void* stream) {
    if (mode == 0) SGC_LAUNCH(mfma_rate_kernel<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    else SGC_LAUNCH(mfma_rate_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
    SGC_CHECK_LAUNCH();
    return SGC_OK;
}

}  // extern "C"
