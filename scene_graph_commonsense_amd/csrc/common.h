// Shared device helpers for the gfx950 (CDNA4) relation-head kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

enum { ELEM_F16 = 0, ELEM_BF16 = 1 };

// Launch-time selections among kernel variants.  The product library uses the measured defaults below (profiles/README.md) and
// reads NO environment variable; a library built with -DSGC_EXPERIMENTS (SGC_EXPERIMENTS=1 python -m ...build, for
// tools/gemm_microbench.py / tools/hook_sweep.sh) reads the SGC_* variables ONCE, in a thread-safe function-local static, for A/B runs.
struct SgcTuning {
    int gemm_cfg = 0;        // SGC_GEMM_CFG: 1 = 128x128 block, 2 = 2-stage 256x256, 3 = 4-stage ring, 4 = halo conv (2-stage), 5 / 7 = ping-pong plain / halo
    int gemm_ring = 0;       // SGC_GEMM_RING: 4-stage ring kernel
    int gemm_pp = 1;         // SGC_GEMM_PP: ping-pong 256x256 loops (0: 2-stage)
    int conv_halo = 1;       // SGC_CONV_HALO: halo-staged implicit 3x3 convolution (0: plain implicit GEMM)
    int epi_lds = 1;         // SGC_EPI_LDS: 16-bit outputs transposed through LDS into 16-byte stores
    int acg_aligned = 0;     // SGC_ACG_ALIGNED: gathered conv grid padded to whole per-XCD patches
    int nt_aligned = -1;     // SGC_NT_ALIGNED: -1 = by grid size
    int halo_walk = -1;      // SGC_HALO_WALK: -1 = by weight-tile size
    int tn_xcd = 1;          // SGC_TN_XCD: XCD-aware assignment of the 4 x 18 conv weight-gradient tile grid
    int tn_patch = 1;        // SGC_TN_PATCH: per-XCD 4x8 patches for TN grids with >= 8 M tiles
    int gather_pp = 1;       // SGC_GATHER_PP: ping-pong block for the gathered conv3 forward
    int owm_pitch = 4160;    // SGC_OWM_PITCH: row pitch (floats) of fc1's f32 products over the window-major rows; 4096 + 256 B (a power of two is 10 % slower)
    int f32_swap = 1;        // SGC_F32_SWAP: f32 products of fc1 over the window-major rows leave as 16-byte stores (operands swapped in the MFMA)
    int fc1_wgrad_group_xcd = 0;   // SGC_FC1_WGRAD_XCD (read in every build: the A/B of profiles/r06_fc1_wgrad_xcd_ab.txt): 1 = fc1's grouped weight gradient
                                   // runs half a window-position group (32 tiles) per XCD at a time.  Measured and left OFF: whole groups per XCD
                                   // 2.39 -> 2.77 ms (eight sequences of unequal groups), half groups 2.37 -> 2.41 ms and the step unchanged -
                                   // the fabric traffic it saves (9.5 -> ~4.8 GB) was Infinity-Cache hits that cost the neighbours nothing
};
inline const SgcTuning& sgc_tuning() {
    static const SgcTuning t = [] {
        SgcTuning v;
#ifdef SGC_EXPERIMENTS
        auto rd = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        v.gemm_cfg = rd("SGC_GEMM_CFG", v.gemm_cfg);       v.gemm_ring = rd("SGC_GEMM_RING", v.gemm_ring);
        v.gemm_pp = rd("SGC_GEMM_PP", v.gemm_pp);          v.conv_halo = rd("SGC_CONV_HALO", v.conv_halo);
        v.epi_lds = rd("SGC_EPI_LDS", v.epi_lds);          v.acg_aligned = rd("SGC_ACG_ALIGNED", v.acg_aligned);
        v.nt_aligned = rd("SGC_NT_ALIGNED", v.nt_aligned); v.halo_walk = rd("SGC_HALO_WALK", v.halo_walk);
        v.tn_xcd = rd("SGC_TN_XCD", v.tn_xcd);             v.tn_patch = rd("SGC_TN_PATCH", v.tn_patch);
        v.gather_pp = rd("SGC_GATHER_PP", v.gather_pp);    v.f32_swap = rd("SGC_F32_SWAP", v.f32_swap);
        v.owm_pitch = rd("SGC_OWM_PITCH", v.owm_pitch);
#endif
        if (const char* e = getenv("SGC_FC1_WGRAD_XCD")) v.fc1_wgrad_group_xcd = atoi(e);
        return v;
    }();
    return t;
}

// f32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, one VALU op per PAIR); the
// integer emulation (NaN test + add + shift, ~6 ops per value) made the bf16 epilogues of the GEMM blocks VALU-bound
// (128 values per lane: fc1 data gradient 13.3 -> see profiles/README.md).
__device__ __forceinline__ u16 f32_to_bf16_bits(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(u16, h);
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_bits(float lo, float hi) {      // lo in bits [15:0]
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_bits_to_f32(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ u16 f32_to_f16_bits(float f) {
    f = fminf(fmaxf(f, -65504.f), 65504.f);        // saturate instead of overflowing to inf (NaN passes through)
    f16 h = (f16)f;
    return *reinterpret_cast<u16*>(&h);
}
__device__ __forceinline__ float f16_bits_to_f32(u16 b) {
    f16 h = *reinterpret_cast<f16*>(&b);
    return (float)h;
}
template <int ELEM> __device__ __forceinline__ u16 to_elem(float f) {
    if constexpr (ELEM == ELEM_F16) return f32_to_f16_bits(f); else return f32_to_bf16_bits(f);
}
template <int ELEM> __device__ __forceinline__ float from_elem(u16 b) {
    if constexpr (ELEM == ELEM_F16) return f16_bits_to_f32(b); else return bf16_bits_to_f32(b);
}

// Counter-based hash shared with scene_graph_commonsense_amd/synthetic.py (lowbias32).
__device__ __host__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// Dropout keep-bit for element idx of a stream: p = 0.5 (reference model.py:120-121).
__device__ __host__ __forceinline__ bool dropout_keep(uint32_t seed, uint32_t idx) {
    return (lowbias32(idx ^ (seed * 0x9E3779B1U + 0x7F4A7C15U)) >> 16) & 1U;
}

// Packed half-open pixel rectangle on the 16-grid (Y0 | Y1 << 5 | X0 << 10 | X1 << 15, 0 = empty): the pixels of a pair that the
// shared-window conv3 path needs (csrc/kernels_shared.hip).
__device__ __forceinline__ bool in_pixel_rect(int r, int Y, int X) {
    return Y >= (r & 31) && Y < ((r >> 5) & 31) && X >= ((r >> 10) & 31) && X < ((r >> 15) & 31);
}

#define SGC_OK 0
#define SGC_ERR_ARG 1
#define SGC_ERR_LAUNCH 2

#define SGC_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return SGC_ERR_LAUNCH;        \
    } while (0)

// hipGetLastError() also reports stale errors of unrelated earlier runtime calls (e.g. a benign failed pointer
// query made by the host framework); clear it before our launch so the check after it is about OUR launch.
#define SGC_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
