// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA 32x32 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "gwdepth.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GWD_CHECK_LAUNCH()                       \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
    // d/dv [ v * Phi(v) ] = Phi(v) + v * phi(v)
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
    return cdf + v * pdf;
}

// GELU for the bf16 kernels: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, four orders below bf16's rounding) - one exp, one
// reciprocal and a degree-5 polynomial instead of erff's ~40 instructions; the exponential is shared with the density in the
// gradient.  The activation-backward + bias-gradient pass and the GELU LayerNorm kernels were VALU-bound on erff + expf (DESIGN.md
// section 5).  The fp32 parity mode keeps erff (gelu_f / gelu_grad_f): gelu_t<T> picks.
__device__ __forceinline__ void gelu_parts_fast(float v, float &cdf, float &pdf_e) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
    const float e = __expf(-0.5f * v * v);                // = exp(-x^2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f + (v < 0.f ? -0.5f : 0.5f) * erf_abs;
    pdf_e = e;
}
__device__ __forceinline__ float gelu_fast(float v) {
    float cdf, e;
    gelu_parts_fast(v, cdf, e);
    return v * cdf;
}
__device__ __forceinline__ float gelu_grad_fast(float v) {
    float cdf, e;
    gelu_parts_fast(v, cdf, e);
    return cdf + v * 0.39894228040143267794f * e;
}
template <typename T> __device__ __forceinline__ float gelu_t(float v) {
    if constexpr (std::is_same<T, float>::value) return gelu_f(v);
    else return gelu_fast(v);
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float v) {
    if constexpr (std::is_same<T, float>::value) return gelu_grad_f(v);
    else return gelu_grad_fast(v);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case GWD_ACT_RELU: return v > 0.f ? v : 0.f;
        case GWD_ACT_GELU: return gelu_f(v);
        case GWD_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case GWD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// gwd_conv_desc.gate: backward of an activation from its OUTPUT rv (what act_bwd_kernel computes with act_scale = 1)
__device__ __forceinline__ float gate_grad(float g, float rv, int act) {
    if (act == GWD_ACT_RELU) return rv > 0.f ? g : 0.f;
    return g * (rv > 0.f ? 1.0f : rv + 1.0f);             // GWD_ACT_ELU (check_desc admits nothing else)
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over the 64 lanes as a wave-UNIFORM value: six DPP adds on the VALU (quad swaps, row mirrors, row broadcasts)
// and one v_readlane - no LDS crossbar (a __shfl_xor butterfly is six ds_bpermute round trips) and the result can live
// in an SGPR.  Needs all 64 lanes active (inactive data must already be zero).
__device__ __forceinline__ float wave_sum_uniform(float v) {
    auto add_dpp = [](float x, auto ctrl, auto row_mask) {
        constexpr int C = decltype(ctrl)::value, RM = decltype(row_mask)::value;
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, RM, 0xf, true));
    };
    v = add_dpp(v, std::integral_constant<int, 0xb1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
    v = add_dpp(v, std::integral_constant<int, 0x4e>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
    v = add_dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
    v = add_dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror: every lane = its row's sum
    v = add_dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast15 into rows 1, 3
    v = add_dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Sum over each aligned group of LPR consecutive lanes (LPR = 8, 16, 32, 64), result in every lane of the group: DPP
// for the strides inside a 16-lane row, ds_bpermute only across rows.
template <int LPR>
__device__ __forceinline__ float segment_sum(float v) {
    auto add_dpp = [](float x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, 0xf, 0xf, true));
    };
    static_assert(LPR == 8 || LPR == 16 || LPR == 32 || LPR == 64, "segment width");
    v = add_dpp(v, std::integral_constant<int, 0xb1>{});            // lane ^ 1
    v = add_dpp(v, std::integral_constant<int, 0x4e>{});            // lane ^ 2
    v = add_dpp(v, std::integral_constant<int, 0x141>{});           // row_half_mirror: completes each group of 8
    if constexpr (LPR >= 16) v = add_dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror: groups of 16
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LPR >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// max over the 64 lanes, wave-uniform, same DPP ladder (the old value of a lane outside the row mask is the lane's own)
__device__ __forceinline__ float wave_max_uniform(float v) {
    auto max_dpp = [](float x, auto ctrl, auto row_mask) {
        constexpr int C = decltype(ctrl)::value, RM = decltype(row_mask)::value;
        const int xi = __builtin_bit_cast(int, x);
        return fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, C, RM, 0xf, false)));
    };
    v = max_dpp(v, std::integral_constant<int, 0xb1>{}, std::integral_constant<int, 0xf>{});
    v = max_dpp(v, std::integral_constant<int, 0x4e>{}, std::integral_constant<int, 0xf>{});
    v = max_dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});
    v = max_dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});
    v = max_dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    v = max_dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own L2.  When `per` consecutive workgroups share one
// window's rows of a packed operand (the head groups of a window attention), this renumbering keeps them on ONE XCD: the rows
// are fetched into one L2 instead of up to 8, and the partial lines the heads write merge there before they go to memory.
// Groups of 8 * per ids: physical j * 8 + g  ->  virtual g * per + j; a ragged last group keeps its ids.
__device__ __forceinline__ long xcd_grouped_block(long b, long nb, int per) {
    const long grp = 8L * per, base = (b / grp) * grp;
    if (base + grp > nb) return b;
    const long r = b - base;
    return base + (r % 8) * per + r / 8;
}
