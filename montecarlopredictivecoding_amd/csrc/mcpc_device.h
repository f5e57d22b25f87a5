// Device-side helpers shared by the MCPC kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
// same vector in the global address space: loads through it are global_load_* (counted on vmcnt only);
// a generic pointer read from a descriptor in memory would compile to flat_load_* and force full drains
typedef __attribute__((address_space(1))) f32x4 gf32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));       // 8 fp16 of an MFMA operand (mcpc_gemm_f16.h)
typedef __attribute__((address_space(1))) u32x4 gu32x4;

namespace mcpc {

// ---- Philox4x32-10 (Salmon et al. SC'11).  Bit-exact twin: oracle/philox.py -------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one v_mad_u64_u32 per product (hi and lo together) instead of v_mul_hi_u32 + v_mul_lo_u32
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Two u32 -> two standard normals (Box-Muller).  u1 in (0,1], u2 in [0,1) (fraction of a turn).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = (float)((a >> 8) + 1u) * 0x1p-24f;
    const float u2 = (float)(b >> 8) * 0x1p-24f;
    // -2 ln(u1) = -2 ln2 * log2(u1); v_log_f32 is log2, v_sin/v_cos take turns.
    // v_sqrt_f32 directly (1 ulp): the IEEE-refined sqrtf costs ~12 more instructions per normal pair
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

// Four normals for units 4g..4g+3 of `layer` of global chain `chain` at global step `step`.
__device__ __forceinline__ f32x4 normals4(uint64_t seed, uint64_t step, uint32_t layer, uint32_t chain, uint32_t group) {
    uint32_t r[4];
    philox4x32_10(chain, (layer << 24) | group, (uint32_t)step, (uint32_t)(step >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    f32x4 z;
    float a, b;
    box_muller(r[0], r[1], a, b); z.x = a; z.y = b;
    box_muller(r[2], r[3], a, b); z.z = a; z.w = b;
    return z;
}

// ---- activations (reference utils/model.py:49-52: nn.ReLU / nn.Tanh; none for the linear toys) --
__device__ __forceinline__ float tanh_f(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exp via v_exp_f32 (2^x).  |err| ~ 1e-7 absolute.
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // exp(2x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float act_f(int a, float x) {
    return a == 1 ? fmaxf(x, 0.0f) : (a == 2 ? tanh_f(x) : x);
}
// derivative given x and f(x)
__device__ __forceinline__ float act_d(int a, float x, float fx) {
    return a == 1 ? (x > 0.0f ? 1.0f : 0.0f) : (a == 2 ? 1.0f - fx * fx : 1.0f);
}

// Adam's x update, operation for operation what torch.optim.Adam's single-tensor path does (torch/optim/adam.py, _single_tensor_adam):
//     denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps);  param.addcdiv_(exp_avg, denom, value=-step_size)
// i.e. x + ((-step_size) * m) / (sqrt(v) / sqrt(bias2) + eps) -- addcdiv multiplies by the scalar FIRST and divides second (ATen
// PointwiseOpsKernel: self + alpha * t1 / t2) -- with a correctly rounded square root and two correctly rounded divisions (hipcc's
// default for HIP: -fhip-fp32-correctly-rounded-divide-sqrt).  Rounds 1-4 used v_sqrt_f32 / v_rcp_f32 (1 ulp each, ~27 VALU
// instructions per element fewer) and held the MAP path's energies to 3e-6 instead of 1e-6; the epilogue waves hide their arithmetic
// behind the GEMM waves (profiles/r04_k1_bounds.txt items 3, 8), so the exact forms cost nothing measurable.
__device__ __forceinline__ float adam_x(float x, float m, float v, float neg_step_size, float bc2_sqrt, float eps) {
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    return x + (neg_step_size * m) / denom;
}
// the moments, as torch updates them: exp_avg.lerp_(grad, 1 - beta1) = fma(w, g - m, m) (ATen lerp, weight < 0.5);
// exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2) = fma((1 - beta2) * g, g, v * beta2)
__device__ __forceinline__ float adam_m(float m, float g, float omb1) { return __builtin_fmaf(omb1, g - m, m); }
__device__ __forceinline__ float adam_v(float v, float g, float beta2, float omb2) { return __builtin_fmaf(omb2 * g, g, v * beta2); }

__device__ __forceinline__ float sigmoid_f(float o) {
    // same arithmetic as sigmoid_bce_f below, so that recording the loss never changes a trajectory
    const float e = __builtin_amdgcn_exp2f(-fabsf(o) * 1.4426950408889634f);  // exp(-|o|)
    return (o >= 0.0f ? 1.0f : e) * __builtin_amdgcn_rcpf(1.0f + e);
}
// BCEWithLogits element: max(o,0) - o*y + log1p(exp(-|o|))
__device__ __forceinline__ float bce_logits_f(float o, float y) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(o) * 1.4426950408889634f);
    // log1p(e), e in (0,1]: log2(1+e)*ln2 via v_log_f32 (absolute error ~1e-7 per element)
    return fmaxf(o, 0.0f) - o * y + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e);
}

// sigmoid(o) and the BCE-with-logits term from ONE exp: with e = exp(-|o|),
//   sigmoid(o) = (o >= 0 ? 1 : e) / (1 + e),   bce = max(o,0) - o*y + log(1 + e)
__device__ __forceinline__ void sigmoid_bce_f(float o, float y, float& sig, float& bce) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(o) * 1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    sig = (o >= 0.0f ? 1.0f : e) * r;
    bce = fmaxf(o, 0.0f) - o * y + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e);
}

// Sum over the 64 lanes, returned wave-uniform.  DPP row shifts and row broadcasts (gfx9 family: row_shr:n, row_bcast:15,
// row_bcast:31) instead of six dependent ds_bpermute round trips through the LDS (~1.1 k cycles of latency per sum on an
// epilogue wave whose partner streams MFMAs; seven sums per step when energies are recorded): a fixed order, bitwise
// reproducible, the total lands in lane 63.
__device__ __forceinline__ float wave_sum(float v) {
    auto dpp_add = [](float x, auto ctrl, auto row_mask) {
        const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true);
        return x + __builtin_bit_cast(float, moved);
    };
    using std::integral_constant;
    v = dpp_add(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});      // row_shr:1
    v = dpp_add(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});      // row_shr:2
    v = dpp_add(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});      // row_shr:4
    v = dpp_add(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});      // row_shr:8  -> lane 15 of a row: its sum
    v = dpp_add(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});      // row_bcast:15 into rows 1 and 3
    v = dpp_add(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});      // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Maximum over the 64 lanes of a non-negative value, returned wave-uniform (the DPP steps of wave_sum with v_max_f32; NaN inputs are
// skipped by v_max).
__device__ __forceinline__ float wave_max(float v) {
    auto dpp_max = [](float x, auto ctrl, auto row_mask) {
        const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true);
        return fmaxf(x, __builtin_bit_cast(float, moved));
    };
    using std::integral_constant;
    v = dpp_max(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});      // row_shr:1   (lanes shifted in from outside read 0: v >= 0)
    v = dpp_max(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});      // row_shr:2
    v = dpp_max(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});      // row_shr:4
    v = dpp_max(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});      // row_shr:8  -> lane 15 of a row: its maximum
    v = dpp_max(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});      // row_bcast:15 into rows 1 and 3
    v = dpp_max(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});      // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

}  // namespace mcpc
