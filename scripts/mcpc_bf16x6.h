// fp32 products on the bf16 matrix pipe ("bf16x6"): the split of an fp32 value into three bf16 pieces and the MFMA wrapper.
//
// x = hi + mid + lo (8 + 8 + 8 significant bits, round to nearest even at every cut); a product of two such sums is nine bf16 x bf16
// products, each exact in fp32.  The six whose magnitude is above 2^-24 of the leading one,
//      a b  ~  a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid),
// accumulated in fp32 small terms first, give a dot product whose error against fp64 is that of the fp32 MFMA chain (measured on
// the Hebbian GEMM's shapes, K = 4096, relative to sum|terms|: max 2.7e-7 / rms 3.1e-8 against 2.1e-7 / 2.7e-8,
// scripts/heb_bf16_ubench.hip) at 6 x 16 cycles per 32-deep block of a 16 x 16 tile instead of 8 x 32.
// Rounds 3-4: the arithmetic of the Hebbian GEMM and of the step kernels' GEMM core.  Round 5 moved both to two scaled fp16 pieces
// (csrc/mcpc_gemm_f16.h); this header left the product and stays here for scripts/k1_decomp_ubench.hip, which models the bf16x6 kernel.
#pragma once

namespace mcpc {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// two floats -> packed bf16x2 (round to nearest even), low half = a
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16lo_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16hi_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// x = hi + mid + lo for a pair of values (exact unless x has more than 24 significant bits below its bf16 exponent range)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a, b);
    const float ra = a - bf16lo_f32(h), rb = b - bf16hi_f32(h);
    m = pk_bf16(ra, rb);
    l = pk_bf16(ra - bf16lo_f32(m), rb - bf16hi_f32(m));
}
// The same split for the step kernels' GEMM loops, where its instruction count is what the loop waits for (2.6 VALU instructions
// per MFMA).  Written like split3_pair, hipcc converts the low element of every pair twice (`(pack(a, b)) << 16` becomes a second
// v_cvt_pk_bf16_f32 of a alone) and subtracts element by element: 13 instructions per pair.  Here the conversion is opaque to the
// optimiser (one v_cvt_pk_bf16_f32 per pair and level) and the residuals of a pair are ONE packed subtraction (v_pk_add_f32 with
// negated second operand): 9 per pair, 36 instead of 52 per 8 values.  Same arithmetic, bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16_opaque(f32x2 v) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y));
    return r;
}
__device__ __forceinline__ void split3_pair_fast(f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16_opaque(x);
    const f32x2 r = x - f32x2{bf16lo_f32(h), bf16hi_f32(h)};
    m = cvt_pk_bf16_opaque(r);
    const f32x2 r2 = r - f32x2{bf16lo_f32(m), bf16hi_f32(m)};
    l = cvt_pk_bf16_opaque(r2);
}
__device__ __forceinline__ f32x4 mfma6(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace mcpc
