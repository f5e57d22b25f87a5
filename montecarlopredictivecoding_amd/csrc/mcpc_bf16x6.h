// fp32 products on the bf16 matrix pipe ("bf16x6"): the split of an fp32 value into three bf16 pieces and the MFMA wrapper.
//
// x = hi + mid + lo (8 + 8 + 8 significant bits, round to nearest even at every cut); a product of two such sums is nine bf16 x bf16
// products, each exact in fp32.  The six whose magnitude is above 2^-24 of the leading one,
//      a b  ~  a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid),
// accumulated in fp32 small terms first, give a dot product whose error against fp64 is that of the fp32 MFMA chain (measured on
// the Hebbian GEMM's shapes, K = 4096, relative to sum|terms|: max 2.7e-7 / rms 3.1e-8 against 2.1e-7 / 2.7e-8,
// scripts/heb_bf16_ubench.hip) at 6 x 16 cycles per 32-deep block of a 16 x 16 tile instead of 8 x 32.
// Used by the Hebbian GEMM (mcpc_hebbian.h: mcpc_heb6_kernel) and by the optional step-kernel GEMM core (mcpc_gemm6.h).
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
__device__ __forceinline__ f32x4 mfma6(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace mcpc
