// GEMM core of the step kernels, "f16x4" form (round 5): fp32 products on the fp16 matrix pipe with per-row power-of-two scaling.
//
// Every contraction of a Langevin step is out^T[unit][chain] = W[unit][k] . act^T[k][chain] in fp32.  Rounds 3-4 emulated the fp32 product
// with three bf16 pieces per operand and six v_mfma_f32_16x16x32_bf16 per 32-deep block ("bf16x6": 24 significant bits per operand).  This
// core uses TWO fp16 pieces per operand (11 + 11 = 22 significant bits) and THREE v_mfma_f32_16x16x32_f16 (four in GEMMs with K <= 64) --
// same issue rate per instruction, half the instructions, two fragment planes instead of three (4 B per weight instead of 6 through the
// CU's vector-memory path) and a split of 24 instead of 47 VALU instructions per 8 values:
//      a b  ~  [a_m b_m] + (a_m b_h + a_h b_m) + a_h b_h,        a = (a_h + a_m) 2^-sa,  b = (b_h + b_m) 2^-sb.
// fp16 has 5 exponent bits, so both operands are SCALED BY POWERS OF TWO (exact) before the split: the weights of a Linear by one
// exponent chosen when they are packed (max |W| -> [2^14, 2^15)), the B operand PER CHAIN ROW by an exponent taken from the row's own
// maximum (rows of one chain never share a scale with another chain's: chains stay independent, and a diverged chain cannot cost its
// neighbours precision) -- by a pre-pass of the GEMM wave over the row (gemm_row_exp), or, in the in-place kernel's lean path, from a word
// the epilogue waves that WROTE the row keep (mcpc_kernels.h: rowexp_track; the caller passes it as GemmScale::fixed_b): the same
// exponent either way, so the kernel forms agree bitwise, without 2.8 us of row scans on the step's critical path.  The fp32 accumulators are un-scaled by 2^-(sa + sb) (exact)
// when the GEMM ends.  Against an fp64 dot product, relative to sum |terms|, this is in the class of the fp32 MFMA chain and of bf16x6 on
// every operand distribution tried (K = 32 / 256 / 784; activations, errors, rows of mixed scale, heavy tails: max 0.6-1.8e-7, rms
// 0.9-3.6e-8 -- scripts/f16x4_study.py): what limits all three is the fp32 accumulation, not the 22 or 24 operand bits,
// whose rounding errors average out over K.  A value more than 2^16 below its row's maximum loses relative precision (its second piece
// becomes an fp16 subnormal); its absolute error stays below 2^-39 of the row's maximum.
//
//   A operand (weights): split at pack time (mcpc_pack_kernel) into two fp16 planes in MFMA fragment order: for output tile ut and
//     32-deep block kb, plane p, lane (m = lane & 15, g = lane >> 4) holds W[16 ut + m][32 kb + 8 g .. + 7] 2^sa as 8 fp16 = 16 B:
//     one block of one tile = 2 x 1 KiB contiguous, two global_load_dwordx4 per wave.
//   B operand (activations / errors): stays fp32 in LDS rows [chain][k]; lane (c, g) reads k = 32 kb + 8 g .. + 7 as two ds_read_b128,
//     multiplies by its chain's 2^sb and splits (v_pk_mul_f32, v_cvt_pk_f16_f32, 2 v_cvt_f32_f16, v_pk_fma_f32, v_cvt_pk_f16_f32 per pair).
//   C: the fp32 accumulator tile of the 16x16 MFMAs (lane (c, q): units 4q..4q+3 of chain c) -- the lane's chain is the one whose
//     row it scaled, so the un-scaling factor is a per-lane scalar.
// The read-out's back-projection accumulates over SEVERAL GEMMs (one per chunk of read-out units) in the same accumulators: they stay
// in scaled units between the chunks (GemmScale::run: the exponent they are in); a chunk whose row needs a smaller exponent rescales
// them (exact), and they are un-scaled when the sum is handed over.  A Bernoulli read-out's error is bounded (|sigmoid(o) - y| <= 1 for
// targets in [0, 1]): its rows take a FIXED exponent, so that the sum does not depend on how the read-out is cut into chunks (the two
// step kernels cut it differently and must agree bitwise: bench.py self_check); an unbounded read-out error (Gaussian) takes the
// exponent of each chunk's maximum and the kernels agree to rounding.
// k ranges that are not a multiple of 32 (kw % 32 == 16): the weights beyond kw are zeros AND the B lanes beyond kw read ZEROS -- in the
// last k-block the lanes whose eight k values lie beyond kw (g >= 2) take their address from a 16-float region of the plan that is
// zero-filled at launch and never written (KParams::lds_zero) instead of from behind their row -- so the excess products are exact
// zeros WHATEVER the LDS holds behind the row (round 3's NaN: DESIGN section 8, tests/test_gpu_lds_poison.py).
#pragma once

namespace mcpc {

constexpr int kKB = 32;                 // k-depth of one fragment block
constexpr int kFragBlock = 2 * 64;      // u32x4 units per (tile, k-block): two fp16 planes of 64 lanes
struct frag_t { u32x4 h, m; };          // one k-block of one tile as seen by a lane

__device__ __forceinline__ frag_t frag_zero() { frag_t f; f.h = f.m = u32x4{0u, 0u, 0u, 0u}; return f; }
__device__ __forceinline__ frag_t load_frag(const gu32x4* A, int off, int lane) {
    frag_t f;
    f.h = A[off + lane]; f.m = A[off + 64 + lane];
    return f;
}

// x 2^sb = h + m for a pair of values (h, m fp16; exact to 22 significant bits while m stays a normal fp16)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2_pair(f32x2 x, float s, unsigned& h, unsigned& m) {
    const f32x2 xs = x * s;                                             // exact: s is a power of two
    const f16x2 hh = __builtin_convertvector(xs, f16x2);                // v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2 r = xs - __builtin_convertvector(hh, f32x2);            // exact: the residual has at most 13 significant bits
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ frag_t split8(f32x4 x0, f32x4 x1, float s) {
    unsigned h[4], m[4];
    split2_pair(f32x2{x0.x, x0.y}, s, h[0], m[0]);
    split2_pair(f32x2{x0.z, x0.w}, s, h[1], m[1]);
    split2_pair(f32x2{x1.x, x1.y}, s, h[2], m[2]);
    split2_pair(f32x2{x1.z, x1.w}, s, h[3], m[3]);
    frag_t f;
    f.h = u32x4{h[0], h[1], h[2], h[3]}; f.m = u32x4{m[0], m[1], m[2], m[3]};
    return f;
}
// the B planes of one k-block as a lane reads them: split here from fp32 (x 2^sb = h + m), or -- PS, unified-wave kernel -- already split by
// the operand's producer (mcpc_ws2_lean.h: lean_headf writes a Bernoulli read-out's error that way: the same split2_pair with the same
// exponent, once per value instead of once per k-block, wave and GEMM): the 32 bytes then hold [h0..h7][m0..m7]
template <bool PS>
__device__ __forceinline__ frag_t b_planes(f32x4 x0, f32x4 x1, float s) {
    if constexpr (PS) { frag_t f; f.h = __builtin_bit_cast(u32x4, x0); f.m = __builtin_bit_cast(u32x4, x1); (void)s; return f; }
    else return split8(x0, x1, s);
}
__device__ __forceinline__ f32x4 mfma4(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// ---- scaling ---------------------------------------------------------------------------------------------------------------------------
constexpr int kScaleTop = 141;          // a maximum with biased fp32 exponent e is scaled by 2^(141 - e): into [2^14, 2^15) (fp16: < 65 504)
constexpr int kScaleClamp = 60;         // |exponent| of any scale: 2^-(sa + sb) stays a normal fp32
// exponent that brings |v| <= mx into fp16's range with its two pieces normal for everything within 2^16 of mx
__device__ __host__ __forceinline__ int scale_exp_for_field(int biased_exponent) {
    int e = kScaleTop - biased_exponent;
    e = e > kScaleClamp ? kScaleClamp : (e < -kScaleClamp ? -kScaleClamp : e);
    return e;
}
__device__ __host__ __forceinline__ int scale_exp_for_max(float mx) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned bits = __float_as_uint(mx);
#else
    unsigned bits; __builtin_memcpy(&bits, &mx, 4);
#endif
    return scale_exp_for_field((int)((bits >> 23) & 0xffu));
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }      // |e| <= 126

// A GEMM's scaling state as its caller sees it.
//   a_exp     exponent the packed weights were scaled by (per Linear, written by the pack kernels);
//   mode      GS_FRESH: the accumulators start at zero and come back UN-SCALED;  GS_ACCUM: they carry the sum of earlier GEMMs in units of
//             2^(a_exp + run) and stay scaled (the caller un-scales with gemm_unscale when the sum is complete);
//   scan      (wave-uniform) 1: the B rows' exponents come from a pre-pass over the rows (gemm_row_exp);  0: from fixed_b;
//   fixed_b   scan == 0: the exponent of this lane's B row -- a constant for a bounded operand, or the word the row's producers keep (then
//             a per-lane value that may still be in flight from LDS: nothing branches on it before the first B block has been requested);
//   run       GS_ACCUM: the B exponent the accumulators are in (kRunNone before the first chunk); per lane = per chain.
enum : int { GS_FRESH = 0, GS_ACCUM = 1 };
constexpr int kScaleAuto = -1000, kRunNone = 1000;
//   short_k   1: the whole contraction has K <= 64 and keeps the a_m b_m term (gemm_fixed: MM); 0: it does not; -1: decided from this GEMM's own
//             k-blocks.  A sum over several GEMMs (GS_ACCUM) must be told: its chunks may be short where the contraction is long, and how
//             the contraction is cut into chunks must not change its arithmetic.
struct GemmScale { int a_exp; int mode; int fixed_b; int run; int short_k; int scan; };

#ifdef MCPC_EXP_NOLOAD   // timing experiment only (wrong results): every fragment load re-reads k-block 0 -> L1 hits
#define MCPC_KSEL(k_) 0
#else
#define MCPC_KSEL(k_) (k_)
#endif

#ifdef MCPC_EXP_NOSPLIT    // timing experiment only (wrong results): the B planes of block 0 serve every block (no LDS reads, no split)
#define MCPC_EXP_SPLIT8(a_, b_) bs[0]
#else
#define MCPC_EXP_SPLIT8(a_, b_) b_planes<PS>(a_, b_, bscale)
#endif
// acc += W-tiles . B over nkb blocks.  On entry `pre` holds block 0 of every tile, requested by the caller's prefetch (one table
// entry early); on return it is free.
//
// Register diet.  The wave's tiles are worked in two GROUPS of at most two (G0 = tiles 0, 1; G1 = tiles 2, 3); one k-block is two
// sub-steps (k, G0), (k, G1) of 4 x 2 MFMAs each, and the fragments of sub-step u + 2 travel while u and u + 1 compute -- the distance
// of a whole k-block -- in THREE rotating half-sets of 16 VGPRs: sub-step u reads set u mod 3, the request for u + 2 goes into the set
// u - 1 has just freed.  Static register names need the rotation unrolled over three k-blocks.
// (Two chain tiles per fragment -- CTT = 2 -- were measured in rounds 4 and 5 and dropped: profiles/r04_k1_bounds.txt, r05_k1_decomp.txt.)
//
// b_exp: 2^b_exp scales this lane's chain row (gemm_row_exp below, or the caller's fixed exponent).
// MM: with the a_m b_m term (2^-22 of the leading one).  GEMMs of at most kShortK k-blocks (K <= 64) keep it: there are too few terms there
// for the rounding errors of the 22-bit operands to average out (K = 32: max 1.6e-7 / rms 3.1e-8 of sum |terms| without it against 1.2e-7 /
// 2.6e-8 with it; fp32 MFMA chain 1.0e-7 / 2.1e-8).  From K = 96 on the term changes neither figure (profiles/r05_f16x4_study.txt) and longer
// GEMMs run THREE MFMAs per product.
constexpr int kShortK = 2;
template <int NT, int NTT, int CTT, int NW, bool MM, bool PS = false>
__device__ __forceinline__ void gemm_fixed(f32x4 (&acc)[NTT][CTT], const gu32x4* __restrict__ A, const int (&aoff)[NTT], int nkb, int kw,
                                           const float* B, int ldb, int lane, frag_t (&pre)[NTT], const float* zeros, int b_exp) {
    static_assert(CTT == 1, "one chain tile per workgroup (two were measured and dropped in rounds 4 and 5: DESIGN.md section 4)");
    constexpr int N0 = NT < 2 ? NT : 2, N1 = NT - N0;            // tiles of group 0 / group 1
    if (nkb <= 0) return;
    const int c = lane & 15, g = lane >> 4;
    const float* bp = B + c * ldb + 8 * g;
    // the LAST block: lanes whose eight k values lie beyond kw (g >= 2 when kw % 32 == 16) read the plan's zero region instead
#ifdef MCPC_EXP_NOTAILMASK      // timing experiment only (results depend on foreign LDS again)
    const float* const bp_last = bp + (nkb - 1) * kKB; (void)kw; (void)zeros;
#else
    const float* const bp_last = (uint32_t)(kKB * (nkb - 1) + 8 * g) < (uint32_t)kw ? bp + (nkb - 1) * kKB : zeros;
#endif
    // wave-uniform base + a 32-bit per-lane byte offset that never changes during the GEMM (no VALU address arithmetic per load)
    uint32_t voff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) voff[t] = (uint32_t)(aoff[t] + lane) * 16u;
    const char __attribute__((address_space(1)))* const Ab = (const char __attribute__((address_space(1)))*)A;
    f32x4 bC[CTT][2];
    frag_t bs[CTT];
    frag_t s0[2], s1[2], s2[2];                                   // the three half-sets
    const int last = nkb - 1;
#define MCPC_LOAD_B(k_)                                                                             \
    do {                                                                                            \
        const float* const src_ = (k_) < last ? bp + (k_) * kKB : bp_last;     /* never conditional */ \
        bC[0][0] = *reinterpret_cast<const f32x4*>(src_);                                           \
        bC[0][1] = *reinterpret_cast<const f32x4*>(src_ + 4);                                       \
    } while (0)
    // fragments of group G_ (0 / 1) of block k_ (clamped) into half-set s_
#define MCPC_LOAD_HALF(s_, G_, k_)                                                                  \
    do {                                                                                            \
        const int kc_ = (k_) < last ? (k_) : last;                                                  \
        const char __attribute__((address_space(1)))* const Ak_ = Ab + (size_t)MCPC_KSEL(kc_) * (kFragBlock * 16u); \
        _Pragma("unroll") for (int i = 0; i < ((G_) ? N1 : N0); ++i) {                              \
            s_[i].h = *(const gu32x4*)(Ak_ + voff[2 * (G_) + i]);                                   \
            s_[i].m = *(const gu32x4*)(Ak_ + voff[2 * (G_) + i] + 1024u);                           \
        }                                                                                           \
    } while (0)
    // the four products of group G_ x chain tile ct_ out of half-set s_, small terms first; consecutive MFMAs alternate between the
    // group's (at most two) accumulators of that chain tile
#define MCPC_M4(s_, G_, ct_, ap_, bp_)                                                              \
    _Pragma("unroll") for (int i = 0; i < ((G_) ? N1 : N0); ++i)                                    \
        acc[2 * (G_) + i][ct_] = mfma4(s_[i].ap_, bs[ct_].bp_, acc[2 * (G_) + i][ct_])
#ifdef MCPC_EXP_HALFMFMA   // timing experiment only (wrong results): two of the four products
#define MCPC_SUB(s_, G_, ct_)                                                                       \
    do { MCPC_M4(s_, G_, ct_, m, h); MCPC_M4(s_, G_, ct_, h, h); } while (0)
#else
#define MCPC_SUB(s_, G_, ct_)                                                                       \
    do { if constexpr (MM) { MCPC_M4(s_, G_, ct_, m, m); }                                           \
         MCPC_M4(s_, G_, ct_, m, h); MCPC_M4(s_, G_, ct_, h, m); MCPC_M4(s_, G_, ct_, h, h); } while (0)
#endif
#define MCPC_SPLIT1(ct_) bs[ct_] = b_planes<PS>(bC[ct_][0], bC[ct_][1], bscale)
    // One k-block: the chain tile's planes are read by both sub-steps, so the next block's split goes into a second copy beside
    // (k, G1) and is moved over at the end of the block (8 v_mov)
#define MCPC_BLOCK1(sa_, sb_, sc_, k_)                                                              \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        MCPC_LOAD_HALF(sc_, 0, (k_) + 1);                                                           \
        MCPC_SUB(sa_, 0, 0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        MCPC_LOAD_HALF(sa_, 1, (k_) + 1);                                                           \
        const frag_t bsn_ = MCPC_EXP_SPLIT8(bC[0][0], bC[0][1]);   /* (k + 1): read at the head of this block */ \
        MCPC_SUB(sb_, 1, 0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        MCPC_LOAD_B((k_) + 2);                                                                      \
        bs[0] = bsn_;                                                                               \
    } while (0)
#define MCPC_BLOCK(sa_, sb_, sc_, k_) MCPC_BLOCK1(sa_, sb_, sc_, k_)
    // The LAST block of a GEMM requests, reads and splits nothing: there is no next block.  (Until round 5 every block ran MCPC_BLOCK1 with
    // its requests clamped to the last block: 8 fragment loads, 2 LDS reads and a 24-instruction split per GEMM for nothing -- ~400 of the
    // ~2000 cycles a table entry costs a GEMM wave beyond its k-blocks, 13 entries per step.)
#define MCPC_BLOCK_LAST(sa_, sb_)                                                                   \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        MCPC_SUB(sa_, 0, 0);                                                                        \
        MCPC_SUB(sb_, 1, 0);                                                                        \
    } while (0)
    // block 0 arrives in `pre`: tiles 0, 1 -> set 0, tiles 2, 3 -> set 1
#pragma unroll
    for (int i = 0; i < N0; ++i) s0[i] = pre[i];
#pragma unroll
    for (int i = 0; i < N1; ++i) s1[i] = pre[2 + i];
    MCPC_LOAD_B(0);
    // (the exponent may still be on its way from LDS -- the row word of the in-place kernel -- : first used here, behind the B reads)
    const float bscale = pow2i(b_exp);
    MCPC_SPLIT1(0);
    MCPC_LOAD_B(1);
    int k = 0;
    // steady state: three k-blocks per round (the rotation's period) while a block FOLLOWS the round; then up to two more full blocks and
    // the last one, which prefetches nothing (the LDS read of k + 2 in the block before it is clamped to the last block: harmless)
    if constexpr (!MM) {                    // (MM: at most kShortK = 2 blocks, the tail below)
        for (; k + 3 < nkb; k += 3) {
            MCPC_BLOCK(s0, s1, s2, k);          // (k, G0) = s0, (k, G1) = s1;  s2 <- (k+1, G0), s0 <- (k+1, G1)
            MCPC_BLOCK(s2, s0, s1, k + 1);      // s1 <- (k+2, G0), s2 <- (k+2, G1)
            MCPC_BLOCK(s1, s2, s0, k + 2);      // s0 <- (k+3, G0), s1 <- (k+3, G1): the round's starting assignment again
        }
    }
    const int rem = nkb - k;                // 1 .. 3 blocks left (MM: 1 .. 2)
    if (rem == 3) {
        MCPC_BLOCK(s0, s1, s2, k);
        MCPC_BLOCK(s2, s0, s1, k + 1);
        MCPC_BLOCK_LAST(s1, s2);
    } else if (rem == 2) {
        MCPC_BLOCK(s0, s1, s2, k);
        MCPC_BLOCK_LAST(s2, s0);
    } else {
        MCPC_BLOCK_LAST(s0, s1);
    }
#undef MCPC_BLOCK
#undef MCPC_BLOCK_LAST
#undef MCPC_BLOCK1
#undef MCPC_SPLIT1
#undef MCPC_SUB
#undef MCPC_M4
#undef MCPC_LOAD_HALF
#undef MCPC_LOAD_B
}

// ---- few tiles per wave: a deeper fragment stream (unified-wave kernel, mcpc_steps_u.h) --------------------------------------------------
// With one or two tiles per wave the rotation above keeps 2 or 4 KiB of fragments in flight per wave -- one k-block ahead of 3 or 6 MFMAs
// (48 / 96 cycles) -- and every k-block waits out an L2 round trip: ~750 cycles per block in the unified-wave kernel's 1-tile GEMMs
// (profiles/r06_small_net.txt).  This form keeps W = 4 / NT k-blocks of the wave's NT tiles in flight (the same four fragment slots the
// caller's prefetch fills, one table entry early: slot j NT + i = block j of tile i): GEMMs of at most W blocks find ALL their fragments in
// registers, longer ones refill a slot as soon as its MFMAs are issued.  Per accumulator the MFMAs are those of gemm_fixed, in its order
// ([a_m b_m,] a_m b_h, a_h b_m, a_h b_h per block, blocks ascending): bitwise the same sums.
template <int NT, int NTT, bool MM, bool PS = false>
__device__ __forceinline__ void gemm_deep(f32x4 (&acc)[NTT][1], const gu32x4* __restrict__ A, const int (&aoff)[NTT], int nkb, int kw,
                                          const float* B, int ldb, int lane, frag_t (&pre)[NTT], const float* zeros, int b_exp) {
    static_assert(NT == 1 || NT == 2, "one or two tiles");
    static_assert(NTT == 4, "four fragment slots");
    constexpr int W = 4 / NT;                                     // k-blocks in flight
    if (nkb <= 0) return;
    const int c = lane & 15, g = lane >> 4;
    const float* bp = B + c * ldb + 8 * g;
    const float* const bp_last = (uint32_t)(kKB * (nkb - 1) + 8 * g) < (uint32_t)kw ? bp + (nkb - 1) * kKB : zeros;
    uint32_t voff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) voff[t] = (uint32_t)(aoff[t] + lane) * 16u;
    const char __attribute__((address_space(1)))* const Ab = (const char __attribute__((address_space(1)))*)A;
    const int last = nkb - 1;
    frag_t f[W][NT];
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) f[j][i] = pre[j * NT + i];
    f32x4 b0, b1;
    frag_t bs;
#define MCPC_DLOAD_B(k_)                                                                            \
    do {                                                                                            \
        const float* const src_ = (k_) < last ? bp + (k_) * kKB : bp_last;                          \
        b0 = *reinterpret_cast<const f32x4*>(src_);                                                 \
        b1 = *reinterpret_cast<const f32x4*>(src_ + 4);                                             \
    } while (0)
#define MCPC_DM(j_, ap_, bp_)                                                                       \
    _Pragma("unroll") for (int i = 0; i < NT; ++i) acc[i][0] = mfma4(f[j_][i].ap_, bs.bp_, acc[i][0])
#define MCPC_DSUB(j_)                                                                               \
    do { if constexpr (MM) { MCPC_DM(j_, m, m); }                                                   \
         MCPC_DM(j_, m, h); MCPC_DM(j_, h, m); MCPC_DM(j_, h, h); } while (0)
#define MCPC_DREFILL(j_, k_)                                                                        \
    do {                                                                                            \
        const char __attribute__((address_space(1)))* const Ak_ = Ab + (size_t)MCPC_KSEL(k_) * (kFragBlock * 16u); \
        _Pragma("unroll") for (int i = 0; i < NT; ++i) {                                            \
            f[j_][i].h = *(const gu32x4*)(Ak_ + voff[i]);                                           \
            f[j_][i].m = *(const gu32x4*)(Ak_ + voff[i] + 1024u);                                   \
        }                                                                                           \
    } while (0)
    MCPC_DLOAD_B(0);
    const float bscale = pow2i(b_exp);
    bs = b_planes<PS>(b0, b1, bscale);
    MCPC_DLOAD_B(1);
    int k = 0;
    // steady state: W blocks per round, every slot refilled with the block W ahead
    for (; k + 2 * W <= nkb; k += W) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            MCPC_DSUB(j);
            MCPC_DREFILL(j, k + j + W);
            const frag_t bsn = b_planes<PS>(b0, b1, bscale);       // block k + j + 1
            MCPC_DLOAD_B(k + j + 2);
            bs = bsn;
        }
    }
    // tail: the up to 2 W - 1 blocks left; refills while a block W ahead exists
#pragma unroll
    for (int j = 0; j < 2 * W - 1; ++j) {
        if (k + j < nkb) {
            __builtin_amdgcn_sched_barrier(0);
            MCPC_DSUB(j % W);
            if (k + j + W < nkb) MCPC_DREFILL(j % W, k + j + W);
            if (k + j + 1 < nkb) {
                const frag_t bsn = b_planes<PS>(b0, b1, bscale);
                MCPC_DLOAD_B(k + j + 2);
                bs = bsn;
            }
        }
    }
#undef MCPC_DREFILL
#undef MCPC_DSUB
#undef MCPC_DM
#undef MCPC_DLOAD_B
}

// Exponent for this lane's chain row of the B operand: 2^sb brings the row's largest |value| into [2^14, 2^15).  Lane (c, g) scans the
// k values it will read in the GEMM (8 per 32-deep block: the four lanes of a chain cover the row between them), the four partial maxima
// meet through two cross-row permutes.  The last block's over-read lanes read the plan's zero region, as in the GEMM.  An all-zero row
// (a ReLU layer that is off, a padding chain) and a row that holds Inf / NaN (a diverged chain: max is Inf, or NaN is skipped by v_max) get
// clamped exponents; their products are zeros resp. Inf / NaN for THAT chain only.
__device__ __forceinline__ int gemm_row_exp(const float* B, int ldb, int nkb, int kw, int lane, const float* zeros) {
    const int c = lane & 15, g = lane >> 4;
    const float* bp = B + c * ldb + 8 * g;
    const float* const bp_last = (uint32_t)(kKB * (nkb - 1) + 8 * g) < (uint32_t)kw ? bp + (nkb - 1) * kKB : zeros;
    const int last = nkb - 1;
    float mx = 0.f;
#pragma unroll 2
    for (int kb = 0; kb < nkb; ++kb) {
        const float* const src = kb < last ? bp + kb * kKB : bp_last;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    return scale_exp_for_max(mx);
}

// un-scale accumulators that are in units of 2^(a_exp + b_exp) (exact)
template <int NTT, int CTT>
__device__ __forceinline__ void gemm_unscale(f32x4 (&acc)[NTT][CTT], int a_exp, int b_exp) {
    const float un = pow2i(-a_exp) * pow2i(-b_exp);
#pragma unroll
    for (int i = 0; i < NTT; ++i)
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) acc[i][ct] = acc[i][ct] * un;
}

// nt (wave-uniform, 1..NTT) selects a straight-line instantiation: no per-tile branches in the loop
template <int N, int NTT, int CTT, int NW, bool MM>
__device__ __forceinline__ void gemm_dispatch(f32x4 (&acc)[NTT][CTT], const gu32x4* __restrict__ A, const int (&aoff)[NTT], int nt, int nkb, int kw,
                                              const float* B, int ldb, int lane, frag_t (&pre0)[NTT], const float* zeros, int b_exp) {
    if constexpr (N >= NTT) {
        gemm_fixed<NTT, NTT, CTT, NW, MM>(acc, A, aoff, nkb, kw, B, ldb, lane, pre0, zeros, b_exp);
    } else {
        if (nt == N) gemm_fixed<N, NTT, CTT, NW, MM>(acc, A, aoff, nkb, kw, B, ldb, lane, pre0, zeros, b_exp);
        else gemm_dispatch<N + 1, NTT, CTT, NW, MM>(acc, A, aoff, nt, nkb, kw, B, ldb, lane, pre0, zeros, b_exp);
    }
}
// kw: valid k width of the B rows (a multiple of 16, 32 (nkb - 1) < kw <= 32 nkb); zeros: 16 floats of LDS that stay zero for the launch.
// gs: see GemmScale.  GS_FRESH: acc must be zero on entry and holds the true sums on return.  GS_ACCUM: acc carries earlier chunks in units
// of 2^(a_exp + gs.run) and stays scaled; gs.run is updated (a smaller exponent rescales acc first: exact).
template <int NTT, int CTT, int NW>
__device__ __forceinline__ void gemm_tiles(f32x4 (&acc)[NTT][CTT], const void* A, const int (&aoff)[NTT], int nt, int nkb, int kw,
                                           const float* B, int ldb, int lane, frag_t (&pre0)[NTT], const float* zeros, GemmScale& gs) {
#ifdef MCPC_EXP_NOROWEXP    // timing experiment only (wrong results): no pre-pass over the row, a constant exponent
    int b_exp = gs.scan ? 8 : gs.fixed_b;
#else
    int b_exp = gs.scan ? gemm_row_exp(B, ldb, nkb, kw, lane, zeros) : gs.fixed_b;
#endif
    if (gs.mode == GS_ACCUM) {
        if (gs.run != kRunNone && b_exp != gs.run) {
            if (b_exp < gs.run) {                       // this chunk's row is larger than any before: bring the sum down to its units
                const float f = pow2i(b_exp - gs.run);
#pragma unroll
                for (int i = 0; i < NTT; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTT; ++ct) acc[i][ct] = acc[i][ct] * f;
            } else {
                b_exp = gs.run;                         // a smaller row joins in the units the sum is already in
            }
        }
        gs.run = b_exp;
    }
    const bool mm = gs.short_k >= 0 ? gs.short_k != 0 : nkb <= kShortK;
    if (mm && nkb <= kShortK) gemm_dispatch<1, NTT, CTT, NW, true>(acc, (const gu32x4*)A, aoff, nt, nkb, kw, B, ldb, lane, pre0, zeros, b_exp);
    else gemm_dispatch<1, NTT, CTT, NW, false>(acc, (const gu32x4*)A, aoff, nt, nkb, kw, B, ldb, lane, pre0, zeros, b_exp);
    if (gs.mode == GS_FRESH) gemm_unscale<NTT, CTT>(acc, gs.a_exp, b_exp);
}

// GEMM of the unified-wave kernel: fresh accumulators, the row's exponent from its producers' word (or a constant), un-scaled on return;
// one or two tiles on the deep fragment stream (gemm_deep: `pre` = four (block, tile) slots), three or four on the rotation of gemm_fixed
// (`pre` = block 0 of every tile).  ps (wave-uniform): the B operand is already in planes (the Bernoulli read-out's error: headf_planes).
template <int NTT>
__device__ __forceinline__ void gemm_tiles_u(f32x4 (&acc)[NTT][1], const void* A, const int (&aoff)[NTT], int nt, int nkb, int kw,
                                             const float* B, int ldb, int lane, frag_t (&pre)[NTT], const float* zeros, int a_exp, int b_exp,
                                             int short_k, bool ps) {
    const bool mm = (short_k >= 0 ? short_k != 0 : nkb <= kShortK) && nkb <= kShortK;
    const gu32x4* const Ag = (const gu32x4*)A;
    if (ps) {                       // (long contractions only: never with the fourth product)
        if (nt == 1) gemm_deep<1, NTT, false, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
        else if (nt == 2) gemm_deep<2, NTT, false, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
        else if (nt == 3) gemm_fixed<3, NTT, 1, 8, false, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
        else gemm_fixed<4, NTT, 1, 8, false, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
    } else if (nt == 1) {
        if (mm) gemm_deep<1, NTT, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
        else gemm_deep<1, NTT, false>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
    } else if (nt == 2) {
        if (mm) gemm_deep<2, NTT, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
        else gemm_deep<2, NTT, false>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
    } else if (nt == 3) {
        if (mm) gemm_fixed<3, NTT, 1, 8, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
        else gemm_fixed<3, NTT, 1, 8, false>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
    } else {
        if (mm) gemm_fixed<4, NTT, 1, 8, true>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
        else gemm_fixed<4, NTT, 1, 8, false>(acc, Ag, aoff, nkb, kw, B, ldb, lane, pre, zeros, b_exp);
    }
    gemm_unscale<NTT, 1>(acc, a_exp, b_exp);
}

// request the fragments of k-block 0 of a phase's GEMM (issued one phase early: weights do not depend on any barrier)
template <int NW, int NTW>
__device__ __forceinline__ void prefetch_first_blocks(const KPhase& ph, int wave, int lane, int& nt, int (&aoff)[NTW],
                                                      frag_t (&pre0)[NTW]) {
    nt = (ph.ntiles - wave + NW - 1) / NW;
    nt = nt < 0 ? 0 : (nt > NTW ? NTW : nt);
    if (ph.type == PH_ENERGY) nt = 0;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        aoff[i] = (ph.tile0 + wave + NW * i) * ph.a_tile_stride + ph.a_off0;
        if (i < nt && ph.nkb > 0) {
            const gu32x4* A = (const gu32x4*)ph.A;      // weights live in global memory: global_load, not flat_load
            pre0[i] = load_frag(A, aoff[i], lane);
        }
    }
}

}  // namespace mcpc
