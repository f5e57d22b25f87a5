// GEMM core of the step kernels, "bf16x6" form: fp32 products on the bf16 matrix pipe (mcpc_bf16x6.h).
//
// Every contraction of a Langevin step is out^T[unit][chain] = W[unit][k] . act^T[k][chain] in fp32.  On gfx950 the fp32 MFMA
// (v_mfma_f32_16x16x4_f32) runs at the vector rate, 157 TFLOP/s; the bf16 MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate) at 16
// times that.  An fp32 number is the sum of three bf16 numbers, x = hi + mid + lo (8 + 8 + 8 significant bits, round to nearest even
// at every cut), and a product of two of them is the sum of nine bf16 x bf16 products, each EXACT in fp32.  Keeping the six whose
// magnitude is above 2^-24 of the leading one,
//      a b  ~  a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid),
// and accumulating them in fp32 small-terms-first gives a dot product whose error against fp64 is that of the fp32 MFMA chain
// (measured on this kernel's shapes: max 2.7e-7 / rms 3.1e-8 of sum|terms| against 2.1e-7 / 2.7e-8, scripts/heb_bf16_ubench.hip;
// tests/test_gpu_engine.py::test_gemm_core_accuracy_against_fp64) at 6 x 16 cycles per 32-deep block of a tile instead of 8 x 32.
//
//   A operand (weights): pre-split at pack time (mcpc_pack_kernel) into three bf16 planes in MFMA fragment order: for output tile ut
//     and 32-deep block kb, plane p, lane (m = lane & 15, g = lane >> 4) holds W[16 ut + m][32 kb + 8 g .. + 7] as 8 bf16 = 16 B:
//     one block of one tile = 3 x 1 KiB contiguous, three global_load_dwordx4 per wave.
//   B operand (activations / errors): stays fp32 in LDS rows [chain][k] exactly as before -- the LDS plan, the epilogues and the
//     in-place hand-over are untouched -- lane (c, g) reads k = 32 kb + 8 g .. + 7 as two ds_read_b128 and splits them itself
//     (11 VALU instructions per pair of values, ~90 per block and chain-tile pair, issued beside the MFMAs of the same wave).
//   C: the fp32 accumulator tile of the 16x16 MFMAs, same layout as before (lane (c, q): units 4q..4q+3 of chain c).
// Fragments of a k-block travel ONE block ahead of their MFMAs in two register sets (P, Q) of 12 VGPRs per tile; the sets double as
// the cross-entry prefetch (blocks 0 and 1 of the next table entry are requested into them while the current block is handed over).
// k ranges that are not a multiple of 32 (kw % 32 == 16): the weights beyond kw are zeros AND the B lanes beyond kw read ZEROS -- in the
// last k-block the lanes whose eight k values lie beyond kw (g >= 2) take their address from a 16-float region of the plan that is
// zero-filled at launch and never written (KParams::lds_zero) instead of from behind their row -- so the excess products are exact
// zeros WHATEVER the LDS holds behind the row: the next chain's row, another region of the plan, or bytes another kernel left behind.
// (One v_cndmask per B load.  Zeroing the loaded values instead, 8 v_and per load, cost 3 % at cfg-M and 7 % at 256 chains.)  (Round 3 relied on "finite neighbours"; a
// non-finite neighbour -- 0 x NaN -- was the intermittent NaN of tests/test_gpu_fuzz.py::test_wide_networks_against_oracle, DESIGN
// section 8, reproduced by scripts/nan_repro.py and pinned by tests/test_gpu_lds_poison.py.)
#pragma once

namespace mcpc {

constexpr int kKB = 32;                 // k-depth of one fragment block
constexpr int kFragBlock = 3 * 64;      // u32x4 units per (tile, k-block): three planes of 64 lanes
struct frag_t { u32x4 h, m, l; };       // one k-block of one tile as seen by a lane

__device__ __forceinline__ frag_t frag_zero() { frag_t f; f.h = f.m = f.l = u32x4{0u, 0u, 0u, 0u}; return f; }
__device__ __forceinline__ frag_t load_frag(const gu32x4* A, int off, int lane) {
    frag_t f;
    f.h = A[off + lane]; f.m = A[off + 64 + lane]; f.l = A[off + 128 + lane];
    return f;
}

__device__ __forceinline__ frag_t split8(f32x4 x0, f32x4 x1) {
    unsigned h[4], m[4], l[4];
    split3_pair_fast(f32x2{x0.x, x0.y}, h[0], m[0], l[0]);
    split3_pair_fast(f32x2{x0.z, x0.w}, h[1], m[1], l[1]);
    split3_pair_fast(f32x2{x1.x, x1.y}, h[2], m[2], l[2]);
    split3_pair_fast(f32x2{x1.z, x1.w}, h[3], m[3], l[3]);
    frag_t f;
    f.h = u32x4{h[0], h[1], h[2], h[3]}; f.m = u32x4{m[0], m[1], m[2], m[3]}; f.l = u32x4{l[0], l[1], l[2], l[3]};
    return f;
}

// the six products of one k-block for NT tiles x CTT chain tiles, small terms first; consecutive MFMAs go to different
// accumulators (an accumulator is touched again after NT * CTT - 1 others)
template <int NT, int NTT, int CTT>
__device__ __forceinline__ void mfma6_block(f32x4 (&acc)[NTT][CTT], const frag_t (&a)[NTT], const frag_t (&b)[CTT]) {
#define MCPC_M6(ap_, bp_)                                                                           \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                  \
        _Pragma("unroll") for (int ct = 0; ct < CTT; ++ct) acc[t][ct] = mfma6(a[t].ap_, b[ct].bp_, acc[t][ct])
    MCPC_M6(m, m);
    MCPC_M6(l, h);
    MCPC_M6(h, l);
    MCPC_M6(m, h);
    MCPC_M6(h, m);
    MCPC_M6(h, h);
#undef MCPC_M6
}

#ifdef MCPC_EXP_NOLOAD   // timing experiment only (wrong results): every fragment load re-reads k-block 0 -> L1 hits
#define MCPC_KSEL(k_) 0
#else
#define MCPC_KSEL(k_) (k_)
#endif

#ifdef MCPC_EXP_NOSPLIT    // timing experiment only (wrong results): the B planes of block 0 serve every block (no LDS reads, no split)
#define MCPC_EXP_SPLIT8(a_, b_) bs[0]
#else
#define MCPC_EXP_SPLIT8(a_, b_) split8(a_, b_)
#endif
// acc += W-tiles . B over nkb blocks.  On entry `pre` holds block 0 of every tile, requested by the caller's prefetch (one table
// entry early); on return it is free.
//
// Register diet.  The wave's tiles are worked in two GROUPS of at most two (G0 = tiles 0, 1; G1 = tiles 2, 3); one k-block is two
// sub-steps (k, G0), (k, G1) of 6 x 2 x CTT MFMAs each, and the fragments of sub-step u + 2 travel while u and u + 1 compute --
// the distance of a whole k-block, as with two full sets -- in THREE rotating half-sets of 24 VGPRs: sub-step u reads set u mod 3,
// the request for u + 2 goes into the set u - 1 has just freed.  72 VGPRs instead of 96 (the kernel allocates 243 of 256 with
// no scratch).  Static register names need the rotation unrolled over three k-blocks.
// (Round 4 measured the two-chain-tile form of this loop -- CTT = 2, every fragment serving 32 chains, without scratch -- and dropped
// it: 79 us per 32-chain step against 2 x 35.5, profiles/r04_k1_bounds.txt.)
template <int NT, int NTT, int CTT, int NW>
__device__ __forceinline__ void gemm_fixed(f32x4 (&acc)[NTT][CTT], const gu32x4* __restrict__ A, const int (&aoff)[NTT], int nkb, int kw,
                                           const float* B, int ldb, int lane, frag_t (&pre)[NTT], const float* zeros) {
    static_assert(CTT == 1, "one chain tile per workgroup (two were measured and dropped in round 4: DESIGN.md section 4)");
    constexpr int N0 = NT < 2 ? NT : 2, N1 = NT - N0;            // tiles of group 0 / group 1
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
            s_[i].l = *(const gu32x4*)(Ak_ + voff[2 * (G_) + i] + 2048u);                           \
        }                                                                                           \
    } while (0)
    // the six products of group G_ x chain tile ct_ out of half-set s_, small terms first; consecutive MFMAs alternate between the
    // group's (at most two) accumulators of that chain tile
#define MCPC_M6(s_, G_, ct_, ap_, bp_)                                                              \
    _Pragma("unroll") for (int i = 0; i < ((G_) ? N1 : N0); ++i)                                    \
        acc[2 * (G_) + i][ct_] = mfma6(s_[i].ap_, bs[ct_].bp_, acc[2 * (G_) + i][ct_])
#ifdef MCPC_EXP_HALFMFMA   // timing experiment only (wrong results): three of the six products
#define MCPC_SUB(s_, G_, ct_)                                                                       \
    do { MCPC_M6(s_, G_, ct_, m, m); MCPC_M6(s_, G_, ct_, l, h); MCPC_M6(s_, G_, ct_, h, l); } while (0)
#else
#define MCPC_SUB(s_, G_, ct_)                                                                       \
    do { MCPC_M6(s_, G_, ct_, m, m); MCPC_M6(s_, G_, ct_, l, h); MCPC_M6(s_, G_, ct_, h, l);         \
         MCPC_M6(s_, G_, ct_, m, h); MCPC_M6(s_, G_, ct_, h, m); MCPC_M6(s_, G_, ct_, h, h); } while (0)
#endif
#define MCPC_SPLIT1(ct_) bs[ct_] = split8(bC[ct_][0], bC[ct_][1])
    // One k-block: the chain tile's planes are read by both sub-steps, so the next block's split goes into a second copy beside
    // (k, G1) and is moved over at the end of the block (12 v_mov)
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
    // block 0 arrives in `pre`: tiles 0, 1 -> set 0, tiles 2, 3 -> set 1
#pragma unroll
    for (int i = 0; i < N0; ++i) s0[i] = pre[i];
#pragma unroll
    for (int i = 0; i < N1; ++i) s1[i] = pre[2 + i];
    MCPC_LOAD_B(0);
    MCPC_SPLIT1(0);
    MCPC_LOAD_B(1);
    int k = 0;
    // steady state: three k-blocks per round (the rotation's period); every request is for an existing block or clamped to the last
    for (; k + 3 <= nkb; k += 3) {
        MCPC_BLOCK(s0, s1, s2, k);          // (k, G0) = s0, (k, G1) = s1;  s2 <- (k+1, G0), s0 <- (k+1, G1)
        MCPC_BLOCK(s2, s0, s1, k + 1);      // s1 <- (k+2, G0), s2 <- (k+2, G1)
        MCPC_BLOCK(s1, s2, s0, k + 2);      // s0 <- (k+3, G0), s1 <- (k+3, G1): the round's starting assignment again
    }
    const int rem = nkb - k;
    if (rem == 2) {
        MCPC_BLOCK(s0, s1, s2, k);
        MCPC_BLOCK(s2, s0, s1, k + 1);
    } else if (rem == 1) {
        MCPC_BLOCK(s0, s1, s2, k);
    }
#undef MCPC_BLOCK
#undef MCPC_BLOCK1
#undef MCPC_SPLIT1
#undef MCPC_SUB
#undef MCPC_M6
#undef MCPC_LOAD_HALF
#undef MCPC_LOAD_B
}

// nt (wave-uniform, 1..NTT) selects a straight-line instantiation: no per-tile branches in the loop
template <int N, int NTT, int CTT, int NW>
__device__ __forceinline__ void gemm_dispatch(f32x4 (&acc)[NTT][CTT], const gu32x4* __restrict__ A, const int (&aoff)[NTT], int nt, int nkb, int kw,
                                              const float* B, int ldb, int lane, frag_t (&pre0)[NTT], const float* zeros) {
    if constexpr (N >= NTT) {
        gemm_fixed<NTT, NTT, CTT, NW>(acc, A, aoff, nkb, kw, B, ldb, lane, pre0, zeros);
    } else {
        if (nt == N) gemm_fixed<N, NTT, CTT, NW>(acc, A, aoff, nkb, kw, B, ldb, lane, pre0, zeros);
        else gemm_dispatch<N + 1, NTT, CTT, NW>(acc, A, aoff, nt, nkb, kw, B, ldb, lane, pre0, zeros);
    }
}
// kw: valid k width of the B rows (a multiple of 16, 32 (nkb - 1) < kw <= 32 nkb); zeros: 16 floats of LDS that stay zero for the launch
template <int NTT, int CTT, int NW>
__device__ __forceinline__ void gemm_tiles(f32x4 (&acc)[NTT][CTT], const void* A, const int (&aoff)[NTT], int nt, int nkb, int kw,
                                           const float* B, int ldb, int lane, frag_t (&pre0)[NTT], const float* zeros) {
    gemm_dispatch<1, NTT, CTT, NW>(acc, (const gu32x4*)A, aoff, nt, nkb, kw, B, ldb, lane, pre0, zeros);
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
