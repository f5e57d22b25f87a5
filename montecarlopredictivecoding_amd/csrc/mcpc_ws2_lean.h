// Lean epilogues of the in-place step kernel's E waves for the paths every script of the reference runs per step:
// SGD on x with or without the fused Philox kick, fused x update, workgroup fully inside the batch.
//
// Why they exist (measured, round 2): on gfx950 the fp32 MFMA and the VALU never co-execute
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for this kernel), so every VALU instruction an E wave issues is taken out of its
// partner's MFMA stream -- the step kernel's time is MFMA cycles + VALU cycles + what neither fills.  With the GEMMs
// switched off the E waves alone needed 35 us of a 94 us step.  The generic epilogues (mcpc_kernels.h) spend much of
// their instruction count on things these paths do not need:
//   * 64-bit per-lane addresses for every access   -> wave-uniform base (SGPR pair) + one 32-bit byte offset per lane:
//     a row offset per chain tile computed once per entry (v_mul_u32_u24), one v_add per tile;
//   * per-element `live` / padding / mask predicates -> padded units and partial loss masks are handled by a wave-uniform branch on
//     the one tile that has them; padding CHAINS (the last workgroup of a batch that is not a multiple of 16) cost one per-lane mask:
//     they evolve like any chain, and their energy terms are dropped by a select, their spills ANDed with 0, their records skipped
//     (LeanLane::livem) -- until round 3 such a workgroup took the generic epilogues and every launch waited for it (+17 %
//     per step at 7000 chains);
//   * three loads per slot whatever the entry type   -> each entry type requests exactly its operands.
// The arithmetic per element is the generic epilogue's, operation for operation: trajectories stay bitwise those of the
// other kernel forms and schedules (tests/test_gpu_fullsize.py, tests/test_gpu_rounds.py; bench.py self_check).
//
// XL (KParams::xl, 16-chain plans whose LDS has the room -- 45 KB more at cfg-M): the state rows x_l of the workgroup's chains, the
// biases, the mu_1 rows and the bit-packed target rows are copied to LDS when the launch starts (ws2_fill_fx / ws2_fill_constants);
// the epilogues read them there, the x update writes the new state there, and the rows go back to global memory once, after the
// step loop (lean_store_x).  An E wave then issues NO global load inside the step loop.  Why that matters: a wave's vector memory
// operations retire in order, so every operand load of an entry also waited for the Hebbian spill stores of the entry before it --
// 12.5 us of a 96 us step in the Hebbian stretches of cfg-M (timing build without the stores: 78.5 us against 88.5 per step of a
// learning call), and the x store / x loads of every step besides.
#pragma once

namespace mcpc {

typedef const char __attribute__((address_space(1)))* gbytes_t;
typedef char __attribute__((address_space(1)))* gbytes_w_t;

__device__ __forceinline__ f32x4 gld4(const float* base, uint32_t boff) {
    return *reinterpret_cast<const gf32x4*>((gbytes_t)base + boff);
}
__device__ __forceinline__ f32x4 gld4s(const float* base, uint32_t boff) {
    return __builtin_nontemporal_load(reinterpret_cast<const gf32x4*>((gbytes_t)base + boff));
}
__device__ __forceinline__ void gst4s(float* base, uint32_t boff, f32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<gf32x4*>((gbytes_w_t)base + boff));
}
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }

// Hebbian spill store through a buffer descriptor (wave-uniform base, 32-bit lane offset, selectable cache policy).  The data is
// read next by another kernel on other CUs, and 44 MB of it per step flow through L2s of 4 MB that hold the 3.3 MB of packed
// weights every GEMM wave streams.  Round 2 (epilogue loads still queued behind these stores in the wave's vector-memory order):
// plain write-back stores were 0.4 us per step faster than any other policy.  Round 3 (no global loads left in the epilogue waves,
// XL): system-scope stores -- sc1, with or without sc0 / nt -- take the learning call of cfg-M from 80.4-81.1 to 77.6-78.0 us
// per step, nt alone or sc0 nt do not; not issuing the stores at all: 71.7.  A shard of 256 chains, whose spill never leaves the
// L2, loses 6 % with them (24.4 -> 25.8 us per step of a learning call): `sys` (KParams::spill_sys, wave-uniform) chooses.
// Round 5 (the step kernel 1.3 x faster, the same stores per step): measured again, one gpurun call, learning call at 6000 / 4096 chains,
// us per step -- write-back 59.1-60.2 / 40.1-40.3, sc1 57.0-57.2 / 38.2-38.4, sc0 sc1 56.6-57.4 / 37.8, sc1 nt 55.9-56.2 / 37.8-38.2, sc0 sc1 nt
// (round 3's choice) 56.6 / 38.2-38.6, and a plain GLOBAL nontemporal store (no buffer descriptor) **55.1-55.5 / 37.3-37.4**: the default now.
#ifndef MCPC_SPILL_AUX
#define MCPC_SPILL_AUX -1          // < 0: global nontemporal store; else the gfx950 cache-policy bits of a buffer store: 1 = sc0, 2 = nt, 16 = sc1; 0 = plain write-back
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void spill_st4(float* base, uint32_t image_bytes, uint32_t boff, f32x4 v, bool sys = true) {
#ifdef MCPC_EXP_NOSPILL        // timing experiment only (wrong Hebbian sums): the spill stores are not issued
    (void)base; (void)image_bytes; (void)boff; (void)v; (void)sys;
#elif MCPC_SPILL_AUX < 0
    if (sys) {
        gst4s(base, boff, v); (void)image_bytes;
    } else {          // (a small shard's spill stays in the L2: plain write-back stores)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)image_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, (int)boff, 0, 0);
    }
#else
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)image_bytes, 0x00020000);
    if (sys) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, (int)boff, 0, MCPC_SPILL_AUX);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, (int)boff, 0, 0);
#endif
}

// per-lane constants of an E wave, fixed for the launch
template <int CTT>
struct LeanLane {
    int c, q;
    uint32_t chain[CTT];       // global row (chain) of this lane in chain tile ct
    uint32_t lrow[CTT];        // 16 ct + c: row inside the workgroup's LDS images
    uint32_t livem[CTT];       // ~0u / 0u
};
__device__ __forceinline__ f32x4 mask4(f32x4 v, uint32_t m) {
    f32x4 r;
    r.x = __uint_as_float(__float_as_uint(v.x) & m); r.y = __uint_as_float(__float_as_uint(v.y) & m);
    r.z = __uint_as_float(__float_as_uint(v.z) & m); r.w = __uint_as_float(__float_as_uint(v.w) & m);
    return r;
}

// tiles of this wave in an entry: tile(i) = tile0 + kk + NW i, i < nt
template <int ACT> __device__ __forceinline__ f32x4 act4(f32x4 x) {
    f32x4 r;
    r.x = actf<ACT>(x.x); r.y = actf<ACT>(x.y); r.z = actf<ACT>(x.z); r.w = actf<ACT>(x.w);
    return r;
}

// ---- FWD entry (layer l): e_l = c_l (x_l - mu_l), energies, E_l -> LDS, Hebbian spills ------------------------------
// mu_l = acc (from G, in LDS at out_lds) + bias for l >= 1; the constant mu_1 row for l == 0 (no GEMM).
// e0acc: Linear 0 sees a constant input, so only sum_t e_1 is needed for its Hebbian sums; when the top layer has at most
// one tile per wave (n_1 <= 64) the running sum is held in registers for the whole launch (lean_load_e0 / lean_flush_e0)
// instead of a global read-modify-write in every step.
// REG (unified-wave kernel, mcpc_steps_u.h): the wave computed the block itself -- it arrives in registers (racc[i][ct], zeros when the entry has
// no GEMM), nothing is waited for and nothing is read from out_lds.
template <int CTT, int NW, int NTW, int ACT, bool XL = false, bool REG = false>
__device__ __forceinline__ float lean_fwd(const KParams& P, const KPhase& ph, float* lds, int nt, int kk, const LeanLane<CTT>& L,
                                          int slot, int rec_idx, const int* prog_g, int need, int* err, int& dead,
                                          f32x4 (&e0acc)[CTT], bool e0_in_regs, float* rx = nullptr, unsigned row_gen = 0u,
                                          const f32x4 (*racc)[CTT] = nullptr) {
    if (nt <= 0) return 0.f;                  // (a layer with fewer tiles than waves: nothing to load, nothing to wait for)
    const int tstep = REG ? ph.rot : NW;      // tile(i) = tile0 + kk + tstep i: the unified-wave kernel's rows carry their own stride (kk = 0)
    const KLayer& Ly = P.layer[ph.layer];
    const int l = ph.layer;
    const bool has_gemm = (ph.flags & PHF_WS_GEMM) && ph.nkb > 0;
    const uint32_t npad4 = 4u * (uint32_t)Ly.npad;
    uint32_t rowb[CTT], lrowb[CTT], orowb[CTT];
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) {
        rowb[ct] = mul24(L.chain[ct], npad4) + 16u * L.q;
        lrowb[ct] = mul24(L.lrow[ct], 4u * (uint32_t)Ly.ld) + 16u * L.q;
        orowb[ct] = mul24(L.lrow[ct], 4u * (uint32_t)ph.out_ld) + 16u * L.q;
    }
    f32x4 xv[NTW][CTT], bv[NTW][CTT];
    const float* const bsrc = l == 0 ? P.mu1 : Ly.bias;                  // mu1: a row per chain; bias: one row
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const uint32_t tb = 64u * (uint32_t)(ph.tile0 + kk + tstep * (i < nt ? i : 0));     // unused slots repeat slot 0
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) {
            if constexpr (XL) {
                xv[i][ct] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + Ly.lds_x) + lrowb[ct] + tb);
                bv[i][ct] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + Ly.lds_bias) + (l == 0 ? lrowb[ct] : 16u * L.q) + tb);
                (void)bsrc;
            } else {
#if defined(MCPC_EXP_NOX) || defined(MCPC_EXP_NOELOAD)       // timing experiment only (wrong results): the state is neither loaded nor stored
            xv[i][ct] = splat(0.5f);
#else
            xv[i][ct] = gld4s(Ly.x, rowb[ct] + tb);
#endif
#ifdef MCPC_EXP_NOELOAD   // timing experiment only (wrong results): no operand loads at all in the E waves
            bv[i][ct] = splat(0.25f); (void)bsrc;
#else
            bv[i][ct] = gld4(bsrc, (l == 0 ? rowb[ct] : 16u * L.q) + tb);
#endif
            }
        }
    }
    f32x4 av[NTW][CTT];
    if constexpr (REG) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = racc[i][ct];
        (void)has_gemm; (void)prog_g; (void)need; (void)err; (void)dead; (void)orowb;
    } else if (has_gemm) {
        ws_wait_one(prog_g, need, err, dead);
        const char* const src = reinterpret_cast<const char*>(lds + ph.out_lds);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const uint32_t tb = 64u * (uint32_t)(ph.tile0 + kk + tstep * (i < nt ? i : 0));
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = *reinterpret_cast<const f32x4*>(src + orowb[ct] + tb);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = splat(0.f);
    }
    const float ecoef = Ly.ecoef;
    char* const e_lds = reinterpret_cast<char*>(lds + Ly.lds_e);
    float* const spill_a = slot >= 0 ? Ly.spill_a + (size_t)slot * P.Bpad * Ly.npad : nullptr;
    float* const spill_e = slot >= 0 ? (l > 0 ? Ly.spill_e + (size_t)slot * P.Bpad * Ly.npad : Ly.spill_e) : nullptr;
    float* const rec = (rec_idx >= 0 && Ly.rec != nullptr) ? Ly.rec + (size_t)rec_idx * P.B * Ly.n : nullptr;
    const uint32_t img_bytes = (uint32_t)P.Bpad * npad4;               // one [Bpad][npad] image (lean_ok: < 4 GiB)
    const bool sys = P.spill_sys != 0;
    float esum = 0.f;
    float amx = 0.f, emx = 0.f;            // largest |value| this call spills of A_l / E_l (mcpc_kernels.h: spill_track)
    float rmx = 0.f;                       // largest |value| this lane writes of its chain's E_l row (rowexp_track)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (i >= nt) continue;
        const int tile = ph.tile0 + kk + tstep * i;
        const uint32_t tb = 64u * (uint32_t)tile;
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) {
            const f32x4 x = xv[i][ct];
            const f32x4 d = x - (av[i][ct] + bv[i][ct]);                  // x - mu
            const f32x4 e = d * ecoef;
            if (l > 0) { *reinterpret_cast<f32x4*>(e_lds + lrowb[ct] + tb) = e; rmx = absmax4(rmx, e); }
            if (slot >= 0) {
#ifdef MCPC_EXP_SPILL_LINEAR      // timing experiment only (rows permuted inside the workgroup's block): one contiguous KiB per store
                const uint32_t sb = mul24(L.chain[ct] - (uint32_t)L.c, npad4) + 1024u * (uint32_t)tile + 16u * (uint32_t)(L.c + 16 * L.q);
#else
                const uint32_t sb = rowb[ct] + tb;
#endif
                // tile-major image: (chain's row tile, unit tile) is one contiguous KiB, lane (c, q) its float4 number c + 16 q
                const uint32_t sb_tm = mul24(L.chain[ct] - (uint32_t)L.c, npad4) + 1024u * (uint32_t)tile + 16u * (uint32_t)(L.c + 16 * L.q);
                spill_st4(spill_a, img_bytes, Ly.spill_a_tm ? sb_tm : sb, mask4(act4<ACT>(x), L.livem[ct]), sys);
                amx = absmax4(amx, mask4(act4<ACT>(x), L.livem[ct]));
                if (l > 0) { spill_st4(spill_e, img_bytes, Ly.spill_e_tm ? sb_tm : sb, mask4(e, L.livem[ct]), sys); emx = absmax4(emx, mask4(e, L.livem[ct])); }
                else if (e0_in_regs) e0acc[ct] = e0acc[ct] + e;                          // (one tile per wave: i == 0 only)
                else gst4s(spill_e, rowb[ct] + tb, gld4s(spill_e, rowb[ct] + tb) + e);   // Linear 0: only sum_t e_1 is needed
            }
            if (rec != nullptr && L.livem[ct]) st_unpadded(rec, (int)L.chain[ct], Ly.n, 16 * tile + 4 * L.q, x);
            const f32x4 dd = d * d;
            esum += L.livem[ct] ? 0.5f * ecoef * (dd.x + dd.y + dd.z + dd.w) : 0.f;          // (a select, not a product: whatever a padding chain holds)
        }
    }
    if (slot >= 0) {
        spill_track(lds + P.lds_spillmax, spill_id_a(l), amx, L.c + 16 * L.q);
        if (l > 0) spill_track(lds + P.lds_spillmax, spill_id_e(l), emx, L.c + 16 * L.q);
    }
    if (CTT == 1 && rx != nullptr && ph.o_row >= 0) rowexp_track(rx, ph.o_row, L.c, rmx, row_gen);
    return esum;
}

// The running sum e0sum[chain][unit] of this wave's tile (wave kk owns tile kk of the top layer) is READ into the registers
// before the step loop and WRITTEN back after it: every chain's sum is then one sequential fp32 chain over its steps, whatever the
// launches a call is cut into (plain segments, the cycles of the round schedule) -- adding a launch's partial sum to the global
// value instead made the result depend on where the launches end.
template <int CTT>
__device__ __forceinline__ void lean_load_e0(const KParams& P, int kk, const LeanLane<CTT>& L, f32x4 (&e0acc)[CTT]) {
    const KLayer& Ly = P.layer[0];
    if (kk >= Ly.ntiles) return;
    const uint32_t npad4 = 4u * (uint32_t)Ly.npad;
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) e0acc[ct] = gld4s(Ly.spill_e, mul24(L.chain[ct], npad4) + 16u * L.q + 64u * (uint32_t)kk);
}
template <int CTT>
__device__ __forceinline__ void lean_flush_e0(const KParams& P, int kk, const LeanLane<CTT>& L, const f32x4 (&e0acc)[CTT]) {
    const KLayer& Ly = P.layer[0];
    if (kk >= Ly.ntiles) return;
    const uint32_t npad4 = 4u * (uint32_t)Ly.npad;
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) gst4s(Ly.spill_e, mul24(L.chain[ct], npad4) + 16u * L.q + 64u * (uint32_t)kk, e0acc[ct]);
}

// XL: after the step loop every E wave writes the state rows of its tiles back (the tiles its BWD entries updated: tile = kk + NW i
// of every layer; FWD and BWD entries of a layer hand out tiles alike, rot == 0, chunk starts multiples of NW)
template <int CTT, int NW>
__device__ __forceinline__ void lean_store_x(const KParams& P, const float* lds, int kk, const LeanLane<CTT>& L) {
    for (int l = 0; l < P.L; ++l) {
        const KLayer& Ly = P.layer[l];
        const uint32_t npad4 = 4u * (uint32_t)Ly.npad;
        for (int tile = kk; tile < Ly.ntiles; tile += NW) {
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) {
                const uint32_t lb = mul24(L.lrow[ct], 4u * (uint32_t)Ly.ld) + 16u * L.q + 64u * (uint32_t)tile;
                gst4s(Ly.x, mul24(L.chain[ct], npad4) + 16u * L.q + 64u * (uint32_t)tile,
                      *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + Ly.lds_x) + lb));
            }
        }
    }
}

// ---- BWD entry (layer l): x_l <- x_l - lr (e_l + sign f'(x_l) back) [+ Philox kick], f(x_l new) -> FX_l ---------------
// back = acc from G (GEMM over E_{l+1}, or the read-out back-projection handed over in registers); none for sign == 0.
// ADAM (the MAP warm-up, torch.optim.Adam on x without noise): the moments m, v of the wave's tiles are requested with x,
// in front of the wait for the partner's block -- the generic epilogue loads them behind it, one L2/HBM round trip exposed
// per x update -- and the arithmetic is the generic epilogue's, operation for operation (s_tab: row of the bias-correction
// table).
template <int CTT, int NW, int NTW, int ACT, bool NOISE, bool ADAM = false, bool XL = false, bool REG = false>
__device__ __forceinline__ void lean_bwd(const KParams& P, const KPhase& ph, float* lds, int nt, int kk, const LeanLane<CTT>& L,
                                         int t, const int* prog_g, int need, int* err, int& dead, int s_tab = 0, float* rx = nullptr,
                                         unsigned row_gen = 0u, const f32x4 (*racc)[CTT] = nullptr) {
    static_assert(!(NOISE && ADAM), "Adam with the fused kick takes the generic epilogue");
    if (nt <= 0) return;
    const int tstep = REG ? ph.rot : NW;
    const KLayer& Ly = P.layer[ph.layer];
    const int l = ph.layer, n = Ly.n;
    const bool from_g = ((ph.flags & PHF_WS_GEMM) && ph.nkb > 0) || (ph.flags & PHF_WS2_HANDOFF);
    const uint32_t npad4 = 4u * (uint32_t)Ly.npad;
    uint32_t rowb[CTT], lrowb[CTT], orowb[CTT];
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) {
        rowb[ct] = mul24(L.chain[ct], npad4) + 16u * L.q;
        lrowb[ct] = mul24(L.lrow[ct], 4u * (uint32_t)Ly.ld) + 16u * L.q;
        orowb[ct] = mul24(L.lrow[ct], 4u * (uint32_t)ph.out_ld) + 16u * L.q;
    }
    const float ecoef = Ly.ecoef;
    f32x4 xv[NTW][CTT], ev[NTW][CTT];
    f32x4 mv[ADAM ? NTW : 1][CTT], vv[ADAM ? NTW : 1][CTT];
    const char* const e_lds = reinterpret_cast<const char*>(lds + Ly.lds_e);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const uint32_t tb = 64u * (uint32_t)(ph.tile0 + kk + tstep * (i < nt ? i : 0));
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) {
            if constexpr (XL) {
                xv[i][ct] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + Ly.lds_x) + lrowb[ct] + tb);
            } else {
#if defined(MCPC_EXP_NOX) || defined(MCPC_EXP_NOELOAD)
            xv[i][ct] = splat(0.5f);
#else
            xv[i][ct] = gld4s(Ly.x, rowb[ct] + tb);
#endif
            }
            if constexpr (ADAM) {
                // (moments are kept tile-major: one contiguous KiB per wave access, tile_major_offset in mcpc_kernels.h)
                const uint32_t mb = (mul24(L.chain[ct] >> 4, (uint32_t)Ly.ntiles) + (uint32_t)(ph.tile0 + kk + tstep * (i < nt ? i : 0))) * 1024u + 16u * (uint32_t)(L.c + 16 * L.q);
                mv[i][ct] = gld4s(Ly.m, mb);
                vv[i][ct] = gld4s(Ly.v, mb);
            }
#ifdef MCPC_EXP_NOELOAD
            if (l == 0) ev[i][ct] = (xv[i][ct] - splat(0.25f)) * ecoef;
#else
            if (l == 0) {                                                               // e_1 = c_1 (x_1 - mu_1), mu_1 constant
                if constexpr (XL) ev[i][ct] = (xv[i][ct] - *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + Ly.lds_bias) + lrowb[ct] + tb)) * ecoef;
                else ev[i][ct] = (xv[i][ct] - gld4(P.mu1, rowb[ct] + tb)) * ecoef;
            }
#endif
            else if constexpr (!ADAM) ev[i][ct] = *reinterpret_cast<const f32x4*>(e_lds + lrowb[ct] + tb);
        }
    }
    f32x4 av[NTW][CTT];
    if constexpr (REG) {       // (unified-wave kernel: the back-projection arrives in registers, zeros when there is none)
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = racc[i][ct];
        (void)from_g; (void)prog_g; (void)need; (void)err; (void)dead; (void)orowb;
    } else if (from_g) {
        ws_wait_one(prog_g, need, err, dead);
        const char* const src = reinterpret_cast<const char*>(lds + ph.out_lds);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const uint32_t tb = 64u * (uint32_t)(ph.tile0 + kk + tstep * (i < nt ? i : 0));
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = *reinterpret_cast<const f32x4*>(src + orowb[ct] + tb);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = splat(0.f);
    }
    const float sign = ph.sign, lr = P.lr, nscale = P.noise_scale;
    const uint64_t seed = P.seed, step = P.step_base + (uint64_t)t, chain_base = P.chain_base;
    char* const fx_lds = reinterpret_cast<char*>(lds + Ly.lds_a);
    float rmx = 0.f;                       // largest |value| this lane writes of its chain's FX_l row (rowexp_track)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (i >= nt) continue;
        const int tile = ph.tile0 + kk + tstep * i;
        const uint32_t tb = 64u * (uint32_t)tile;
        const bool pad_tile = 16 * tile + 16 > n;                  // wave-uniform: only the last tile of a ragged layer
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) {
            const f32x4 x = xv[i][ct], back = av[i][ct];
            f32x4 e;
            if constexpr (ADAM) {         // (the error rows are read here, not held across the wait: registers go to m and v)
                if (l == 0) e = ev[i][ct];
                else e = *reinterpret_cast<const f32x4*>(e_lds + lrowb[ct] + tb);
            } else {
                e = ev[i][ct];
            }
            f32x4 g;
            g.x = e.x + sign * actd<ACT>(x.x, actf<ACT>(x.x)) * back.x;
            g.y = e.y + sign * actd<ACT>(x.y, actf<ACT>(x.y)) * back.y;
            g.z = e.z + sign * actd<ACT>(x.z, actf<ACT>(x.z)) * back.z;
            g.w = e.w + sign * actd<ACT>(x.w, actf<ACT>(x.w)) * back.w;
            f32x4 xn;
            if constexpr (ADAM) {
                // torch.optim.Adam single-tensor path: lerp_, mul_/addcmul_, sqrt/bias2 + eps, addcdiv_ (adam_x, mcpc_device.h)
                f32x4 m = mv[i][ct], v = vv[i][ct];
                m.x = adam_m(m.x, g.x, P.omb1); m.y = adam_m(m.y, g.y, P.omb1); m.z = adam_m(m.z, g.z, P.omb1); m.w = adam_m(m.w, g.w, P.omb1);
                v.x = adam_v(v.x, g.x, P.beta2, P.omb2); v.y = adam_v(v.y, g.y, P.beta2, P.omb2); v.z = adam_v(v.z, g.z, P.beta2, P.omb2); v.w = adam_v(v.w, g.w, P.beta2, P.omb2);
                const uint32_t mb = (mul24(L.chain[ct] >> 4, (uint32_t)Ly.ntiles) + (uint32_t)tile) * 1024u + 16u * (uint32_t)(L.c + 16 * L.q);
                gst4s(Ly.m, mb, m);
                gst4s(Ly.v, mb, v);
                const float nss = P.adam_coef[2 * s_tab], bc2s = P.adam_coef[2 * s_tab + 1], eps = P.eps;
                xn.x = adam_x(x.x, m.x, v.x, nss, bc2s, eps);
                xn.y = adam_x(x.y, m.y, v.y, nss, bc2s, eps);
                xn.z = adam_x(x.z, m.z, v.z, nss, bc2s, eps);
                xn.w = adam_x(x.w, m.w, v.w, nss, bc2s, eps);
            } else {
                xn = x - g * lr;
            }
            if constexpr (NOISE)
                xn = xn + normals4(seed, step, (uint32_t)l, (uint32_t)(chain_base + (uint64_t)L.chain[ct]), (uint32_t)(4 * tile + L.q)) * nscale;
            if (pad_tile) {       // padded units stay exactly zero (their gradient is zero; only the noise must be masked)
                const int u0 = 16 * tile + 4 * L.q;
                if (u0 + 0 >= n) xn.x = 0.f;
                if (u0 + 1 >= n) xn.y = 0.f;
                if (u0 + 2 >= n) xn.z = 0.f;
                if (u0 + 3 >= n) xn.w = 0.f;
            }
            if constexpr (XL) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds + Ly.lds_x) + lrowb[ct] + tb) = xn;
            } else {
#ifndef MCPC_EXP_NOX
            gst4s(Ly.x, rowb[ct] + tb, xn);
#endif
            }
            const f32x4 fxn = act4<ACT>(xn);
            *reinterpret_cast<f32x4*>(fx_lds + lrowb[ct] + tb) = fxn;                // the next step's GEMMs read f(x_new) from FX_l
            rmx = absmax4(rmx, fxn);
        }
    }
    if (CTT == 1 && rx != nullptr && ph.o_row >= 0) rowexp_track(rx, ph.o_row, L.c, rmx, row_gen);
}

// ---- HEADF entry (read-out chunk): out = acc + bias, e_o = dL/dout -> ring slot (LDS), loss, spills, output records -----
// XL: bias from LDS; YB (with XL only): the target is 0/1, its words come from LDS and no fp32 target is requested -- a compile-time
// choice, so that the 0/1 case holds no global load at all (see the note on loads under an `if` below).
// YWG (with XL, YB, REG): the target words were requested from global memory by the caller, in front of the row's GEMM (ywreg[i])
template <int CTT, int NW, int NTW, bool XL = false, bool YB = false, bool REG = false, bool YWG = false>
__device__ __forceinline__ float lean_headf(const KParams& P, const KPhase& ph, float* lds, int nt, int kk, const LeanLane<CTT>& L,
                                            int slot, int rec_idx, bool do_energy, const int* prog_g, int need, int* err, int& dead,
                                            bool ybin, float* rx = nullptr, const f32x4 (*racc)[CTT] = nullptr, unsigned row_gen = 0u,
                                            bool planes_ok = false, const uint32_t* ywreg = nullptr) {
    if (nt <= 0) return 0.f;
    const int tstep = REG ? ph.rot : NW;
    const KHead& H = P.head;
    // unified-wave kernel: a Bernoulli read-out's error goes to LDS as planes (see below; headf_planes is the GEMM side's test too)
    const bool planes = REG && planes_ok;
    const int kind = H.loss_kind, n = H.n, mask_start = H.mask_start;
    const uint32_t npad4 = 4u * (uint32_t)H.npad;
    uint32_t rowb[CTT], orowb[CTT];
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) {
        rowb[ct] = mul24(L.chain[ct], npad4) + 16u * L.q;
        orowb[ct] = mul24(L.lrow[ct], 4u * (uint32_t)ph.out_ld) + 16u * L.q;
    }
    f32x4 yv[NTW][CTT], bv[NTW];
    uint32_t yw[NTW][CTT];                                   // ybin: the word of the bit-packed target that holds this lane's four units
    const uint32_t wrow4 = 4u * (uint32_t)H.ywords;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tile = ph.tile0 + kk + tstep * (i < nt ? i : 0);
        const uint32_t tb = 64u * (uint32_t)tile;
        if constexpr (XL) {
            bv[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds + H.lds_bias) + 16u * L.q + tb);
        } else {
#ifdef MCPC_EXP_NOELOAD
        bv[i] = splat(0.25f);
#else
        bv[i] = gld4(H.bias, 16u * L.q + tb);
#endif
        }
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) {
            if constexpr (XL && YB && YWG) {
                yv[i][ct] = splat(0.f);
                yw[i][ct] = ywreg[i];                 // (i: a compile-time index after unrolling -- the caller's array stays in registers)
            } else if constexpr (XL && YB) {
                yv[i][ct] = splat(0.f);
                yw[i][ct] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds + H.lds_yw) + mul24(L.lrow[ct], wrow4) + 4u * (uint32_t)(tile >> 1));
            } else if constexpr (XL) {
                // (an fp32 target is read from its tile-major image: one contiguous KiB per wave access, like Adam's moments)
                const bool has_y = kind != MCPC_LOSS_NONE;
                yv[i][ct] = gld4s(has_y ? H.ytile : H.bias,
                                  has_y ? (mul24(L.chain[ct] >> 4, (uint32_t)H.ntiles) + (uint32_t)tile) * 1024u + 16u * (uint32_t)(L.c + 16 * L.q) : 16u * L.q + tb);
                yw[i][ct] = 0u;
            } else {
            // Branch-free on purpose (a load under an `if` makes hipcc join the paths behind `s_waitcnt vmcnt(0)`, which here
            // would also wait for every spill store still in flight): both loads are always issued; the one that is not
            // needed reads a harmless hot address (the bias row again / the word array, which always exists).
#if defined(MCPC_EXP_NOY) || defined(MCPC_EXP_NOELOAD)        // (timing experiment only, wrong results: targets are not loaded)
            yv[i][ct] = splat(0.f); yw[i][ct] = 0u;
#else
            const bool fp32_y = kind != MCPC_LOSS_NONE && !ybin;
            yv[i][ct] = gld4s(fp32_y ? H.y : H.bias, fp32_y ? rowb[ct] + tb : 16u * L.q + tb);
            yw[i][ct] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(
                (gbytes_t)H.ybits + mul24(L.chain[ct], wrow4) + 4u * (uint32_t)(tile >> 1));
#endif
            }
        }
    }
    char* const eo = reinterpret_cast<char*>(lds + ph.out_lds);
    f32x4 av[NTW][CTT];
    if constexpr (REG) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = racc[i][ct];
        (void)prog_g; (void)err; (void)dead;
    } else {
        ws_wait_one(prog_g, need, err, dead);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const uint32_t cb = 64u * (uint32_t)(kk + tstep * (i < nt ? i : 0));               // column inside the chunk
#pragma unroll
            for (int ct = 0; ct < CTT; ++ct) av[i][ct] = *reinterpret_cast<const f32x4*>(eo + orowb[ct] + cb);
        }
    }
    const float inv_var = H.inv_var;
    float* const spill = slot >= 0 ? H.spill_e + (size_t)slot * P.Bpad * H.npad : nullptr;
    float* const rec = (rec_idx >= 0 && H.rec_out != nullptr) ? H.rec_out + (size_t)rec_idx * P.B * H.n : nullptr;
    float lsum = 0.f;
    float omx = 0.f;                       // largest |value| this call spills of E_o
    float rmx = 0.f;                       // largest |value| this lane writes of its chain's row of the chunk (rowexp_track)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (i >= nt) continue;
        const int tile = ph.tile0 + kk + tstep * i;
        const uint32_t tb = 64u * (uint32_t)tile, cb = 64u * (uint32_t)(kk + tstep * i);
        const bool inside = 16 * tile >= mask_start && 16 * tile + 16 <= n;        // wave-uniform: every unit of the tile counts
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) {
            const f32x4 o = av[i][ct] + bv[i];
            f32x4 e = splat(0.f);
            if (kind != MCPC_LOSS_NONE) {
                f32x4 y = yv[i][ct];
                if (ybin) {                                     // units 16 tile + 4 q .. + 3: a nibble of the word
                    const uint32_t nib = yw[i][ct] >> (16u * (uint32_t)(tile & 1) + 4u * (uint32_t)L.q);
                    y.x = (float)(nib & 1u); y.y = (float)((nib >> 1) & 1u); y.z = (float)((nib >> 2) & 1u); y.w = (float)((nib >> 3) & 1u);
                }
                const float ov[4] = {o.x, o.y, o.z, o.w}, yy[4] = {y.x, y.y, y.z, y.w};
                float ev[4];
                const int u0 = 16 * tile + 4 * L.q;
                if (kind == MCPC_LOSS_GAUSSIAN) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool on = inside || ((u0 + r) >= mask_start && (u0 + r) < n);
                        const float dlt = ov[r] - yy[r];
                        ev[r] = on ? inv_var * dlt : 0.f;
                        lsum += (on && L.livem[ct]) ? 0.5f * inv_var * dlt * dlt : 0.f;
                    }
                } else if (do_energy) {
                    if (inside) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float sg, bc;
                            sigmoid_bce_f(ov[r], yy[r], sg, bc);
                            ev[r] = sg - yy[r];
                            lsum += L.livem[ct] ? bc : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool on = (u0 + r) >= mask_start && (u0 + r) < n;
                            float sg, bc;
                            sigmoid_bce_f(ov[r], yy[r], sg, bc);
                            ev[r] = on ? sg - yy[r] : 0.f;
                            lsum += (on && L.livem[ct]) ? bc : 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool on = inside || ((u0 + r) >= mask_start && (u0 + r) < n);
                        ev[r] = on ? sigmoid_f(ov[r]) - yy[r] : 0.f;
                    }
                }
                e.x = ev[0]; e.y = ev[1]; e.z = ev[2]; e.w = ev[3];
            }
            if (REG && planes) {
                // the chunk as the back-projection's GEMM reads it: per 8 units [h0..h7][m0..m7] (mcpc_gemm_f16.h: b_planes), split here, once,
                // with the constant exponent of a bounded read-out error -- the lane's units 4 q .. 4 q + 3 of the tile are one half of a group
                unsigned h0, m0, h1, m1;
                const float sc = pow2i(headb_fixed_exp(MCPC_LOSS_BERNOULLI, true));
                split2_pair(f32x2{e.x, e.y}, sc, h0, m0);
                split2_pair(f32x2{e.z, e.w}, sc, h1, m1);
                char* const gp = eo + (orowb[ct] - 16u * L.q) + cb + 32u * (uint32_t)(L.q >> 1) + 8u * (uint32_t)(L.q & 1);
                *reinterpret_cast<uint2*>(gp) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(gp + 16) = make_uint2(m0, m1);
            } else {
                *reinterpret_cast<f32x4*>(eo + orowb[ct] + cb) = e;
            }
            rmx = absmax4(rmx, e);
#ifdef MCPC_EXP_SPILL_LINEAR
            if (slot >= 0) spill_st4(spill, (uint32_t)P.Bpad * npad4, mul24(L.chain[ct] - (uint32_t)L.c, npad4) + 1024u * (uint32_t)tile + 16u * (uint32_t)(L.c + 16 * L.q), mask4(e, L.livem[ct]), P.spill_sys != 0);
#else
            if (slot >= 0) {
                spill_st4(spill, (uint32_t)P.Bpad * npad4,
                          H.spill_tm ? mul24(L.chain[ct] - (uint32_t)L.c, npad4) + 1024u * (uint32_t)tile + 16u * (uint32_t)(L.c + 16 * L.q) : rowb[ct] + tb,
                          mask4(e, L.livem[ct]), P.spill_sys != 0);
                omx = absmax4(omx, mask4(e, L.livem[ct]));
            }
#endif
            if (rec != nullptr && L.livem[ct]) st_unpadded(rec, (int)L.chain[ct], H.n, 16 * tile + 4 * L.q, o);
        }
    }
    if (slot >= 0) spill_track(lds + P.lds_spillmax, kSpillIdEo, omx, L.c + 16 * L.q);
    // (a ring slot is reused inside a step: its generation is the entry, `need` = entries completed so far)
    // (the unified-wave kernel keeps the whole e_o row in LDS: its chunks of one step combine under the step's generation)
    if (CTT == 1 && rx != nullptr && ph.o_row >= 0) rowexp_track(rx, ph.o_row, L.c, rmx, REG ? row_gen : (unsigned)need);
    return lsum;
}

}  // namespace mcpc
