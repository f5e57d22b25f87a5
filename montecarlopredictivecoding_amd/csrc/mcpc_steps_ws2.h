// Wave-specialised Langevin step kernel, in-place variant: the default kernel (16 chains per workgroup, one workgroup per CU).
//
// Roles as in mcpc_steps_ws.h: waves 0-3 ("G", one per SIMD) stream weight fragments and issue MFMAs, waves 4-7
// ("E") run the epilogues.  What differs is how the two roles meet:
//   * no staging slots: G_k stores a finished accumulator block straight into the LDS rows its consumer reads
//       FWD_l   -> E_l rows   (E_k turns mu into e_l in place; BWD_{l-1} reads E_l as its B operand)
//       HEADF c -> e_o chunk  (E_k turns o into e_o in place; HEADB c reads the chunk)
//       BWD_l   -> FX_l rows  (E_k turns the back-projection into f(x_l_new) in place; the NEXT step reads FX_l)
//     so G never waits for a staging slot to drain, and the 16 KiB of slots pay for un-overlaid FX_l buffers;
//   * every FX_l has its own rows and is refreshed by the x update, so the forward GEMMs of a step depend on the
//     previous step's updates, not on each other: the table (build_phases_ws2) orders a step as
//       read-out chunks (need f(x_{L-1}), updated first) interleaved with FWD_{L-1} ... FWD_1, then the updates
//       BWD_{L-1} ... BWD_0,
//     which leaves one or more GEMMs of slack between every producer epilogue and its consumer GEMM.
// Progress counters (LDS, monotonic, polled with bounded spins):
//   prog_g[k] = table entries completed by G_k, prog_e[k] = by E_k   (absolute: step * n_entries + index + 1)
//   E_k entry p waits prog_g[k] > p (its own pair's block);  G entry p waits "all E past dep_e" (B operand ready)
//   and, for read-out chunks only, "all G past dep_g" (chunk buffer no longer read).  Every other write-after-read
//   hazard is implied by a read dependency -- see the table builder.
#pragma once

namespace mcpc {

#ifdef MCPC_EXP_NOGEMM    // timing experiment only (wrong results): G waves skip their GEMMs
#define MCPC_EXP_GEMM_GATE && P.n_steps < 0
#else
#define MCPC_EXP_GEMM_GATE
#endif

#ifndef MCPC_WS2_PAIRS
#define MCPC_WS2_PAIRS 4
#endif
constexpr int kWs2Pairs = MCPC_WS2_PAIRS;          // (G, E) pairs per workgroup: 4 = one GEMM and one epilogue wave per SIMD.
                                                   // (8 = two of each: measured 99 vs 95 us per step at cfg-M -- the two GEMM waves of a SIMD walk the
                                                   // same table behind the same dependencies, so they stall together instead of filling each other's gaps)
#ifndef MCPC_WS2_SPAN
#define MCPC_WS2_SPAN 16
#endif
#ifndef MCPC_WS2_WAVES_PER_EU
#define MCPC_WS2_WAVES_PER_EU 2
#endif
constexpr int kWs2NT = MCPC_WS2_SPAN / kWs2Pairs;   // unit tiles per pair per table entry: an entry hands out 16 tiles
// Tiles per pair and entry (build-time knob).  Measured: 6 (a third group of two tiles in the register rotation) and 5 against 4 --
// 50.7 / 50.0 / 51.4 us per step at 4096 chains in round 2, 34.9 against 34.4 with the bf16x6 core in round 3: the hand-over count is
// not what a step pays for, and the default stays 4 (one table layout; mcpc_gemm_f16.h's rotation is written for at most four).
#ifndef MCPC_WS2_NT16
#define MCPC_WS2_NT16 kWs2NT
#endif
template <int CTT> constexpr int ws2_nt() { return MCPC_WS2_NT16; }
constexpr int kWs2Threads = 2 * kWs2Pairs * 64;
static_assert(kWs2Pairs == 4 || kWs2Pairs == 8, "4 or 8 pairs");

// Diagnostic build variant -DMCPC_STAMPS -DMCPC_STAMPS_ENTRY=q: the 16 slots hold, per TABLE ENTRY p (p < 16),
//   q = 1: cycles the G wave waits in front of / behind the GEMM of entry p (dep_e, dep_g, dep_se)
//   q = 2: cycles the G wave spends in entry p altogether
//   q = 3: cycles the E wave waits for its partner's block of entry p
//   q = 4: cycles the E wave spends in entry p altogether
#ifdef MCPC_STAMPS_ENTRY
#define STAMP_ENTRY_ADD(p_, d_)                                                                      \
    do {                                                                                             \
        const unsigned long long dd_ = (d_);                                                         \
        switch (p_) {                                                                                \
            case 0: st_sum[0] += dd_; break; case 1: st_sum[1] += dd_; break; case 2: st_sum[2] += dd_; break;     \
            case 3: st_sum[3] += dd_; break; case 4: st_sum[4] += dd_; break; case 5: st_sum[5] += dd_; break;     \
            case 6: st_sum[6] += dd_; break; case 7: st_sum[7] += dd_; break; case 8: st_sum[8] += dd_; break;     \
            case 9: st_sum[9] += dd_; break; case 10: st_sum[10] += dd_; break; case 11: st_sum[11] += dd_; break; \
            case 12: st_sum[12] += dd_; break; case 13: st_sum[13] += dd_; break; case 14: st_sum[14] += dd_; break; \
            default: st_sum[15] += dd_; break;                                                       \
        }                                                                                            \
    } while (0)
#undef STAMP
#define STAMP(i) do {} while (0)
#endif

struct alignas(16) Ws2Sync {
    int prog_e[kWs2Pairs];
    int prog_g[kWs2Pairs];
};

// wait until all kWs2Pairs counters are >= need (shape: see ws_wait_one).  The four counters of a role are one aligned 16-byte
// row of Ws2Sync: one ds_read_b128 per poll.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int ws2_min_all(const int* p) {
    if constexpr (kWs2Pairs == 4) {
        const i32x4_t v = *reinterpret_cast<const volatile i32x4_t*>(p);
        return min(min(v.x, v.y), min(v.z, v.w));
    } else {
        int m = ws_ld(p);
#pragma unroll
        for (int i = 1; i < kWs2Pairs; ++i) m = min(m, ws_ld(p + i));
        return m;
    }
}
__device__ __forceinline__ void ws2_wait_all(const int* p, int need, int* err, int& dead) {
    if (__builtin_expect(ws2_min_all(p) < need, 0)) {
        if (!dead) {
            int spin = 0;
            bool ok = false;
#pragma clang loop unroll(disable)
            do {
                if (MCPC_WS_SLEEP > 0) __builtin_amdgcn_s_sleep(MCPC_WS_SLEEP);
                ok = ws2_min_all(p) >= need;
            } while (!ok && ++spin < kWsSpinLimit);
            if (!ok) { if ((threadIdx.x & 63) == 0) atomicOr(err, 1); dead = 1; }
        }
    }
    MCPC_WS_FENCE(__ATOMIC_ACQUIRE);
}

__device__ __forceinline__ int ws2_need(int base, int n_ent, int p, int dep) {
    return (dep > p ? base - n_ent : base) + dep + 1;      // an index above the entry's own: previous step
}

// G side: request the first two k-blocks of the fragments of an upcoming entry (weights need no dependency).
// Branch-free: all four tile slots always issue both loads; unused slots repeat slot 0 (L1 hits) and entries without
// a GEMM read `dummy` (any 2 KiB of valid global memory).  With the loads under `if (i < nt)` hipcc joined every
// branch behind `s_waitcnt vmcnt(0)`: four serial L2 round trips (~1 k cycles) per table entry.
template <int NT>
__device__ __forceinline__ void ws2_prefetch(const KPhase& ph, int k, int lane, const void* dummy, int& nt_out, int (&aoff)[NT],
                                             frag_t (&pre0)[NT]) {
    const int kk = (k + ph.rot) & (kWs2Pairs - 1);
    int nt = (ph.ntiles - kk + kWs2Pairs - 1) / kWs2Pairs;
    nt = nt < 0 ? 0 : (nt > NT ? NT : nt);
    if (!(ph.flags & PHF_WS_GEMM) || ph.nkb <= 0) nt = 0;
    nt_out = nt;
    const bool valid = nt > 0;
    const gu32x4* const A = valid ? (const gu32x4*)ph.A : (const gu32x4*)dummy;      // (dummy: any 6 KiB of valid global memory)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int ii = i < nt ? i : 0;
        const int off = valid ? (ph.tile0 + kk + kWs2Pairs * ii) * ph.a_tile_stride + ph.a_off0 : 0;
        aoff[i] = off;
        pre0[i] = load_frag(A, off, lane);
    }
}

template <int ACT>
__device__ __forceinline__ void ws2_fill_fx(const KLayer& Ly, float* lds, int chain0, int tid, int ct_rows, bool keep_x,
                                            float* lds_rowexp = nullptr, int row_id = 0) {
    const int qpr = Ly.npad / 4;                            // quads per row
    for (int idx = tid; idx < ct_rows * qpr; idx += kWs2Threads) {
        const int r = idx / qpr, u0 = 4 * (idx - r * qpr);
        const f32x4 x = ld4s(Ly.x + (size_t)(chain0 + r) * Ly.npad + u0);
        f32x4 fx;
        fx.x = actf<ACT>(x.x); fx.y = actf<ACT>(x.y); fx.z = actf<ACT>(x.z); fx.w = actf<ACT>(x.w);
        st4(lds + Ly.lds_a + r * Ly.ld + u0, fx);
        if (keep_x) st4(lds + Ly.lds_x + r * Ly.ld + u0, x);
        // generation 0 of the row's exponent word (mcpc_kernels.h: rowexp_track; the words were zeroed with the plan)
        if (lds_rowexp != nullptr && r < 16) rowexp_track(lds_rowexp, row_id, r, absmax4(0.f, fx), 0u);
    }
}

// KParams::xl: what the epilogues of a launch read over and over and nobody else writes -- biases, the mu_1 rows and the bit-packed
// target rows of the workgroup's chains -- copied to LDS once (the state rows X_l: ws2_fill_fx)
__device__ __forceinline__ void ws2_fill_constants(const KParams& P, float* lds, int chain0, int tid, int ct_rows) {
    for (int l = 1; l < P.L; ++l) {
        const KLayer& Ly = P.layer[l];
        for (int i = tid; i < Ly.npad / 4; i += kWs2Threads) st4(lds + Ly.lds_bias + 4 * i, ld4(Ly.bias + 4 * i));
    }
    {
        const KLayer& Ly = P.layer[0];
        const int qpr = Ly.npad / 4;
        for (int idx = tid; idx < ct_rows * qpr; idx += kWs2Threads) {
            const int r = idx / qpr, u0 = 4 * (idx - r * qpr);
            st4(lds + Ly.lds_bias + r * Ly.ld + u0, ld4(P.mu1 + (size_t)(chain0 + r) * Ly.npad + u0));
        }
    }
    if (P.has_head) {
        const KHead& H = P.head;
        for (int i = tid; i < H.npad / 4; i += kWs2Threads) st4(lds + H.lds_bias + 4 * i, ld4(H.bias + 4 * i));
        if (H.lds_yw >= 0) {
            uint32_t* const yw = reinterpret_cast<uint32_t*>(lds + H.lds_yw);
            for (int idx = tid; idx < ct_rows * H.ywords; idx += kWs2Threads) {
                const int r = idx / H.ywords, w = idx - r * H.ywords;
                yw[idx] = H.ybits[(size_t)(chain0 + r) * H.ywords + w];
            }
        }
    }
}

// Uniform values the epilogue loops use over and over are parked in VGPRs: the E waves have ~100 registers to
// spare but no SGPRs (hipcc re-loaded them from the kernel arguments inside the loops: s_load + s_waitcnt lgkmcnt(0),
// a scalar-cache round trip per use).
__device__ __forceinline__ int vreg(int s) { int v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }
__device__ __forceinline__ float vreg(float s) { float v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }
__device__ __forceinline__ float* vreg(float* s) {
    const uint64_t u = (uint64_t)s;
    const uint32_t lo = (uint32_t)vreg((int)(uint32_t)u), hi = (uint32_t)vreg((int)(uint32_t)(u >> 32));
    return (float*)(((uint64_t)hi << 32) | lo);
}

// HEADF epilogue of the E waves: arithmetic of headf_epilogue (mcpc_kernels.h), uniforms in VGPRs
template <int CTT, int NW, int NTW>
__device__ __forceinline__ float ws2_headf_epilogue(const KParams& P, const KPhase& ph, float* lds, int nt, int wave, int lane,
                                                    int chain0, const f32x4 (&acc)[NTW][CTT], const f32x4 (&pa)[NTW][CTT],
                                                    const f32x4 (&pb)[NTW][CTT], int slot, int rec_idx, bool do_energy) {
    const KHead& H = P.head;
    const int c = lane & 15, q = lane >> 4;
    const int kind = H.loss_kind;
    const int npad = vreg(H.npad), n = vreg(H.n), ld = vreg(ph.out_ld), B = vreg(P.B), mask_start = vreg(H.mask_start);
    const float inv_var = vreg(H.inv_var);
    const int eo_off = vreg(ph.out_lds), tile0x16 = vreg(16 * ph.tile0);
    float* const spill = slot >= 0 ? vreg(H.spill_e + (size_t)slot * P.Bpad * H.npad) : nullptr;
    const int spill_tm = vreg(H.spill_tm);
    float* const rec = (rec_idx >= 0 && H.rec_out != nullptr) ? H.rec_out + (size_t)rec_idx * P.B * H.n : nullptr;
    float lsum = 0.f;
    float omx = 0.f;                       // largest |value| this call spills of E_o (mcpc_kernels.h: spill_track)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (i >= nt) continue;
        const int u0 = tile0x16 + 16 * (wave + NW * i) + 4 * q;
#pragma unroll
        for (int ct = 0; ct < CTT; ++ct) {
            const int cl = 16 * ct + c, chain = chain0 + cl;
            const bool live = chain < B;
            const f32x4 o = acc[i][ct] + pb[i][ct];
            f32x4 e = splat(0.f);
            if (kind != MCPC_LOSS_NONE) {
                const f32x4 y = pa[i][ct];
                const float ov[4] = {o.x, o.y, o.z, o.w}, yv[4] = {y.x, y.y, y.z, y.w};
                float ev[4];
                if (kind == MCPC_LOSS_GAUSSIAN) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool on = live && (u0 + r) >= mask_start && (u0 + r) < n;
                        const float dlt = ov[r] - yv[r];
                        ev[r] = on ? inv_var * dlt : 0.f;
                        lsum += on ? 0.5f * inv_var * dlt * dlt : 0.f;
                    }
                } else if (do_energy) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool on = live && (u0 + r) >= mask_start && (u0 + r) < n;
                        float sg, bc;
                        sigmoid_bce_f(ov[r], yv[r], sg, bc);
                        ev[r] = on ? sg - yv[r] : 0.f;
                        lsum += on ? bc : 0.f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool on = live && (u0 + r) >= mask_start && (u0 + r) < n;
                        ev[r] = on ? sigmoid_f(ov[r]) - yv[r] : 0.f;
                    }
                }
                e.x = ev[0]; e.y = ev[1]; e.z = ev[2]; e.w = ev[3];
            }
            st4(lds + eo_off + cl * ld + (u0 - tile0x16), e);
            if (slot >= 0) { st4s(spill + spill_offset(spill_tm, (size_t)chain, u0, npad), e); omx = absmax4(omx, e); }
            if (rec != nullptr && live) st_unpadded(rec, chain, H.n, u0, o);
        }
    }
    if (slot >= 0) spill_track(lds + P.lds_spillmax, kSpillIdEo, omx, lane);
    return lsum;
}

// MIX: the launch is one launch of a round-schedule cycle (setup_rounds, mcpc_api.hip) -- workgroups take their unit and the steps it
// has already done from per-launch lists (KParams::wg_list / wg_rel).
// (The body is a textual include: wrapped into a device function and inlined, the SAME code compiled about 1 % slower.)
template <int CTT, bool MIX = false>
__global__ __launch_bounds__(kWs2Threads, MCPC_WS2_WAVES_PER_EU) void mcpc_steps_ws2_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#define WS2_BLOCK blockIdx.x
#define WS2_NBLOCKS gridDim.x
#include "mcpc_steps_ws2_body.inc"
#undef WS2_BLOCK
#undef WS2_NBLOCKS
}

}  // namespace mcpc
