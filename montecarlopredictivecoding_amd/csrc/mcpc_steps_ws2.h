// Wave-specialised Langevin step kernel, in-place variant (the default for large shards; 32 chains per workgroup).
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
// Tiles per pair and entry of a 16-CHAIN workgroup (build-time knob): with one chain tile the same accumulator registers would
// hold 8 unit tiles, i.e. half as many table entries and hand-overs per step.  Measured (round 2): 6 (the most that compiles
// without spilling: three fragment sets of NT registers each) and 5 against 4 -- 50.7 / 50.0 / 51.4 us per step at 4096 chains,
// 76.7 / 76.4 / 76.2 on the mixed schedule at 6000: within the noise, so the hand-over count is not what the 16-chain form
// pays for, and the default stays 4 (one table layout, 140 KB less code).
#ifndef MCPC_WS2_NT16
#define MCPC_WS2_NT16 kWs2NT
#endif
template <int CTT> constexpr int ws2_nt() { return CTT == 1 ? MCPC_WS2_NT16 : kWs2NT; }
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

struct Ws2Sync {
    int prog_e[kWs2Pairs];
    int prog_g[kWs2Pairs];
};

// wait until all kWs2Pairs counters are >= need (see ws_wait_all)
__device__ __forceinline__ void ws2_wait_all(const int* p, int need, int* err, int& dead) {
    int spin = dead ? kWsSpinLimit : 0;
    for (; spin < kWsSpinLimit; ++spin) {
        int m = ws_ld(p);
#pragma unroll
        for (int i = 1; i < kWs2Pairs; ++i) m = min(m, ws_ld(p + i));
        if (m >= need) break;
        if (MCPC_WS_SLEEP > 0) __builtin_amdgcn_s_sleep(MCPC_WS_SLEEP);
    }
    if (spin == kWsSpinLimit) { if (!dead && (threadIdx.x & 63) == 0) atomicOr(err, 1); dead = 1; }
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
                                             f32x4 (&pre0)[NT], f32x4 (&pre1)[NT]) {
    const int kk = (k + ph.rot) & (kWs2Pairs - 1);
    int nt = (ph.ntiles - kk + kWs2Pairs - 1) / kWs2Pairs;
    nt = nt < 0 ? 0 : (nt > NT ? NT : nt);
    if (!(ph.flags & PHF_WS_GEMM) || ph.nkb <= 0) nt = 0;
    nt_out = nt;
    const bool valid = nt > 0;
    const gf32x4* const A = valid ? (const gf32x4*)ph.A : (const gf32x4*)dummy;
    const int second = (valid && ph.nkb > 1) ? 64 : 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int ii = i < nt ? i : 0;
        const int off = valid ? (ph.tile0 + kk + kWs2Pairs * ii) * ph.a_tile_stride + ph.a_off0 : 0;
        aoff[i] = off;
        pre0[i] = A[off + lane];
        pre1[i] = A[off + second + lane];
    }
}

template <int ACT>
__device__ __forceinline__ void ws2_fill_fx(const KLayer& Ly, float* lds, int chain0, int tid, int ct_rows) {
    const int qpr = Ly.npad / 4;                            // quads per row
    for (int idx = tid; idx < ct_rows * qpr; idx += kWs2Threads) {
        const int r = idx / qpr, u0 = 4 * (idx - r * qpr);
        const f32x4 x = ld4s(Ly.x + (size_t)(chain0 + r) * Ly.npad + u0);
        f32x4 fx;
        fx.x = actf<ACT>(x.x); fx.y = actf<ACT>(x.y); fx.z = actf<ACT>(x.z); fx.w = actf<ACT>(x.w);
        st4(lds + Ly.lds_a + r * Ly.ld + u0, fx);
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
    float* const rec = (rec_idx >= 0 && H.rec_out != nullptr) ? H.rec_out + (size_t)rec_idx * P.B * H.n : nullptr;
    float lsum = 0.f;
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
            if (slot >= 0) st4s(spill + (size_t)chain * npad + u0, e);
            if (rec != nullptr && live) st_unpadded(rec, chain, H.n, u0, o);
        }
    }
    return lsum;
}

// MIX: the launch is one half of a mixed schedule -- workgroups take their unit and their step offset from lists
template <int CTT, bool MIX = false>
__global__ __launch_bounds__(kWs2Threads, MCPC_WS2_WAVES_PER_EU) void mcpc_steps_ws2_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = kWs2Pairs, NTW = ws2_nt<CTT>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_g = wave8 < kWs2Pairs;                   // waves 0..NW-1 and NW..2NW-1 both spread evenly over the 4 SIMDs
    const int k = wave8 & (kWs2Pairs - 1);                 // pair id
    const int c = lane & 15, q = lane >> 4;
    const int unit = MIX ? P.wg_list[blockIdx.x] : (int)blockIdx.x;             // pair of chain tiles (CTT = 2) or single tile
    const int chain0 = unit * (16 * CTT);
    // first step of this unit in this launch (mixed schedule: units advance at different rates, so each carries the number
    // of segments it has spent split / paired since the cycle began)
    const int t_first = MIX ? P.t0 + (P.wg_rel[blockIdx.x] & 0xffff) * P.mix_ms + (P.wg_rel[blockIdx.x] >> 16) * P.mix_mp : P.t0;
    const int L = P.L;
    const int n_ent = P.n_phases;
    Ws2Sync* sync = reinterpret_cast<Ws2Sync*>(lds + P.lds_ws_sync);
    int dead = 0;                                          // set once a bounded wait of this wave ran out
    if (tid < 2 * kWs2Pairs) reinterpret_cast<int*>(sync)[tid] = 0;
    // f(x_l) of the state the launch starts from; afterwards the x updates keep FX_l current
    for (int l = 0; l < L; ++l) {
        const KLayer& Ly = P.layer[l];
        if (Ly.act == MCPC_ACT_RELU) ws2_fill_fx<MCPC_ACT_RELU>(Ly, lds, chain0, tid, 16 * CTT);
        else if (Ly.act == MCPC_ACT_TANH) ws2_fill_fx<MCPC_ACT_TANH>(Ly, lds, chain0, tid, 16 * CTT);
        else ws2_fill_fx<MCPC_ACT_IDENTITY>(Ly, lds, chain0, tid, 16 * CTT);
    }
    __syncthreads();                                       // the only barrier

    if (is_g) {
        // =========================== G: fragments + MFMAs ==============================================
        if (P.ws_prio == 2) __builtin_amdgcn_s_setprio(2);
        // Loop-carried: the descriptor of the upcoming entry and the first two k-blocks of its fragments.  The GEMM
        // moves pre0/pre1 into its own register sets first thing, so the ONE prefetch site below refills the same
        // variables -- with separate "next" copies hipcc put `s_waitcnt vmcnt(0)` + 16 v_mov at the loop latch, i.e.
        // every entry waited out the L2 round trip of the prefetch it had just issued.
        KPhase ph_next = load_phase(P.phases, 0);
        int nt_next, aoff[NTW];
        f32x4 pre0[NTW], pre1[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) { pre0[i] = splat(0.f); pre1[i] = splat(0.f); aoff[i] = 0; }
        ws2_prefetch(ph_next, k, lane, P.mu1, nt_next, aoff, pre0, pre1);
        STAMP_DECL
#ifdef MCPC_STAMPS
        const unsigned long long clk_m0 = mcpc_stamp(), clk_r0 = wall_clock64();
#endif
        for (int s = 0; s < P.n_steps; ++s) {
            const int base = s * n_ent;
            f32x4 accb[NTW][CTT];                       // back-projection of the read-out error: 16 tiles over the pairs
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int ct = 0; ct < CTT; ++ct) accb[i][ct] = splat(0.f);
#pragma unroll 1
            for (int p = 0; p < n_ent; ++p) {
                const KPhase ph = ph_next;
                const int nt = nt_next;
                const bool has_next = (p + 1 < n_ent) || (s + 1 < P.n_steps);
                if (has_next) ph_next = load_phase(P.phases, p + 1 < n_ent ? p + 1 : 0);
                const bool handoff = (ph.flags & PHF_WS2_HANDOFF) != 0;             // accb goes to the partner
                const bool is_headb = ph.type == PH_HEADB;
                const bool works = (ph.flags & PHF_WS_GEMM) || handoff;
                const bool stores = works && !is_headb;
                f32x4 acc[NTW][CTT];
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTT; ++ct) acc[i][ct] = splat(0.f);
                STAMP(0);
#ifdef MCPC_STAMPS_ENTRY
                const unsigned long long ge0 = mcpc_stamp();
                unsigned long long gwait = 0;
#endif
                if (works) {
                    if (ph.dep_e >= 0) ws2_wait_all(sync->prog_e, ws2_need(base, n_ent, p, ph.dep_e), P.err, dead);
#ifdef MCPC_STAMPS_ENTRY
                    gwait = mcpc_stamp() - ge0;
#endif
                    STAMP(1);
                    if (is_headb) {
                        if (nt > 0 MCPC_EXP_GEMM_GATE) gemm_tiles<NTW, CTT, NW>(accb, (const gf32x4*)ph.A, aoff, nt, ph.nkb, lds + ph.b_lds, ph.ldb, lane, pre0, pre1);
                        STAMP(2);
                    } else {
                        if (handoff) {
#pragma unroll
                            for (int i = 0; i < NTW; ++i)
#pragma unroll
                                for (int ct = 0; ct < CTT; ++ct) acc[i][ct] = accb[i][ct];     // one entry covers all of accb
                        }
                        if (nt > 0 && ph.nkb > 0 MCPC_EXP_GEMM_GATE)
                            gemm_tiles<NTW, CTT, NW>(acc, (const gf32x4*)ph.A, aoff, nt, ph.nkb, lds + ph.b_lds, ph.ldb, lane, pre0, pre1);
                        STAMP(3);
                    }
                }
                // the one prefetch site: fragments of the next entry travel while this block is handed over
                ws2_prefetch(ph_next, k, lane, P.mu1, nt_next, aoff, pre0, pre1);   // (past the last entry: the old descriptor again)
                STAMP(4);
                if (stores) {
                    // write-after-read: the rows this block goes to may still be read by GEMMs (dep_g) or epilogues (dep_se)
                    // of entries that share them -- waited for here, behind the GEMM, not in front of it
#ifdef MCPC_STAMPS_ENTRY
                    const unsigned long long gw0 = mcpc_stamp();
#endif
                    if (ph.dep_g >= 0) ws2_wait_all(sync->prog_g, ws2_need(base, n_ent, p, ph.dep_g), P.err, dead);
                    if (ph.dep_se >= 0) ws2_wait_all(sync->prog_e, ws2_need(base, n_ent, p, ph.dep_se), P.err, dead);
#ifdef MCPC_STAMPS_ENTRY
                    gwait += mcpc_stamp() - gw0;
#endif
                    // the block goes where its consumer reads it; E_k finishes it in place
                    const int kk = (k + ph.rot) & (NW - 1);
                    int ntw = (ph.ntiles - kk + NW - 1) / NW;
                    ntw = ntw < 0 ? 0 : (ntw > NTW ? NTW : ntw);
                    float* const out = lds + ph.out_lds;
                    const int col0 = (ph.type == PH_HEADF) ? 0 : 16 * ph.tile0;
#pragma unroll
                    for (int i = 0; i < NTW; ++i) {
                        if (i >= ntw) continue;
                        const int col = col0 + 16 * (kk + NW * i) + 4 * q;
#pragma unroll
                        for (int ct = 0; ct < CTT; ++ct) st4(out + (16 * ct + c) * ph.out_ld + col, acc[i][ct]);
                    }
                }
                if (lane == 0) ws_publish(&sync->prog_g[k], base + p + 1);
                STAMP(5);
#ifdef MCPC_STAMPS_ENTRY
                if (MCPC_STAMPS_ENTRY == 1) STAMP_ENTRY_ADD(p, gwait);
                else if (MCPC_STAMPS_ENTRY == 2) STAMP_ENTRY_ADD(p, mcpc_stamp() - ge0);
#endif
            }
        }
#if defined(MCPC_STAMPS) && !defined(MCPC_STAMPS_ENTRY)
        st_sum[6] = mcpc_stamp() - clk_m0;          // whole launch in s_memtime ticks ...
        st_sum[7] = wall_clock64() - clk_r0;        // ... and in 100 MHz wall-clock ticks: their ratio is the shader clock
#endif
#ifdef MCPC_STAMPS
        if (lane == 0 && P.dbg != nullptr)      // (null in the warm-up launch of setup_mixed_schedule)
            for (int i = 0; i < 16; ++i) P.dbg[((size_t)blockIdx.x * (2 * kWs2Pairs) + wave8) * 16 + i] = st_sum[i];
#endif
        return;
    }

    // =============================== E: epilogues ===========================================================
    if (P.ws_prio == 1) __builtin_amdgcn_s_setprio(2);
    const int upd_mode = (P.update_x && P.xopt == MCPC_XOPT_SGD)
                             ? (P.noise_mode == MCPC_NOISE_PHILOX ? 2 : (P.noise_mode == MCPC_NOISE_NONE ? 1 : 0)) : 0;
    // lean epilogues (mcpc_ws2_lean.h): fused SGD update (or Adam without noise: the MAP warm-up), every chain of the
    // workgroup inside the batch, 32-bit offsets fit
    const bool lean_adam = P.update_x && P.xopt == MCPC_XOPT_ADAM && P.noise_mode == MCPC_NOISE_NONE;
    const bool lean = (upd_mode != 0 || lean_adam) && chain0 + 16 * CTT <= P.B && P.lean_ok;
    LeanLane<CTT> LL;
    LL.c = c; LL.q = q;
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) { LL.chain[ct] = (uint32_t)(chain0 + 16 * ct + c); LL.lrow[ct] = (uint32_t)(16 * ct + c); }
    // 0/1 targets are read bit-packed (a wave-uniform flag the library set when the target was bound)
    const bool ybin = lean && P.has_head && *P.head.y_binary != 0;
    // sum_t e_1 of this launch in registers (lean path, top layer of at most one tile per wave, rot == 0 for FWD entries)
    const bool e0_in_regs = lean && P.layer[0].ntiles <= NW;
    bool e0_dirty = false;
    f32x4 e0acc[CTT];
#pragma unroll
    for (int ct = 0; ct < CTT; ++ct) e0acc[ct] = splat(0.f);
    STAMP_DECL
    for (int s = 0; s < P.n_steps; ++s) {
        const int t = t_first + s;
        const int s_tab = MIX ? t - P.t0 : s;          // index into per-step tables (Adam coefficients, external noise): they start at P.t0
        const int base = s * n_ent;
        const bool do_energy = (P.energy_mode == MCPC_ENERGY_ALL) || (P.energy_mode == MCPC_ENERGY_LAST && t == P.T - 1);
        const int slot = (t >= P.acc_begin && t < P.acc_end) ? (t - P.spill_t0) : -1;
        int rec_idx = -1;
        if (P.rec_count > 0 && t >= P.rec_begin) {
            const int kk = (t - P.rec_begin) / P.rec_stride;
            if (kk < P.rec_count && P.rec_begin + kk * P.rec_stride == t) rec_idx = kk;
        }
        // energies of this step: lane c of every E wave carries the wave's partial sum of column c (layers 0..5, loss) in a
        // register; the LDS scratch is written once per step, at the ENERGY entry (an LDS read-modify-write behind every
        // epilogue cost the E waves seven LDS round trips per step)
        float* red = lds + P.lds_red + (s & 1) * (kMaxLatent + 1) * kMaxWaves;
        float en_acc = 0.f;
#pragma unroll 1
        for (int p = 0; p < n_ent; ++p) {
            const KPhase ph = load_phase(P.phases, p);
#ifdef MCPC_STAMPS_ENTRY
            const unsigned long long ee0 = mcpc_stamp();
            unsigned long long ewait = 0;
#endif
            if (ph.type == PH_ENERGY) {
                if (do_energy && lane <= kMaxLatent) red[lane * kMaxWaves + k] = en_acc;
                if (lane == 0) ws_publish(&sync->prog_e[k], base + p + 1);
                if (do_energy && k == 0) {
                    // every E wave has written its partial sums of this step
                    ws2_wait_all(sync->prog_e + 0, base + p + 1, P.err, dead);
                    if (lane <= kMaxLatent) {
                        double v = 0.0;
                        const bool used = (lane < L) || (lane == kMaxLatent && P.has_head);
                        if (used) {
#pragma unroll
                            for (int w = 0; w < kWs2Pairs; ++w) v += (double)red[lane * kMaxWaves + w];
                        }
                        const int erow = (P.energy_mode == MCPC_ENERGY_ALL) ? t : 0;
                        // a row has one slot per 16-chain tile: a 32-chain workgroup writes the slot of its first tile and
                        // clears the other (a 16-chain workgroup may have written it for the same step of an earlier call)
                        double* const ep = P.epart + (MIX ? (size_t)erow * P.epart_slots + (size_t)unit * CTT
                                                          : (size_t)erow * (CTT * gridDim.x) + (size_t)blockIdx.x * CTT) * (kMaxLatent + 1) + lane;
                        ep[0] = v;
                        if (CTT == 2) ep[kMaxLatent + 1] = 0.0;
                    }
                }
                continue;
            }
            if (!(ph.flags & PHF_WS_EPI)) {
                if (lane == 0) ws_publish(&sync->prog_e[k], base + p + 1);
                continue;
            }
            const int kk = (k + ph.rot) & (NW - 1);
            int nt = (ph.ntiles - kk + NW - 1) / NW;
            nt = nt < 0 ? 0 : (nt > NTW ? NTW : nt);
#ifndef MCPC_EXP_NOLEAN
            if (lean) {
                const int act = P.layer[ph.layer].act;
                const int* const pg = &sync->prog_g[k];
                const int need = base + p + 1;
#ifdef MCPC_EXP_NOEPI
                if (P.n_steps < 0)
#endif
                if (ph.type == PH_FWD) {
                    float esum;
                    if (ph.layer == 0 && slot >= 0) e0_dirty = true;
                    if (act == MCPC_ACT_RELU) esum = lean_fwd<CTT, NW, NTW, MCPC_ACT_RELU>(P, ph, lds, nt, kk, LL, slot, rec_idx, pg, need, P.err, dead, e0acc, e0_in_regs);
                    else if (act == MCPC_ACT_TANH) esum = lean_fwd<CTT, NW, NTW, MCPC_ACT_TANH>(P, ph, lds, nt, kk, LL, slot, rec_idx, pg, need, P.err, dead, e0acc, e0_in_regs);
                    else esum = lean_fwd<CTT, NW, NTW, MCPC_ACT_IDENTITY>(P, ph, lds, nt, kk, LL, slot, rec_idx, pg, need, P.err, dead, e0acc, e0_in_regs);
                    if (do_energy) { esum = wave_sum(esum); if (lane == ph.layer) en_acc += esum; }
                } else if (ph.type == PH_HEADF) {
                    float lsum = lean_headf<CTT, NW, NTW>(P, ph, lds, nt, kk, LL, slot, rec_idx, do_energy, pg, need, P.err, dead, ybin);
                    if (do_energy) { lsum = wave_sum(lsum); if (lane == kMaxLatent) en_acc += lsum; }
                } else if (ph.type == PH_BWD) {
                    if (lean_adam) {
                        if (act == MCPC_ACT_RELU) lean_bwd<CTT, NW, NTW, MCPC_ACT_RELU, false, true>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead, s_tab);
                        else if (act == MCPC_ACT_TANH) lean_bwd<CTT, NW, NTW, MCPC_ACT_TANH, false, true>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead, s_tab);
                        else lean_bwd<CTT, NW, NTW, MCPC_ACT_IDENTITY, false, true>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead, s_tab);
                    } else if (upd_mode == 2) {
                        if (act == MCPC_ACT_RELU) lean_bwd<CTT, NW, NTW, MCPC_ACT_RELU, true>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead);
                        else if (act == MCPC_ACT_TANH) lean_bwd<CTT, NW, NTW, MCPC_ACT_TANH, true>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead);
                        else lean_bwd<CTT, NW, NTW, MCPC_ACT_IDENTITY, true>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead);
                    } else {
                        if (act == MCPC_ACT_RELU) lean_bwd<CTT, NW, NTW, MCPC_ACT_RELU, false>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead);
                        else if (act == MCPC_ACT_TANH) lean_bwd<CTT, NW, NTW, MCPC_ACT_TANH, false>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead);
                        else lean_bwd<CTT, NW, NTW, MCPC_ACT_IDENTITY, false>(P, ph, lds, nt, kk, LL, t, pg, need, P.err, dead);
                    }
                }
                if (lane == 0) ws_publish(&sync->prog_e[k], base + p + 1);
                continue;
            }
#endif
            f32x4 acc[NTW][CTT], pa[NTW][CTT], pb[NTW][CTT];
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int ct = 0; ct < CTT; ++ct) { acc[i][ct] = splat(0.f); pa[i][ct] = splat(0.f); pb[i][ct] = splat(0.f); }
            const KLayer& Ly = P.layer[ph.layer];
            STAMP(8);
            // operands of the epilogue travel while the partner still computes
            issue_epilogue_loads<CTT, NW, NTW>(P, ph, lds, nt, kk, lane, chain0, pa, pb);
            STAMP(9);
            const bool from_g = ((ph.flags & PHF_WS_GEMM) && ph.nkb > 0) || (ph.flags & PHF_WS2_HANDOFF);
            if (from_g) {
#ifdef MCPC_STAMPS_ENTRY
                const unsigned long long ew0 = mcpc_stamp();
#endif
                ws_wait_one(&sync->prog_g[k], base + p + 1, P.err, dead);
#ifdef MCPC_STAMPS_ENTRY
                ewait = mcpc_stamp() - ew0;
#endif
                STAMP(10);
                const float* const src = lds + ph.out_lds;
                const int col0 = (ph.type == PH_HEADF) ? 0 : 16 * ph.tile0;
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const int col = col0 + 16 * (kk + NW * (i < nt ? i : 0)) + 4 * q;     // unused slots re-read slot 0
#pragma unroll
                    for (int ct = 0; ct < CTT; ++ct) acc[i][ct] = ld4(src + (16 * ct + c) * ph.out_ld + col);
                }
            }
#ifdef MCPC_EXP_NOEPI   // timing experiment only (wrong results): E waves skip the epilogue arithmetic and stores
            if (P.n_steps < 0)
#endif
            if (ph.type == PH_FWD) {
                float esum;
                if (Ly.act == MCPC_ACT_RELU) esum = fwd_epilogue<CTT, NW, NTW, MCPC_ACT_RELU, false>(P, ph, lds, nt, kk, lane, chain0, acc, pa, pb, slot, rec_idx);
                else if (Ly.act == MCPC_ACT_TANH) esum = fwd_epilogue<CTT, NW, NTW, MCPC_ACT_TANH, false>(P, ph, lds, nt, kk, lane, chain0, acc, pa, pb, slot, rec_idx);
                else esum = fwd_epilogue<CTT, NW, NTW, MCPC_ACT_IDENTITY, false>(P, ph, lds, nt, kk, lane, chain0, acc, pa, pb, slot, rec_idx);
                if (do_energy) { esum = wave_sum(esum); if (lane == ph.layer) en_acc += esum; }
            } else if (ph.type == PH_HEADF) {
                float lsum = ws2_headf_epilogue<CTT, NW, NTW>(P, ph, lds, nt, kk, lane, chain0, acc, pa, pb, slot, rec_idx, do_energy);
                if (do_energy) { lsum = wave_sum(lsum); if (lane == kMaxLatent) en_acc += lsum; }
            } else if (ph.type == PH_BWD) {
                if (Ly.act == MCPC_ACT_RELU) bwd_epilogue_mode<CTT, NW, NTW, MCPC_ACT_RELU, true>(P, ph, nt, kk, lane, chain0, acc, pa, pb, s_tab, t, upd_mode, lds);
                else if (Ly.act == MCPC_ACT_TANH) bwd_epilogue_mode<CTT, NW, NTW, MCPC_ACT_TANH, true>(P, ph, nt, kk, lane, chain0, acc, pa, pb, s_tab, t, upd_mode, lds);
                else bwd_epilogue_mode<CTT, NW, NTW, MCPC_ACT_IDENTITY, true>(P, ph, nt, kk, lane, chain0, acc, pa, pb, s_tab, t, upd_mode, lds);
            }
            if (lane == 0) ws_publish(&sync->prog_e[k], base + p + 1);
            if (ph.type == PH_FWD) STAMP(11); else if (ph.type == PH_HEADF) STAMP(12); else STAMP(13);
#ifdef MCPC_STAMPS_ENTRY
            if (MCPC_STAMPS_ENTRY == 3) STAMP_ENTRY_ADD(p, ewait);
            else if (MCPC_STAMPS_ENTRY == 4) STAMP_ENTRY_ADD(p, mcpc_stamp() - ee0);
#endif
        }
    }
    if (e0_in_regs && e0_dirty) lean_flush_e0<CTT>(P, k, LL, e0acc);      // (FWD entries have rot == 0: this wave's tile is k)
#ifdef MCPC_STAMPS
    if (lane == 0 && P.dbg != nullptr)
        for (int i = 0; i < 16; ++i) P.dbg[((size_t)blockIdx.x * 8 + wave8) * 16 + i] = st_sum[i];
#endif
}

}  // namespace mcpc
