// Wave-specialised variant of the Langevin step kernel (selected with MCPC_WS=1; 32 chains per workgroup).
//
// A workgroup has 8 waves: waves 0-3 ("G", one per SIMD) only stream weight fragments and issue MFMAs,
// waves 4-7 ("E", the other wave of each SIMD) only run epilogues: operand loads (x, targets), errors,
// energies, activations, loss, x updates with the fused Philox noise, spills and records.  Pair k = (G_k, E_k)
// owns the unit tiles tile0 + k + 4 i (i < 2) of every table entry; G_k hands each finished accumulator
// block to E_k through a 4 KiB LDS staging slot and moves on to the next GEMM whose operands are ready,
// so the matrix pipe keeps running while the VALU work of the previous GEMM is done by the partner wave.
// There is no s_barrier in the step loop: every dependency is a monotonic progress counter in LDS
//   prog_e[k] / prog_g[k]  = table entries completed by E_k / G_k (absolute: step * n_entries + index + 1)
//   stage_full[k] / stage_empty[k] = hand-offs written by G_k / consumed by E_k
// polled by one ds_read per wait.  Spins are bounded (a broken schedule yields wrong results, which the
// parity tests catch, never a hung GPU).
#pragma once

namespace mcpc {

constexpr int kWsPairs = 4;            // (G, E) pairs per workgroup = tile stride
constexpr int kWsNT = 2;               // unit tiles per pair per table entry (an entry hands out 8 tiles)
#ifndef MCPC_WS_SLEEP
#define MCPC_WS_SLEEP 2
#endif
constexpr int kWsSpinLimit = 1 << 22;  // iterations (~0.1-0.2 us each): ~0.5 s, far beyond any legitimate wait (< 1 ms)

enum : int { PHF_WS_GEMM = 16, PHF_WS_EPI = 32 };   // which role has work in a table entry

// Everything the two roles hand to each other lives in LDS (global data written by an E wave -- x, spills, energies --
// is only re-read by the same lane or by later kernels), so the hand-off fences order LDS accesses only: a full
// workgroup-scope release would also drain every outstanding global store (`s_waitcnt vmcnt(0)`) at each publish.
#ifdef MCPC_EXP_FULLFENCE
#define MCPC_WS_FENCE(order_) __builtin_amdgcn_fence(order_, "workgroup")
#else
#define MCPC_WS_FENCE(order_) __builtin_amdgcn_fence(order_, "workgroup", "local")
#endif

struct WsSync {                        // lives in LDS
    int prog_e[4];
    int prog_g[4];
    int stage_full[4];
    int stage_empty[4];
};

__device__ __forceinline__ int ws_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ws_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// wait until all four counters are >= need; a spin that runs out records the fact in *err (host: mcpc_sync_check)
// (a wave whose wait ran out once stops waiting altogether -- `dead` -- so a broken schedule drains in about one
// spin limit instead of one per remaining table entry)
__device__ __forceinline__ void ws_wait_all(const int* p, int need, int* err, int& dead) {
    int spin = dead ? kWsSpinLimit : 0;
    for (; spin < kWsSpinLimit; ++spin) {
        const int a = ws_ld(p), b = ws_ld(p + 1), c = ws_ld(p + 2), d = ws_ld(p + 3);
        if (min(min(a, b), min(c, d)) >= need) break;
        if (MCPC_WS_SLEEP > 0) __builtin_amdgcn_s_sleep(MCPC_WS_SLEEP);
    }
    if (spin == kWsSpinLimit) { if (!dead && (threadIdx.x & 63) == 0) atomicOr(err, 1); dead = 1; }
    MCPC_WS_FENCE(__ATOMIC_ACQUIRE);
}
__device__ __forceinline__ void ws_wait_one(const int* p, int need, int* err, int& dead) {
    int spin = dead ? kWsSpinLimit : 0;
    for (; spin < kWsSpinLimit; ++spin) {
        if (ws_ld(p) >= need) break;
        if (MCPC_WS_SLEEP > 0) __builtin_amdgcn_s_sleep(MCPC_WS_SLEEP);
    }
    if (spin == kWsSpinLimit) { if (!dead && (threadIdx.x & 63) == 0) atomicOr(err, 2); dead = 1; }
    MCPC_WS_FENCE(__ATOMIC_ACQUIRE);
}
__device__ __forceinline__ void ws_publish(int* p, int v) {
    MCPC_WS_FENCE(__ATOMIC_RELEASE);     // LDS writes of this wave before the counter
    ws_st(p, v);
}

// G side: request the first two k-blocks of the fragments of an upcoming entry (weights need no dependency)
__device__ __forceinline__ void ws_prefetch(const KPhase& ph, int k, int lane, int& nt_out, int (&aoff)[4],
                                            f32x4 (&pre0)[4], f32x4 (&pre1)[4]) {
    const int ntmax = (ph.type == PH_HEADB) ? 4 : kWsNT;
    int nt = (ph.ntiles - k + kWsPairs - 1) / kWsPairs;
    nt = nt < 0 ? 0 : (nt > ntmax ? ntmax : nt);
    if (!(ph.flags & PHF_WS_GEMM) || ph.nkb <= 0) nt = 0;
    nt_out = nt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = (ph.tile0 + k + kWsPairs * i) * ph.a_tile_stride + ph.a_off0;
        if (i < nt) {
            const gf32x4* A = (const gf32x4*)ph.A;
            pre0[i] = A[aoff[i] + lane];
            if (ph.nkb > 1) pre1[i] = A[aoff[i] + 64 + lane];
        }
    }
}

template <int CTT>
__global__ __launch_bounds__(512, 2) void mcpc_steps_ws_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = kWsPairs, NTW = kWsNT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_g = wave8 < 4;
    const int k = wave8 & 3;                               // pair id: tile owner index
    const int c = lane & 15, q = lane >> 4;
    const int chain0 = blockIdx.x * (16 * CTT);
    const int L = P.L;
    const int n_ent = P.n_phases;
    WsSync* sync = reinterpret_cast<WsSync*>(lds + P.lds_ws_sync);
    float* stage = lds + P.lds_ws_stage + k * (kWsNT * CTT * 64 * 4);
    int dead = 0;                                          // set once a bounded wait of this wave ran out
    if (tid < 16) reinterpret_cast<int*>(sync)[tid] = 0;
    __syncthreads();                                       // the only barrier: counters start at zero

    if (is_g) {
        // =========================== G: fragments + MFMAs ==============================================
        KPhase ph_next = load_phase(P.phases, 0);
        int nt_next, aoff_next[4];
        f32x4 pre0_next[4], pre1_next[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { pre0_next[i] = splat(0.f); pre1_next[i] = splat(0.f); aoff_next[i] = 0; }
        ws_prefetch(ph_next, k, lane, nt_next, aoff_next, pre0_next, pre1_next);
        int handoffs = 0;                                   // staging blocks written so far
        STAMP_DECL
        for (int s = 0; s < P.n_steps; ++s) {
            const int base = s * n_ent;
            f32x4 accb[4][CTT];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ct = 0; ct < CTT; ++ct) accb[i][ct] = splat(0.f);
#pragma unroll 1
            for (int p = 0; p < n_ent; ++p) {
                const KPhase ph = ph_next;
                const int nt = nt_next;
                int aoff4[4];
                f32x4 pre0[4], pre1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { aoff4[i] = aoff_next[i]; pre0[i] = pre0_next[i]; pre1[i] = pre1_next[i]; }
                const bool has_next = (p + 1 < n_ent) || (s + 1 < P.n_steps);
                if (has_next) ph_next = load_phase(P.phases, p + 1 < n_ent ? p + 1 : 0);
                const bool upd_only = (ph.type == PH_BWD && ph.nkb == 0);       // hand accb to the partner
                if (!(ph.flags & PHF_WS_GEMM) && !upd_only) {
                    if (has_next) ws_prefetch(ph_next, k, lane, nt_next, aoff_next, pre0_next, pre1_next);
                    if (lane == 0) ws_publish(&sync->prog_g[k], base + p + 1);
                    continue;
                }
                STAMP(0);
                if (ph.dep_e >= 0) ws_wait_all(sync->prog_e, base + ph.dep_e + 1, P.err, dead);
                STAMP(1);
                if (ph.type == PH_HEADB) {
                    if (nt > 0) gemm_tiles<4, CTT, NW>(accb, (const gf32x4*)ph.A, aoff4, nt, ph.nkb, lds + ph.b_lds, ph.ldb, lane, pre0, pre1);
                    STAMP(2);
                    if (has_next) ws_prefetch(ph_next, k, lane, nt_next, aoff_next, pre0_next, pre1_next);
                    if (lane == 0) ws_publish(&sync->prog_g[k], base + p + 1);
                    continue;
                }
                f32x4 acc[kWsNT][CTT];
                const int sub = ph.tile0 / (NW * kWsNT);                         // which pair of accb tiles (UPD)
#pragma unroll
                for (int i = 0; i < kWsNT; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTT; ++ct)
                        acc[i][ct] = (ph.flags & PHF_ACC_FROM_B) ? (sub == 0 ? accb[i][ct] : accb[i + 2][ct]) : splat(0.f);
                if (nt > 0 && ph.nkb > 0) {
                    int aoff2[kWsNT];
                    f32x4 p0[kWsNT], p1[kWsNT];
#pragma unroll
                    for (int i = 0; i < kWsNT; ++i) { aoff2[i] = aoff4[i]; p0[i] = pre0[i]; p1[i] = pre1[i]; }
                    gemm_tiles<kWsNT, CTT, NW>(acc, (const gf32x4*)ph.A, aoff2, nt, ph.nkb, lds + ph.b_lds, ph.ldb, lane, p0, p1);
                }
                STAMP(3);
                if (has_next) ws_prefetch(ph_next, k, lane, nt_next, aoff_next, pre0_next, pre1_next);
                // hand the block to E_k: wait until the previous block was copied out, write, signal
                ws_wait_one(&sync->stage_empty[k], handoffs, P.err, dead);
                STAMP(4);
#pragma unroll
                for (int i = 0; i < kWsNT; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTT; ++ct) st4(stage + ((i * CTT + ct) * 64 + lane) * 4, acc[i][ct]);
                ++handoffs;
                if (lane == 0) {
                    ws_publish(&sync->stage_full[k], handoffs);
                    ws_st(&sync->prog_g[k], base + p + 1);
                }
                STAMP(5);
            }
        }
#ifdef MCPC_STAMPS
        if (lane == 0)
            for (int i = 0; i < 16; ++i) P.dbg[((size_t)blockIdx.x * 8 + wave8) * 16 + i] = st_sum[i];
#endif
        return;
    }

    // =============================== E: epilogues ===========================================================
    // The E wave is the younger wave of its SIMD: at equal priority its VALU instructions only get the issue
    // slots the MFMA-issuing partner leaves over (MI355X_MICROARCH.md, "Two waves per SIMD", items 2 and 4).
    // An MFMA needs the issue port 8 cycles out of 32, so giving E the higher static priority costs G nothing.
    if (P.ws_prio > 0) __builtin_amdgcn_s_setprio(2);
    const int upd_mode = (P.update_x && P.xopt == MCPC_XOPT_SGD)
                             ? (P.noise_mode == MCPC_NOISE_PHILOX ? 2 : (P.noise_mode == MCPC_NOISE_NONE ? 1 : 0)) : 0;
    int handoffs = 0;
    STAMP_DECL
    for (int s = 0; s < P.n_steps; ++s) {
        const int t = P.t0 + s;
        const int base = s * n_ent;
        const bool do_energy = (P.energy_mode == MCPC_ENERGY_ALL) || (P.energy_mode == MCPC_ENERGY_LAST && t == P.T - 1);
        const int slot = (t >= P.acc_begin && t < P.acc_end) ? (t - P.spill_t0) : -1;
        int rec_idx = -1;
        if (P.rec_count > 0 && t >= P.rec_begin) {
            const int kk = (t - P.rec_begin) / P.rec_stride;
            if (kk < P.rec_count && P.rec_begin + kk * P.rec_stride == t) rec_idx = kk;
        }
        float* red = lds + P.lds_red + (s & 1) * (kMaxLatent + 1) * kMaxWaves;
        if (do_energy && lane <= kMaxLatent) red[lane * kMaxWaves + k] = 0.f;
#pragma unroll 1
        for (int p = 0; p < n_ent; ++p) {
            const KPhase ph = load_phase(P.phases, p);
            if (ph.type == PH_ENERGY) {
                if (do_energy && k == 0) {
                    // every E wave has finished the forward entries of this step (their red[] slots are final)
                    ws_wait_all(sync->prog_e + 0, base + p, P.err, dead);     // own counter equals base + p already
                    if (lane <= kMaxLatent) {
                        double v = 0.0;
                        const bool used = (lane < L) || (lane == kMaxLatent && P.has_head);
                        if (used) {
#pragma unroll
                            for (int w = 0; w < kWsPairs; ++w) v += (double)red[lane * kMaxWaves + w];
                        }
                        const int erow = (P.energy_mode == MCPC_ENERGY_ALL) ? t : 0;
                        P.epart[((size_t)erow * gridDim.x + blockIdx.x) * (kMaxLatent + 1) + lane] = v;
                    }
                }
                if (lane == 0) ws_publish(&sync->prog_e[k], base + p + 1);
                continue;
            }
            if (!(ph.flags & PHF_WS_EPI)) {
                if (lane == 0) ws_publish(&sync->prog_e[k], base + p + 1);
                continue;
            }
            int nt = (ph.ntiles - k + NW - 1) / NW;
            nt = nt < 0 ? 0 : (nt > kWsNT ? kWsNT : nt);
            f32x4 acc[kWsNT][CTT], pa[kWsNT][CTT], pb[kWsNT][CTT];
#pragma unroll
            for (int i = 0; i < kWsNT; ++i)
#pragma unroll
                for (int ct = 0; ct < CTT; ++ct) { acc[i][ct] = splat(0.f); pa[i][ct] = splat(0.f); pb[i][ct] = splat(0.f); }
            const KLayer& Ly = P.layer[ph.layer];
            // operands of the epilogue travel while the partner still computes
            STAMP(8);
            issue_epilogue_loads<CTT, NW, NTW>(P, ph, lds, nt, k, lane, chain0, pa, pb);
            if (ph.dep_g >= 0) ws_wait_all(sync->prog_g, base + ph.dep_g + 1, P.err, dead);   // e.g. the e_o chunk is free again
            STAMP(9);
            const bool from_g = (ph.nkb > 0) || (ph.type == PH_BWD);              // G hands a block for this entry
            if (from_g) {
                ws_wait_one(&sync->stage_full[k], handoffs + 1, P.err, dead);
                STAMP(10);
#pragma unroll
                for (int i = 0; i < kWsNT; ++i)
#pragma unroll
                    for (int ct = 0; ct < CTT; ++ct) acc[i][ct] = ld4(stage + ((i * CTT + ct) * 64 + lane) * 4);
                ++handoffs;
                if (lane == 0) ws_publish(&sync->stage_empty[k], handoffs);
            }
#ifdef MCPC_EXP_NOEPI   // timing experiment only (wrong results): E waves skip the epilogue arithmetic and stores
            if (P.n_steps < 0)
#endif
            if (ph.type == PH_FWD) {
                float esum;
                if (Ly.act == MCPC_ACT_RELU) esum = fwd_epilogue<CTT, NW, NTW, MCPC_ACT_RELU>(P, ph, lds, nt, k, lane, chain0, acc, pa, pb, slot, rec_idx);
                else if (Ly.act == MCPC_ACT_TANH) esum = fwd_epilogue<CTT, NW, NTW, MCPC_ACT_TANH>(P, ph, lds, nt, k, lane, chain0, acc, pa, pb, slot, rec_idx);
                else esum = fwd_epilogue<CTT, NW, NTW, MCPC_ACT_IDENTITY>(P, ph, lds, nt, k, lane, chain0, acc, pa, pb, slot, rec_idx);
                if (do_energy) { esum = wave_sum(esum); if (lane == 0) red[ph.layer * kMaxWaves + k] += esum; }
            } else if (ph.type == PH_HEADF) {
                float lsum = headf_epilogue<CTT, NW, NTW>(P, ph, lds, nt, k, lane, chain0, acc, pa, pb, slot, rec_idx, do_energy);
                if (do_energy) { lsum = wave_sum(lsum); if (lane == 0) red[kMaxLatent * kMaxWaves + k] += lsum; }
            } else if (ph.type == PH_BWD) {
                if (Ly.act == MCPC_ACT_RELU) bwd_epilogue_mode<CTT, NW, NTW, MCPC_ACT_RELU>(P, ph, nt, k, lane, chain0, acc, pa, pb, s, t, upd_mode);
                else if (Ly.act == MCPC_ACT_TANH) bwd_epilogue_mode<CTT, NW, NTW, MCPC_ACT_TANH>(P, ph, nt, k, lane, chain0, acc, pa, pb, s, t, upd_mode);
                else bwd_epilogue_mode<CTT, NW, NTW, MCPC_ACT_IDENTITY>(P, ph, nt, k, lane, chain0, acc, pa, pb, s, t, upd_mode);
            }
            if (lane == 0) ws_publish(&sync->prog_e[k], base + p + 1);
            STAMP(11);
        }
    }
#ifdef MCPC_STAMPS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) P.dbg[((size_t)blockIdx.x * 8 + wave8) * 16 + i] = st_sum[i];
#endif
}

}  // namespace mcpc
