// Unified-wave Langevin step kernel ("U", round 6): the form small networks run on -- the reference's own 20-128-128-784 net of every figure
// script (figure_2.py:159-161, figure_3.py:130-132, table_1.py:31-32) at any batch, replacing pc_trainer.py:712-981 + utils/model.py:35-44 per step
// like the in-place kernel (mcpc_steps_ws2.h) does for wide ones.
//
// Why another form.  The in-place kernel splits a workgroup into 4 GEMM and 4 epilogue waves that hand accumulator blocks over through
// LDS: at cfg-M the two roles hide each other, but a step of the small net is a chain of 13 GEMM entries + 10 epilogue entries whose
// cost is the hand-over itself (block store, publish, poll, block load, ~2-4 k cycles per entry and role): 21 us per step whether 1 or
// 4096 chains run it, 10 % of it MFMA time (profiles/r05_k1_bounds.txt item 7, profiles/r06_small_net.txt).  Here EVERY wave of the
// workgroup (8, two per SIMD) computes the GEMM of its own unit tiles and runs their epilogue straight from the accumulator registers
// (the MFMA C layout -- lane (c, q): units 4q..4q+3 of chain c -- is the epilogues' own), so that
//   * no block crosses LDS between a GEMM and its epilogue, nothing is published or polled in between;
//   * the whole read-out error e_o stays in LDS (the plan only exists when it fits: plan_lds_u), so its back-projection is ONE GEMM over
//     K = n_out inside the x update of the last latent layer, not a ring of chunks with an accumulator carried across entries;
//   * a step is two LEVELS, not a chain: every forward job (read-out tiles, FWD_{L-1} .. FWD_0) depends only on the previous step's x
//     updates, every x update only on this step's forward jobs -- two workgroup barriers per step (LDS-scoped fences) and nothing else
//     synchronises; with a barrier on either side of a level no tile belongs to a wave;
//   * every wave walks its OWN table (P.phases[w * n_rows + p]): a row is a JOB -- up to four consecutive unit tiles of one entry = one
//     GEMM call + one epilogue call -- and the host deals a level's jobs to the eight waves by a cost model, longest first, with the grain
//     of every entry chosen for the shortest makespan (mcpc_api.hip: build_phases_u);
//   * with a zero loss (reference utils/model.py:31-33 `zero_fn`: unclamped generation, figure_3.py:125-161) the read-out is dead code on
//     every step whose output is not recorded: e_o = 0, so its back-projection vanishes and `out` is consumed by nothing.  Those steps skip
//     the read-out rows and the back-projection GEMM -- the arithmetic of everything that IS computed is unchanged (the x update adds
//     the same +0), the recorded outputs are those of the steps they are recorded at.  (A step that spills for the Hebbian sums keeps the
//     read-out: its zeros are what the flush must see.)
//   * a Bernoulli read-out's error (bounded: a constant fp16 scale) goes to LDS already split into the two fp16 planes its back-projection
//     reads (mcpc_ws2_lean.h: lean_headf, headf_planes) -- once per value instead of once per k-block, wave and step.
// What was measured on the way and is NOT here (profiles/r06_small_net.txt): per-entry progress counters instead of the barriers (25.5
// against 24.4 us per step in the first table form); every level's B operands split once by the whole workgroup between two more barriers
// (the GEMMs 20 % shorter, the pass and its barrier dearer: 19.5 against 18.8); the jobs of a level taken by the waves at RUN TIME (one
// LDS atomic per job, longest first, per-job energy slots): level with the host's deal on the reference's net (16.4 against 16.2) and
// much slower on nets of many short jobs (30-200-72-100: 16.0 against 10.1 us) -- a job's descriptor can then be requested only when the
// job before it has been taken, and its scalar-load latency lands in front of the short rows.
// Arithmetic: the GEMM core (mcpc_gemm_f16.h) and the lean epilogues (mcpc_ws2_lean.h, REG = true) of the in-place kernel, operation for
// operation -- trajectories are bitwise those of the other kernel forms (tests/test_gpu_unified.py); energies differ in the order their
// per-wave partial sums are formed (8 waves' shares instead of 4) and agree to rounding.
// Only the lean paths run here (fused SGD update with or without the Philox kick, Adam without noise; state in LDS): mcpc_run picks this
// kernel per run and keeps the in-place / barrier kernels for everything else.
#pragma once

namespace mcpc {

constexpr int kUWaves = 8;                      // waves per workgroup, all alike (two per SIMD)
constexpr int kUNT = 4;                         // unit tiles per row (job) at most: the four fragment slots and accumulator tiles of a wave
constexpr int kUThreads = kUWaves * 64;

// request the first fragments of an upcoming row's GEMM (weights need no dependency), into the four slots the GEMM forms expect
// (mcpc_gemm_f16.h: gemm_tiles_u): three or four tiles -- block 0 of every tile; one or two tiles -- the first 4 / nt blocks of each
// (slot j nt + i = block j of tile i).  Branch-free like ws2_prefetch: slots without a fragment (a tile the row does not have, a block
// beyond the GEMM, a row without a GEMM) read 1 KiB of `dummy` -- L1 hits, no fill traffic.  Tiles of a row: tile0 + rot i, i < ntiles.
__device__ __forceinline__ void u_prefetch(const KPhase& ph, int lane, const void* dummy, int& nt_out, int (&aoff)[kUNT], frag_t (&pre0)[kUNT]) {
    int nt = ph.ntiles;
    nt = nt < 0 ? 0 : (nt > kUNT ? kUNT : nt);
    nt_out = nt;
    const bool valid = nt > 0 && (ph.flags & PHF_WS_GEMM) && ph.nkb > 0;
#pragma unroll
    for (int i = 0; i < kUNT; ++i) aoff[i] = valid ? (ph.tile0 + ph.rot * (i < nt ? i : 0)) * ph.a_tile_stride + ph.a_off0 : 0;
    // the slots were resolved by the host (build_phases_u): offsets from ph.A in 16-byte units, -1 = no fragment
    const int slot[kUNT] = {ph.dep_e, ph.dep_g, ph.dep_se, ph.next_g};
#pragma unroll
    for (int sl = 0; sl < kUNT; ++sl) {
        const bool have = slot[sl] >= 0;
        pre0[sl] = load_frag(have ? (const gu32x4*)ph.A : (const gu32x4*)dummy, have ? slot[sl] : 0, lane);
    }
}

// level boundary: a workgroup barrier that orders LDS accesses only (a full __syncthreads() also drains the wave's global loads -- the
// fragments requested for the next entry -- and the Hebbian spill stores: s_waitcnt vmcnt(0) twice per step)
__device__ __forceinline__ void u_barrier() {
    MCPC_WS_FENCE(__ATOMIC_RELEASE);
    __builtin_amdgcn_s_barrier();
    MCPC_WS_FENCE(__ATOMIC_ACQUIRE);
}

// step s (global step t) of this workgroup: the eight waves' partial sums -> one fp64 row of the energy partials (fixed order)
__device__ __forceinline__ void u_energy_row(const KParams& P, const float* lds, int s, int t, int unit, int lane, int L, bool has_head) {
    const float* const red = lds + P.lds_red + (s & 1) * (kMaxLatent + 1) * kMaxWaves;
    if (lane <= kMaxLatent) {
        double v = 0.0;
        const bool used = (lane < L) || (lane == kMaxLatent && has_head);
        if (used) {
#pragma unroll
            for (int ww = 0; ww < kUWaves; ++ww) v += (double)red[lane * kMaxWaves + ww];
        }
        const int erow = (P.energy_mode == MCPC_ENERGY_ALL) ? t : 0;
        P.epart[((size_t)erow * P.epart_slots + (size_t)unit) * (kMaxLatent + 1) + lane] = v;
    }
}

// the epilogue of a row, straight from the accumulators (mcpc_ws2_lean.h, REG = true).  PLAIN: the caller passes slot = rec_idx = -1 and
// do_energy = false as literals -- after inlining, the spill, record and energy code of the epilogues is gone from that instantiation.
template <bool MIX, bool PLAIN>
__device__ __forceinline__ void u_epilogue(const KParams& P, const KPhase& ph, float* lds, int nt, const LeanLane<1>& LL, int slot, int rec_idx,
                                           bool do_energy, int t, int s_tab, float* rx, unsigned row_gen, const f32x4 (&acc)[kUNT][1],
                                           f32x4 (&e0acc)[1], bool e0_in_regs, bool& e0_dirty, float& en_acc, bool ybin, bool hplanes, bool lean_adam,
                                           int upd_mode, int lane, bool yw_glob, const uint32_t (&ywreg)[kUNT]) {
    constexpr int CTT = 1, NW = kUWaves, NTW = kUNT;
    int dead = 0;
    const int act = P.layer[ph.layer].act;
    if (ph.type == PH_FWD) {
        float esum;
        if (!PLAIN && ph.layer == 0 && slot >= 0) e0_dirty = true;
        if (act == MCPC_ACT_RELU) esum = lean_fwd<CTT, NW, NTW, MCPC_ACT_RELU, true, true>(P, ph, lds, nt, 0, LL, slot, rec_idx, nullptr, 0, P.err, dead, e0acc, e0_in_regs, rx, row_gen, acc);
        else if (act == MCPC_ACT_TANH) esum = lean_fwd<CTT, NW, NTW, MCPC_ACT_TANH, true, true>(P, ph, lds, nt, 0, LL, slot, rec_idx, nullptr, 0, P.err, dead, e0acc, e0_in_regs, rx, row_gen, acc);
        else esum = lean_fwd<CTT, NW, NTW, MCPC_ACT_IDENTITY, true, true>(P, ph, lds, nt, 0, LL, slot, rec_idx, nullptr, 0, P.err, dead, e0acc, e0_in_regs, rx, row_gen, acc);
        if (do_energy) { esum = wave_sum(esum); if (lane == ph.layer) en_acc += esum; }
    } else if (ph.type == PH_HEADF) {
        float lsum;
        if (ybin && yw_glob) lsum = lean_headf<CTT, NW, NTW, true, true, true, true>(P, ph, lds, nt, 0, LL, slot, rec_idx, do_energy, nullptr, 0, P.err, dead, true, rx, acc, row_gen, hplanes, ywreg);
        else if (ybin) lsum = lean_headf<CTT, NW, NTW, true, true, true>(P, ph, lds, nt, 0, LL, slot, rec_idx, do_energy, nullptr, 0, P.err, dead, true, rx, acc, row_gen, hplanes);
        else lsum = lean_headf<CTT, NW, NTW, true, false, true>(P, ph, lds, nt, 0, LL, slot, rec_idx, do_energy, nullptr, 0, P.err, dead, false, rx, acc, row_gen, hplanes);
        if (do_energy) { lsum = wave_sum(lsum); if (lane == kMaxLatent) en_acc += lsum; }
    } else if (ph.type == PH_BWD) {
        if (lean_adam) {
            if (act == MCPC_ACT_RELU) lean_bwd<CTT, NW, NTW, MCPC_ACT_RELU, false, true, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, s_tab, rx, row_gen, acc);
            else if (act == MCPC_ACT_TANH) lean_bwd<CTT, NW, NTW, MCPC_ACT_TANH, false, true, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, s_tab, rx, row_gen, acc);
            else lean_bwd<CTT, NW, NTW, MCPC_ACT_IDENTITY, false, true, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, s_tab, rx, row_gen, acc);
        } else if (upd_mode == 2) {
            if (act == MCPC_ACT_RELU) lean_bwd<CTT, NW, NTW, MCPC_ACT_RELU, true, false, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, 0, rx, row_gen, acc);
            else if (act == MCPC_ACT_TANH) lean_bwd<CTT, NW, NTW, MCPC_ACT_TANH, true, false, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, 0, rx, row_gen, acc);
            else lean_bwd<CTT, NW, NTW, MCPC_ACT_IDENTITY, true, false, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, 0, rx, row_gen, acc);
        } else {
            if (act == MCPC_ACT_RELU) lean_bwd<CTT, NW, NTW, MCPC_ACT_RELU, false, false, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, 0, rx, row_gen, acc);
            else if (act == MCPC_ACT_TANH) lean_bwd<CTT, NW, NTW, MCPC_ACT_TANH, false, false, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, 0, rx, row_gen, acc);
            else lean_bwd<CTT, NW, NTW, MCPC_ACT_IDENTITY, false, false, true, true>(P, ph, lds, nt, 0, LL, t, nullptr, 0, P.err, dead, 0, rx, row_gen, acc);
        }
    }
}

template <bool MIX>
__global__ __launch_bounds__(kUThreads, 2) void mcpc_steps_u_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CTT = 1, NW = kUWaves, NTW = kUNT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int unit = MIX ? P.wg_list[blockIdx.x] : (int)blockIdx.x;
    const int chain0 = unit * 16;
    const int t_first = MIX ? P.t0 + P.wg_rel[blockIdx.x] * P.rr_q : P.t0;
    const int L = P.L;
    const int n_ent = P.n_phases;
    for (int i = tid; i < P.lds_floats / 4; i += kUThreads) st4(lds + 4 * i, splat(0.f));
    __syncthreads();
    // (the host launches this kernel for lean runs of a plan with the state in LDS only: mcpc_run)
    const int upd_mode = P.xopt == MCPC_XOPT_SGD ? (P.noise_mode == MCPC_NOISE_PHILOX ? 2 : 1) : 0;
    const bool lean_adam = P.xopt == MCPC_XOPT_ADAM;
    float* const rx = lds + P.lds_rowexp;
    for (int l = 0; l < L; ++l) {
        const KLayer& Ly = P.layer[l];
        if (Ly.act == MCPC_ACT_RELU) ws2_fill_fx<MCPC_ACT_RELU>(Ly, lds, chain0, tid, 16, true, rx, rowexp_fx(l));
        else if (Ly.act == MCPC_ACT_TANH) ws2_fill_fx<MCPC_ACT_TANH>(Ly, lds, chain0, tid, 16, true, rx, rowexp_fx(l));
        else ws2_fill_fx<MCPC_ACT_IDENTITY>(Ly, lds, chain0, tid, 16, true, rx, rowexp_fx(l));
    }
    ws2_fill_constants(P, lds, chain0, tid, 16);
    __syncthreads();

    const bool clk_wave = P.clk != nullptr && blockIdx.x == 0 && w == 0;
    unsigned long long clk_c0 = 0, clk_w0 = 0;
    if (clk_wave) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_w0 = __builtin_amdgcn_s_memrealtime(); }
    LeanLane<CTT> LL;
    LL.c = c; LL.q = q;
    LL.chain[0] = (uint32_t)(chain0 + c); LL.lrow[0] = (uint32_t)c;
    LL.livem[0] = chain0 + c < P.B ? ~0u : 0u;
    const bool has_head = P.has_head != 0;
    const bool ybin = has_head && *P.head.y_binary != 0;
    const bool y_bounded = has_head && *P.head.y_bounded != 0;      // (headb_fixed_exp: the bound target lies in [-1, 2])
    const bool hplanes = has_head && headf_planes(P.head.loss_kind, y_bounded, P.head.npad);   // the read-out error travels as fp16 planes
    // a plan without the room for the bit-packed target rows (cfg-M): a 0/1 target's words come from global memory, requested in front of
    // the row's GEMM so that they are older than every fragment load the epilogue would otherwise wait behind
    const bool yw_glob = has_head && P.head.lds_yw < 0 && ybin;
    const bool no_loss = has_head && P.head.loss_kind == MCPC_LOSS_NONE;
    const bool e0_in_regs = P.layer[0].ntiles <= NW;
    bool e0_dirty = false;
    f32x4 e0acc[CTT];
    e0acc[0] = splat(0.f);
    if (e0_in_regs && t_first + P.n_steps > P.acc_begin && t_first < P.acc_end) lean_load_e0<CTT>(P, w, LL, e0acc);

    // the exponents the packed weights were scaled by (one per Linear), once: lane j of a VGPR holds Linear j's -- a v_readlane per GEMM
    // where a scalar load from device memory sat in front of every GEMM's first block (its latency exposed: the exponent is needed at once)
    const int wexp_v = lane <= kMaxLatent ? P.wexp[lane] : 0;
    // loop-carried: the descriptor of the upcoming row and the first fragments of its GEMM (one prefetch site, as in the in-place
    // kernel's GEMM waves: mcpc_steps_ws2_body.inc).  Every wave walks its OWN rows: P.phases[w * n_ent .. + n_ent).
    const KPhase* const tab = P.phases + (size_t)w * n_ent;
    KPhase ph_next = load_phase(tab, 0);
    int nt_next, aoff[NTW];
    frag_t pre0[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) { pre0[i] = frag_zero(); aoff[i] = 0; }
    u_prefetch(ph_next, lane, P.dummy, nt_next, aoff, pre0);

    bool energy_prev = false;                  // the step before this one left partial energy sums in LDS
    STAMP_DECL
    for (int s = 0; s < P.n_steps; ++s) {
        const int t = t_first + s;
        const int s_tab = MIX ? t - P.t0 : s;
        const bool do_energy = (P.energy_mode == MCPC_ENERGY_ALL) || (P.energy_mode == MCPC_ENERGY_LAST && t == P.T - 1);
        const int slot = (t >= P.acc_begin && t < P.acc_end) ? (t - P.spill_t0) : -1;
        int rec_idx = -1;
        if (P.rec_count > 0 && t >= P.rec_begin) {
            const int kk = (t - P.rec_begin) / P.rec_stride;
            if (kk < P.rec_count && P.rec_begin + kk * P.rec_stride == t) rec_idx = kk;
        }
        // zero loss: the read-out is computed on the steps that record it (or spill it), nowhere else -- see the header
        const bool skip_head = no_loss && slot < 0 && (rec_idx < 0 || P.head.rec_out == nullptr);
        const bool plain = slot < 0 && rec_idx < 0 && !do_energy;
        float* const red = lds + P.lds_red + (s & 1) * (kMaxLatent + 1) * kMaxWaves;
        float en_acc = 0.f;
        const unsigned row_gen = (unsigned)(s + 1);
#pragma unroll 1
        for (int p = 0; p < n_ent; ++p) {
            KPhase ph = ph_next;
            const int nt = nt_next;
            const int pn = p + 1 < n_ent ? p + 1 : 0;
            if (pn != 0 || s + 1 < P.n_steps) ph_next = load_phase(tab, pn);
            f32x4 acc[NTW][CTT];
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i][0] = splat(0.f);
            STAMP(0);
            // level boundary: what this level reads is complete, what it writes is no longer read
            if (ph.flags & PHF_SYNC) {
                u_barrier();
                STAMP(1);
                // the previous step's energies: every wave's partial sums are in LDS now (the other half of `red`)
                if (p == 0 && s > 0 && w == 0 && energy_prev) u_energy_row(P, lds, s - 1, t - 1, unit, lane, L, has_head);
            }
            // rows this step does not need (zero loss: read-out chunks; the read-out's back-projection)
            const bool head_gemm = ph.type == PH_BWD && ph.a_lin == L && (ph.flags & PHF_WS_GEMM);
            if (head_gemm && no_loss) ph.flags &= ~PHF_WS_GEMM;
            const bool live = nt > 0 && !(ph.type == PH_HEADF && skip_head);
            uint32_t ywreg[NTW] = {0u, 0u, 0u, 0u};
            if (yw_glob && ph.type == PH_HEADF) {
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const int tile = ph.tile0 + ph.rot * (i < nt ? i : 0);
                    ywreg[i] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(
                        (gbytes_t)P.head.ybits + mul24(LL.chain[0], 4u * (uint32_t)P.head.ywords) + 4u * (uint32_t)(tile >> 1));
                }
            }
            if (live && (ph.flags & PHF_WS_GEMM) && ph.nkb > 0) {
                int fixed_b;
                int short_k = -1;
                if (head_gemm) {
                    const int hb_exp = headb_fixed_exp(P.head.loss_kind, y_bounded);
                    fixed_b = hb_exp == kScaleAuto ? rowexp_read(rx, ph.b_row, c) : hb_exp;
                    short_k = P.head.npad <= kShortK * kKB ? 1 : 0;
                } else {
                    fixed_b = rowexp_read(rx, ph.b_row, c);
                }
                gemm_tiles_u<NTW>(acc, ph.A, aoff, nt, ph.nkb, ph.kw, lds + ph.b_lds, ph.ldb, lane, pre0, lds + P.lds_zero,
                                  __builtin_amdgcn_readlane(wexp_v, ph.a_lin), fixed_b, short_k, head_gemm && hplanes);
            }
#ifdef MCPC_STAMPS        // GEMM time by row kind: slots 8..11 forward rows of 1..4 tiles, 12..15 update rows of 1..4 tiles
            {
                const unsigned long long now_ = mcpc_stamp(), d_ = now_ - st_last;
                st_last = now_;
                switch ((ph.type == PH_BWD ? 4 : 0) + (nt > 0 ? nt - 1 : 0)) {
                    case 0: st_sum[8] += d_; break; case 1: st_sum[9] += d_; break; case 2: st_sum[10] += d_; break; case 3: st_sum[11] += d_; break;
                    case 4: st_sum[12] += d_; break; case 5: st_sum[13] += d_; break; case 6: st_sum[14] += d_; break; default: st_sum[15] += d_; break;
                }
            }
#endif
            // fragments of the next row travel while this one's epilogue runs
            u_prefetch(ph_next, lane, P.dummy, nt_next, aoff, pre0);
            STAMP(3);
            if (live) {
                // (ordinary steps -- nothing spilled, nothing recorded, no energies -- run instantiations in which all of that is compiled out)
                if (plain) u_epilogue<MIX, true>(P, ph, lds, nt, LL, -1, -1, false, t, s_tab, rx, row_gen, acc, e0acc, e0_in_regs, e0_dirty, en_acc, ybin, hplanes, lean_adam, upd_mode, lane, yw_glob, ywreg);
                else u_epilogue<MIX, false>(P, ph, lds, nt, LL, slot, rec_idx, do_energy, t, s_tab, rx, row_gen, acc, e0acc, e0_in_regs, e0_dirty, en_acc, ybin, hplanes, lean_adam, upd_mode, lane, yw_glob, ywreg);
                STAMP(4);
            }
        }
        // this wave's share of the step's energies (lane l: layer l, lane kMaxLatent: the loss)
        if (do_energy && lane <= kMaxLatent) red[lane * kMaxWaves + w] = en_acc;
        energy_prev = do_energy;
    }
    // every wave's last x update is in LDS (and its partial energy sums)
    u_barrier();
    if (energy_prev && w == 0) u_energy_row(P, lds, P.n_steps - 1, t_first + P.n_steps - 1, unit, lane, L, has_head);
    lean_store_x<CTT, NW>(P, lds, w, LL);                       // the state of this wave's tiles goes back to global memory
    spill_max_publish(P.spillmax, lds + P.lds_spillmax, lane);
    if (clk_wave && lane == 0) {
        atomicAdd(P.clk, __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(P.clk + 1, __builtin_amdgcn_s_memrealtime() - clk_w0);
    }
    if (e0_in_regs && e0_dirty) lean_flush_e0<CTT>(P, w, LL, e0acc);
#ifdef MCPC_STAMPS
    if (lane == 0 && P.dbg != nullptr)
        for (int i = 0; i < 16; ++i) P.dbg[((size_t)blockIdx.x * kUWaves + w) * 16 + i] = st_sum[i];
#endif
}

}  // namespace mcpc
