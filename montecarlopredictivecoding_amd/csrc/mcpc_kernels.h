// MCPC kernels for gfx950 (MI355X).  See DESIGN.md for the data layout and the per-kernel rooflines.
//
// K1  mcpc_steps_kernel   fused Langevin / PC inference steps, persistent over n_steps:
//                         one workgroup = kCT chains, all layers, weights streamed from L2 in
//                         MFMA-fragment order, state in HBM/L2 (padded), activations/errors in LDS.
//                         replaces reference pc_trainer.py:733-918 + utils/model.py:35-44 per step.
// K3  mcpc_dw_kernel      Hebbian sums  G[u][i] = sum_r E[r][u] * A[r][i]  over spilled (error,
//                         activation) rows, split-K with deterministic slab reduction.
//                         replaces the parameter part of overall.backward(), pc_trainer.py:862.
#pragma once
#include "mcpc_device.h"
#include "../../include/mcpc.h"

namespace mcpc {

constexpr int kMaxLatent = MCPC_MAX_LATENT;
constexpr int kCT = 32;           // chains per workgroup = 2 MFMA column tiles of 16
constexpr int kWaves = 4;         // one wave per SIMD
constexpr int kThreads = kWaves * 64;
constexpr int kNT = 4;            // unit tiles a wave accumulates at once (forward / hidden backward)
constexpr int kNTB = 4;           // in-unit tiles a wave holds across read-out chunks (n_L <= 512)
constexpr int kChunkTiles = 16;   // read-out units processed per chunk = 256
constexpr int kLdPad = 4;         // LDS row padding (floats)
constexpr int kEnergyCols = kMaxLatent + 2;   // loss, E_1..E_L(max), overall

struct KLayer {
    float* x;              // state [Bpad][npad]
    float* m;              // Adam exp_avg    [Bpad][npad]
    float* v;              // Adam exp_avg_sq [Bpad][npad]
    float* xgrad;          // gradients-only mode: [B][n] (unpadded, caller's tensor)
    float* rec;            // trajectory records [rec_count][B][n] (unpadded) or null
    float* spill_e;        // l>=1: [slots][Bpad][npad] errors;  l==0: running sum of e_1 [Bpad][npad]
    float* spill_a;        // [slots][Bpad][npad] activations f(x_l)
    const float* ext_noise;// [n_steps][B][n] or null
    const f32x4* Wf;       // l>=1: packed forward weights of Linear l  [ntiles][nkb_in][64]
    const f32x4* Wb;       // l>=1: packed backward weights of Linear l [ntiles(l-1)][ntiles][64]
    const float* bias;     // l>=1: padded bias [npad]
    int n, npad, ntiles;   // units, padded to 16, npad/16
    int act;
    float ecoef;
    int lds_a, lds_e;      // float offsets of FX_l / E_l in LDS
    int ld;                // LDS row stride (floats) = npad + kLdPad
};

struct KHead {
    const f32x4* Wf;       // [ntiles][nkb_in][64]
    const f32x4* Wb;       // [ntiles(L-1)][ntiles][64]
    const float* bias;     // [npad]
    const float* y;        // padded target [Bpad][npad]
    float* rec_out;        // [rec_count][B][n] or null
    float* spill_e;        // [slots][Bpad][npad]
    int n, npad, ntiles;
    int loss_kind;
    float inv_var;
    int mask_start;
    int lds_eo;            // float offset of the read-out error chunk
    int ld;                // its row stride = kChunkTiles*16 + kLdPad
};

struct KParams {
    KLayer layer[kMaxLatent];
    KHead head;
    const float* mu1;      // prediction of the top latent layer [Bpad][npad_0] (inputs W0^T + b0)
    const float* adam_coef;// [n_steps][2]: step_size = lr/(1-b1^t), 1/sqrt(1-b2^t)
    double* epart;         // [energy rows][nWG][kEnergyCols] per-workgroup partial sums
    int L, has_head;
    int B, Bpad;
    int T, t0, n_steps;
    int xopt, update_x;
    float lr, beta2, omb1, omb2, eps;
    int noise_mode;
    float noise_scale;     // sqrt(noise_var*lr)
    uint64_t seed, step_base, chain_base;
    int acc_begin, acc_end, spill_t0;   // spill slot of step t = t - spill_t0 when acc_begin <= t < acc_end
    int energy_mode;
    int rec_begin, rec_stride, rec_count;
    int lds_red;           // float offset of the energy reduction scratch [2][kMaxLatent+1][kWaves]
};

// ------------------------------------------------------------------------------------------------
// Tile GEMM: acc[t][ct] (16 units x 16 chains, C layout: lane (c=lane&15, q=lane>>4) holds units
// 4q..4q+3 of chain c) += A_t * B, K = 16*nkb.
//   A: packed global fragments, one coalesced 1 KiB load per (tile, k-block): lane (m,q) holds
//      W[u0+m][k0+4q+r] (forward) or W[k0+4q+r][i0+m] (backward), r = 0..3
//   B: LDS rows [chain][k]: lane (c,q) reads 16 B at [c][k0+4q] -> the same k set per MFMA.
// MFMA r pairs A.r with B.r: k-set {k0+4q+r : q=0..3}; four MFMAs cover the 16-deep block.
// The fragment loads of k-block kb+1 are issued before the MFMAs of k-block kb (two named register
// sets, loop unrolled by two) so that an L2 round trip hides behind 8*NT MFMAs.
template <int NT, int NTT>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[NTT][2], const f32x4 (&a)[NT], f32x4 b0, f32x4 b1) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t][0] = mfma16(a[t].x, b0.x, acc[t][0]);
        acc[t][1] = mfma16(a[t].x, b1.x, acc[t][1]);
        acc[t][0] = mfma16(a[t].y, b0.y, acc[t][0]);
        acc[t][1] = mfma16(a[t].y, b1.y, acc[t][1]);
        acc[t][0] = mfma16(a[t].z, b0.z, acc[t][0]);
        acc[t][1] = mfma16(a[t].z, b1.z, acc[t][1]);
        acc[t][0] = mfma16(a[t].w, b0.w, acc[t][0]);
        acc[t][1] = mfma16(a[t].w, b1.w, acc[t][1]);
    }
}

template <int NT, int NTT>
__device__ __forceinline__ void gemm_fixed(f32x4 (&acc)[NTT][2], const f32x4* __restrict__ A,
                                           const int (&aoff)[NTT], int nkb,
                                           const float* B, int ldb, int lane) {
    const int c = lane & 15, q = lane >> 4;
    const float* b0p = B + c * ldb + 4 * q;
    const float* b1p = b0p + 16 * ldb;
    f32x4 aP[NT], aQ[NT], bP0, bP1, bQ0, bQ1;
#pragma unroll
    for (int t = 0; t < NT; ++t) aP[t] = A[aoff[t] + lane];
    bP0 = *reinterpret_cast<const f32x4*>(b0p);
    bP1 = *reinterpret_cast<const f32x4*>(b1p);
    int kb = 0;
    for (; kb + 1 < nkb; kb += 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t) aQ[t] = A[aoff[t] + (kb + 1) * 64 + lane];
        bQ0 = *reinterpret_cast<const f32x4*>(b0p + (kb + 1) * 16);
        bQ1 = *reinterpret_cast<const f32x4*>(b1p + (kb + 1) * 16);
        mfma_block<NT, NTT>(acc, aP, bP0, bP1);
        if (kb + 2 < nkb) {
#pragma unroll
            for (int t = 0; t < NT; ++t) aP[t] = A[aoff[t] + (kb + 2) * 64 + lane];
            bP0 = *reinterpret_cast<const f32x4*>(b0p + (kb + 2) * 16);
            bP1 = *reinterpret_cast<const f32x4*>(b1p + (kb + 2) * 16);
        }
        mfma_block<NT, NTT>(acc, aQ, bQ0, bQ1);
    }
    if (kb < nkb) mfma_block<NT, NTT>(acc, aP, bP0, bP1);
}

// nt (wave-uniform, 1..NTT) selects a straight-line instantiation: no per-tile branches in the loop
template <int NTT>
__device__ __forceinline__ void gemm_tiles(f32x4 (&acc)[NTT][2], const f32x4* __restrict__ A,
                                           const int (&aoff)[NTT], int nt, int nkb,
                                           const float* B, int ldb, int lane) {
    if (nkb <= 0) return;
    switch (nt) {
        case 1: gemm_fixed<1, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
        case 2: gemm_fixed<2, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
        case 3: gemm_fixed<3, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
        case 4: gemm_fixed<4, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
        default:
            if constexpr (NTT > 4) {
                switch (nt) {
                    case 5: gemm_fixed<5, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
                    case 6: gemm_fixed<6, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
                    case 7: gemm_fixed<7, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
                    case 8: gemm_fixed<8, NTT>(acc, A, aoff, nkb, B, ldb, lane); break;
                    default: break;
                }
            }
            break;
    }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 splat(float v) { f32x4 r = {v, v, v, v}; return r; }

// guarded scalar store of a C-layout quad into an unpadded [B][n] tensor row
__device__ __forceinline__ void st_unpadded(float* base, int chain, int n, int u0, f32x4 v) {
    float* p = base + (size_t)chain * n + u0;
    if (u0 + 0 < n) p[0] = v.x;
    if (u0 + 1 < n) p[1] = v.y;
    if (u0 + 2 < n) p[2] = v.z;
    if (u0 + 3 < n) p[3] = v.w;
}
__device__ __forceinline__ f32x4 ld_unpadded(const float* base, int chain, int n, int u0) {
    const float* p = base + (size_t)chain * n + u0;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (u0 + 0 < n) v.x = p[0];
    if (u0 + 1 < n) v.y = p[1];
    if (u0 + 2 < n) v.z = p[2];
    if (u0 + 3 < n) v.w = p[3];
    return v;
}

// Everything the owner lane of a (layer, unit tile, chain tile) quad does when x_l(t) is first
// touched in a step: error, energy, activation to LDS, spills, trajectory record.  `x` was fetched
// by the caller (ahead of the GEMM that produced `mu`).
// Returns the energy contribution 0.5*c*sum(d^2) of the quad (0 for padded chains).
__device__ __forceinline__ float owner_forward(const KParams& P, const KLayer& Ly, int l, float* lds,
                                               int chain_local, int chain, int u0, f32x4 x, f32x4 mu,
                                               int slot, int rec_idx) {
    const size_t row = (size_t)chain * Ly.npad + u0;
    const f32x4 d = x - mu;
    const f32x4 e = d * Ly.ecoef;
    f32x4 fx;
    fx.x = act_f(Ly.act, x.x); fx.y = act_f(Ly.act, x.y); fx.z = act_f(Ly.act, x.z); fx.w = act_f(Ly.act, x.w);
    const bool live = chain < P.B;
    st4(lds + Ly.lds_a + chain_local * Ly.ld + u0, fx);
    if (l > 0) st4(lds + Ly.lds_e + chain_local * Ly.ld + u0, e);
    if (slot >= 0) {
        const f32x4 z = splat(0.f);
        const size_t srow = ((size_t)slot * P.Bpad + chain) * Ly.npad + u0;
        st4(Ly.spill_a + srow, live ? fx : z);
        if (l > 0) {
            st4(Ly.spill_e + srow, live ? e : z);
        } else if (live) {        // Linear 0: only sum_t e_1 is needed (its input is constant)
            float* s = Ly.spill_e + row;
            st4(s, ld4(s) + e);
        }
    }
    if (rec_idx >= 0 && Ly.rec != nullptr && live)
        st_unpadded(Ly.rec + (size_t)rec_idx * P.B * Ly.n, chain, Ly.n, u0, x);
    const f32x4 dd = d * d;
    return live ? 0.5f * Ly.ecoef * (dd.x + dd.y + dd.z + dd.w) : 0.0f;
}

// x_l update of one quad given x (prefetched), its error e_l, the back-projected error (C layout)
// and its sign:   g = e_l + sign * f'(x_l) * back
// (sign -1: next layer is a PCLayer, +1: read-out loss, 0: nothing below)
__device__ __forceinline__ void owner_update(const KParams& P, const KLayer& Ly, int l,
                                             int chain, int u0, f32x4 x, f32x4 e, f32x4 back, float sign,
                                             int s, int t) {
    const size_t row = (size_t)chain * Ly.npad + u0;
    f32x4 g;
    {
        const float fx0 = act_f(Ly.act, x.x), fx1 = act_f(Ly.act, x.y), fx2 = act_f(Ly.act, x.z), fx3 = act_f(Ly.act, x.w);
        g.x = e.x + sign * act_d(Ly.act, x.x, fx0) * back.x;
        g.y = e.y + sign * act_d(Ly.act, x.y, fx1) * back.y;
        g.z = e.z + sign * act_d(Ly.act, x.z, fx2) * back.z;
        g.w = e.w + sign * act_d(Ly.act, x.w, fx3) * back.w;
    }
    const bool live = chain < P.B;
    if (!P.update_x) {
        if (live && Ly.xgrad != nullptr) st_unpadded(Ly.xgrad, chain, Ly.n, u0, g);
        return;
    }
    f32x4 xn;
    if (P.xopt == MCPC_XOPT_SGD) {
        xn = x - g * P.lr;
    } else {
        // torch.optim.Adam single-tensor path: lerp_, mul_/addcmul_, sqrt/bias2 + eps, addcdiv_
        f32x4 m = ld4(Ly.m + row), v = ld4(Ly.v + row);
        m = m + (g - m) * P.omb1;
        v = v * P.beta2 + (g * g) * P.omb2;
        st4(Ly.m + row, m);
        st4(Ly.v + row, v);
        const float step_size = P.adam_coef[2 * s], inv_bc2 = P.adam_coef[2 * s + 1];
        f32x4 den;
        den.x = __builtin_sqrtf(v.x) * inv_bc2 + P.eps; den.y = __builtin_sqrtf(v.y) * inv_bc2 + P.eps;
        den.z = __builtin_sqrtf(v.z) * inv_bc2 + P.eps; den.w = __builtin_sqrtf(v.w) * inv_bc2 + P.eps;
        xn.x = x.x - step_size * (m.x / den.x); xn.y = x.y - step_size * (m.y / den.y);
        xn.z = x.z - step_size * (m.z / den.z); xn.w = x.w - step_size * (m.w / den.w);
    }
    if (P.noise_mode == MCPC_NOISE_PHILOX) {
        const f32x4 z = normals4(P.seed, P.step_base + (uint64_t)t, (uint32_t)l,
                                 (uint32_t)(P.chain_base + (uint64_t)chain), (uint32_t)(u0 >> 2));
        xn = xn + z * P.noise_scale;
    } else if (P.noise_mode == MCPC_NOISE_EXTERNAL) {
        if (live) {
            const f32x4 z = ld_unpadded(Ly.ext_noise + (size_t)s * P.B * Ly.n, chain, Ly.n, u0);
            xn = xn + z * P.noise_scale;
        }
    }
    // padded units stay exactly zero (their gradient is zero; only the noise must be masked)
    if (u0 + 0 >= Ly.n) xn.x = 0.f;
    if (u0 + 1 >= Ly.n) xn.y = 0.f;
    if (u0 + 2 >= Ly.n) xn.z = 0.f;
    if (u0 + 3 >= Ly.n) xn.w = 0.f;
    st4(Ly.x + row, xn);
}

__global__ __launch_bounds__(kThreads) void mcpc_steps_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int chain0 = blockIdx.x * kCT;
    const int L = P.L;

    for (int s = 0; s < P.n_steps; ++s) {
        const int t = P.t0 + s;
        const bool do_energy = (P.energy_mode == MCPC_ENERGY_ALL) ||
                               (P.energy_mode == MCPC_ENERGY_LAST && t == P.T - 1);
        const int slot = (t >= P.acc_begin && t < P.acc_end) ? (t - P.spill_t0) : -1;
        int rec_idx = -1;
        if (P.rec_count > 0 && t >= P.rec_begin) {
            const int k = (t - P.rec_begin) / P.rec_stride;
            if (k < P.rec_count && P.rec_begin + k * P.rec_stride == t) rec_idx = k;
        }
        float* red = lds + P.lds_red + (s & 1) * (kMaxLatent + 1) * kWaves;

        // ---- top latent layer: prediction is the constant mu1 ---------------------------------
        {
            const KLayer& Ly = P.layer[0];
            float esum = 0.f;
            for (int ut = wave; ut < Ly.ntiles; ut += kWaves) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int cl = 16 * ct + c, chain = chain0 + cl, u0 = 16 * ut + 4 * q;
                    const size_t row = (size_t)chain * Ly.npad + u0;
                    const f32x4 x = ld4(Ly.x + row);
                    const f32x4 mu = ld4(P.mu1 + row);
                    esum += owner_forward(P, Ly, 0, lds, cl, chain, u0, x, mu, slot, rec_idx);
                }
            }
            if (do_energy) { esum = wave_sum(esum); if (lane == 0) red[0 * kWaves + wave] = esum; }
        }
        __syncthreads();

        // ---- hidden predictions mu_l = f(x_{l-1}) W_l^T + b_l, errors e_l ----------------------
        for (int l = 1; l < L; ++l) {
            const KLayer& Ly = P.layer[l];
            const KLayer& Lp = P.layer[l - 1];
            const int nkb = Lp.ntiles;
            float esum = 0.f;
            for (int base = 0; base < Ly.ntiles; base += kNT * kWaves) {
                f32x4 acc[kNT][2], xq[kNT][2], bias[kNT];
                int aoff[kNT];
                int nt = 0;
#pragma unroll
                for (int i = 0; i < kNT; ++i) {
                    acc[i][0] = splat(0.f); acc[i][1] = splat(0.f);
                    const int ut = base + wave + kWaves * i;
                    aoff[i] = ut * nkb * 64;
                    if (ut < Ly.ntiles) {
                        nt = i + 1;
                        // operands of the epilogue are fetched ahead of the GEMM
                        const int u0 = 16 * ut + 4 * q;
                        bias[i] = ld4(Ly.bias + u0);
                        xq[i][0] = ld4(Ly.x + (size_t)(chain0 + c) * Ly.npad + u0);
                        xq[i][1] = ld4(Ly.x + (size_t)(chain0 + 16 + c) * Ly.npad + u0);
                    }
                }
                gemm_tiles<kNT>(acc, Ly.Wf, aoff, nt, nkb, lds + Lp.lds_a, Lp.ld, lane);
#pragma unroll
                for (int i = 0; i < kNT; ++i) {
                    if (i < nt) {
                        const int ut = base + wave + kWaves * i, u0 = 16 * ut + 4 * q;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const int cl = 16 * ct + c, chain = chain0 + cl;
                            esum += owner_forward(P, Ly, l, lds, cl, chain, u0, xq[i][ct], acc[i][ct] + bias[i], slot, rec_idx);
                        }
                    }
                }
            }
            if (do_energy) { esum = wave_sum(esum); if (lane == 0) red[l * kWaves + wave] = esum; }
            __syncthreads();
        }

        // ---- read-out: out = f(x_L) W^T + b, loss error e_o, back-projection into x_L ----------
        const KLayer& Ll = P.layer[L - 1];
        f32x4 accb[kNTB][2];
#pragma unroll
        for (int i = 0; i < kNTB; ++i) { accb[i][0] = splat(0.f); accb[i][1] = splat(0.f); }
        if (P.has_head) {
            const KHead& H = P.head;
            const int nkbF = Ll.ntiles;
            float lsum = 0.f;
            int boff[kNTB];
            int ntb = 0;
#pragma unroll
            for (int i = 0; i < kNTB; ++i) {
                const int it = wave + kWaves * i;
                boff[i] = it * H.ntiles * 64;
                if (it < Ll.ntiles) ntb = i + 1;
            }
            for (int tile0 = 0; tile0 < H.ntiles; tile0 += kChunkTiles) {
                const int ntc = min(kChunkTiles, H.ntiles - tile0);
                f32x4 acc[kNT][2], yq[kNT][2], bias[kNT];
                int aoff[kNT];
                int nt = 0;
#pragma unroll
                for (int i = 0; i < kNT; ++i) {
                    acc[i][0] = splat(0.f); acc[i][1] = splat(0.f);
                    const int ut = tile0 + wave + kWaves * i;
                    aoff[i] = ut * nkbF * 64;
                    if (wave + kWaves * i < ntc) {
                        nt = i + 1;
                        const int u0 = 16 * ut + 4 * q;
                        bias[i] = ld4(H.bias + u0);
                        if (H.loss_kind != MCPC_LOSS_NONE) {
                            yq[i][0] = ld4(H.y + (size_t)(chain0 + c) * H.npad + u0);
                            yq[i][1] = ld4(H.y + (size_t)(chain0 + 16 + c) * H.npad + u0);
                        }
                    }
                }
                gemm_tiles<kNT>(acc, H.Wf, aoff, nt, nkbF, lds + Ll.lds_a, Ll.ld, lane);
#pragma unroll
                for (int i = 0; i < kNT; ++i) {
                    if (i < nt) {
                        const int ut = tile0 + wave + kWaves * i, u0 = 16 * ut + 4 * q;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const int cl = 16 * ct + c, chain = chain0 + cl;
                            const bool live = chain < P.B;
                            const f32x4 o = acc[i][ct] + bias[i];
                            f32x4 e = splat(0.f);
                            if (H.loss_kind != MCPC_LOSS_NONE) {
                                const f32x4 y = yq[i][ct];
                                float ov[4] = {o.x, o.y, o.z, o.w}, yv[4] = {y.x, y.y, y.z, y.w}, ev[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int u = u0 + r;
                                    const bool on = live && u >= H.mask_start && u < H.n;
                                    float er = 0.f, lv = 0.f;
                                    if (H.loss_kind == MCPC_LOSS_GAUSSIAN) {
                                        const float dlt = ov[r] - yv[r];
                                        er = H.inv_var * dlt;
                                        lv = 0.5f * H.inv_var * dlt * dlt;
                                    } else {
                                        er = sigmoid_f(ov[r]) - yv[r];
                                        if (do_energy) lv = bce_logits_f(ov[r], yv[r]);
                                    }
                                    ev[r] = on ? er : 0.f;
                                    lsum += on ? lv : 0.f;
                                }
                                e.x = ev[0]; e.y = ev[1]; e.z = ev[2]; e.w = ev[3];
                            }
                            st4(lds + H.lds_eo + cl * H.ld + (u0 - 16 * tile0), e);
                            if (slot >= 0)
                                st4(H.spill_e + ((size_t)slot * P.Bpad + chain) * H.npad + u0, e);
                            if (rec_idx >= 0 && H.rec_out != nullptr && live)
                                st_unpadded(H.rec_out + (size_t)rec_idx * P.B * H.n, chain, H.n, u0, o);
                        }
                    }
                }
                __syncthreads();
                // back-projection of this chunk: accb[it] += W[chunk units][it]^T e_o[chunk]
                {
                    int boffc[kNTB];
#pragma unroll
                    for (int i = 0; i < kNTB; ++i) boffc[i] = boff[i] + tile0 * 64;
                    gemm_tiles<kNTB>(accb, H.Wb, boffc, ntb, ntc, lds + H.lds_eo, H.ld, lane);
                }
                __syncthreads();
            }
            if (do_energy) { lsum = wave_sum(lsum); if (lane == 0) red[kMaxLatent * kWaves + wave] = lsum; }
        }

        // ---- energies of this step -> per-workgroup partials (deterministic final reduce on host side kernel)
        if (do_energy) {
            __syncthreads();   // uniform branch: publishes every wave's red[] entries of this step
            if (tid <= kMaxLatent) {
                double v = 0.0;
                const bool used = (tid < L) || (tid == kMaxLatent && P.has_head);
                if (used) {
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) v += (double)red[tid * kWaves + w];
                }
                const int erow = (P.energy_mode == MCPC_ENERGY_ALL) ? t : 0;
                P.epart[((size_t)erow * gridDim.x + blockIdx.x) * (kMaxLatent + 1) + tid] = v;
            }
        }

        // ---- x updates, bottom-up: x_L first (needs the read-out back-projection) ---------------
#pragma unroll
        for (int i = 0; i < kNTB; ++i) {
            const int it = wave + kWaves * i;
            if (it < Ll.ntiles) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int cl = 16 * ct + c, chain = chain0 + cl, u0 = 16 * it + 4 * q;
                    const size_t row = (size_t)chain * Ll.npad + u0;
                    const f32x4 x = ld4(Ll.x + row);
                    const f32x4 e = (L == 1) ? (x - ld4(P.mu1 + row)) * Ll.ecoef
                                             : ld4(lds + Ll.lds_e + cl * Ll.ld + u0);
                    owner_update(P, Ll, L - 1, chain, u0, x, e, accb[i][ct], P.has_head ? 1.0f : 0.0f, s, t);
                }
            }
        }
        for (int l = L - 1; l >= 1; --l) {
            // back-projection through Linear l into layer l-1: back = e_l W_l
            const KLayer& Ly = P.layer[l];
            const KLayer& Lp = P.layer[l - 1];
            const int nkb = Ly.ntiles;
            for (int base = 0; base < Lp.ntiles; base += kNT * kWaves) {
                f32x4 acc[kNT][2], xq[kNT][2], eq[kNT][2];
                int aoff[kNT];
                int nt = 0;
#pragma unroll
                for (int i = 0; i < kNT; ++i) {
                    acc[i][0] = splat(0.f); acc[i][1] = splat(0.f);
                    const int it = base + wave + kWaves * i;
                    aoff[i] = it * nkb * 64;
                    if (it < Lp.ntiles) {
                        nt = i + 1;
                        const int u0 = 16 * it + 4 * q;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const int cl = 16 * ct + c;
                            const size_t row = (size_t)(chain0 + cl) * Lp.npad + u0;
                            xq[i][ct] = ld4(Lp.x + row);
                            eq[i][ct] = (l == 1) ? (xq[i][ct] - ld4(P.mu1 + row)) * Lp.ecoef
                                                 : ld4(lds + Lp.lds_e + cl * Lp.ld + u0);
                        }
                    }
                }
                gemm_tiles<kNT>(acc, Ly.Wb, aoff, nt, nkb, lds + Ly.lds_e, Ly.ld, lane);
#pragma unroll
                for (int i = 0; i < kNT; ++i) {
                    if (i < nt) {
                        const int it = base + wave + kWaves * i;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const int cl = 16 * ct + c;
                            owner_update(P, Lp, l - 1, chain0 + cl, 16 * it + 4 * q, xq[i][ct], eq[i][ct], acc[i][ct], -1.0f, s, t);
                        }
                    }
                }
            }
        }
        // no barrier here: the next step's first barrier (after the top-layer pass, which only
        // writes FX_0 and red[(s+1)&1]) orders this step's E_l / e_o readers before their next writers.
    }
}

// ------------------------------------------------------------------------------------------------
// Weight packing into MFMA fragment order (run once per parameter change).
//   forward : Wf[ut][kb][lane][r] = W[16ut + (lane&15)][16kb + 4(lane>>4) + r]
//   backward: Wb[it][ub][lane][r] = W[16ub + 4(lane>>4) + r][16it + (lane&15)]
__global__ void mcpc_pack_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                 float* __restrict__ Wf, float* __restrict__ Wb, float* __restrict__ bias_pad,
                                 int n_out, int n_in, int out_tiles, int in_tiles) {
    const size_t total = (size_t)out_tiles * in_tiles * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int r = idx & 3, lane = (idx >> 2) & 63;
        const size_t blk = idx >> 8;
        {   // forward: blk = ut*in_tiles + kb
            const int ut = blk / in_tiles, kb = blk % in_tiles;
            const int u = 16 * ut + (lane & 15), k = 16 * kb + 4 * (lane >> 4) + r;
            Wf[idx] = (u < n_out && k < n_in) ? W[(size_t)u * n_in + k] : 0.f;
        }
        {   // backward: blk = it*out_tiles + ub
            const int it = blk / out_tiles, ub = blk % out_tiles;
            const int u = 16 * ub + 4 * (lane >> 4) + r, i = 16 * it + (lane & 15);
            Wb[idx] = (u < n_out && i < n_in) ? W[(size_t)u * n_in + i] : 0.f;
        }
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < out_tiles * 16) bias_pad[gid] = (bias != nullptr && gid < n_out) ? bias[gid] : 0.f;
}

// dst[Bpad][npad] <- src[B][n] (zero padded)      /     dst[B][n] <- src[Bpad][npad]
__global__ void mcpc_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int n, int Bpad, int npad) {
    const size_t total = (size_t)Bpad * npad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = idx / npad, col = idx % npad;
        dst[idx] = (src != nullptr && row < B && col < n) ? src[(size_t)row * n + col] : 0.f;
    }
}
__global__ void mcpc_unpad_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int n, int npad) {
    const size_t total = (size_t)B * n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = idx / n, col = idx % n;
        dst[idx] = src[(size_t)row * npad + col];
    }
}

// mu1[Bpad][npad] = inputs W0^T + b0 (inputs may be null = zeros).  Tiny (n_in*n_1 per chain).
__global__ void mcpc_mu1_kernel(const float* __restrict__ inputs, const float* __restrict__ W0,
                                const float* __restrict__ b0, float* __restrict__ mu1,
                                int B, int n_in, int n1, int Bpad, int npad) {
    const size_t total = (size_t)Bpad * npad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = idx / npad, u = idx % npad;
        float acc = 0.f;
        if (u < n1) {
            if (b0 != nullptr) acc = b0[u];
            if (inputs != nullptr && row < B) {
                float dot = 0.f;
                for (int k = 0; k < n_in; ++k) dot = fmaf(inputs[(size_t)row * n_in + k], W0[(size_t)u * n_in + k], dot);
                acc += dot;
            }
        }
        mu1[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// K3: split-K Hebbian GEMM.  slab[ks][u][i] = sum_{r in split ks} E[r][u] * A[r][i].
// Operands are row-major [rows][npad]; each lane loads 16 B = 4 consecutive columns of one row and
// feeds component j to MFMA tile j whose rows/cols are the columns {base + 4*lane16 + j}: the
// column permutation lives only in the epilogue index, loads stay 256 B-contiguous per row.
// Wave tile = 64 x 64 outputs (4x4 MFMA tiles), one wave tile per wave, 4 waves per workgroup.
__global__ __launch_bounds__(256) void mcpc_dw_kernel(const float* __restrict__ E, const float* __restrict__ A,
                                                      float* __restrict__ slab, int rows, int ne, int na,
                                                      int rows_per_split) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_a = (na + 63) / 64, tiles_e = (ne + 63) / 64;
    const int wt = blockIdx.x * 4 + wave;
    if (wt >= tiles_a * tiles_e) return;
    const int te = wt / tiles_a, ta = wt % tiles_a;
    const int m = lane & 15, q = lane >> 4;
    const int ue = 64 * te + 4 * m, ua = 64 * ta + 4 * m;      // this lane's 4 columns of E / A
    const bool ve = ue < ne, va = ua < na;
    const int r0 = blockIdx.y * rows_per_split;
    const int r1 = min(rows, r0 + rows_per_split);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = splat(0.f);
    const f32x4 z = splat(0.f);
#pragma unroll 4
    for (int r = r0; r < r1; r += 4) {
        const f32x4 e = ve ? ld4(E + (size_t)(r + q) * ne + ue) : z;
        const f32x4 a = va ? ld4(A + (size_t)(r + q) * na + ua) : z;
        const float ev[4] = {e.x, e.y, e.z, e.w}, av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(ev[i], av[j], acc[i][j]);
    }
    // C layout of tile (i,j): row 4q+reg -> E column 64te + 4(4q+reg) + i ; col m -> A column 64ta + 4m + j
    float* out = slab + (size_t)blockIdx.y * ne * na;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int u = 64 * te + 4 * (4 * q + reg) + i;
            if (u < ne && va) {
                f32x4 v;
                v.x = acc[i][0][reg]; v.y = acc[i][1][reg]; v.z = acc[i][2][reg]; v.w = acc[i][3][reg];
                st4(out + (size_t)u * na + ua, v);
            }
        }
    }
}

// column sums for the bias gradients: slab[ks][u] = sum_{r in split} E[r][u]
__global__ __launch_bounds__(256) void mcpc_colsum_kernel(const float* __restrict__ E, float* __restrict__ slab,
                                                          int rows, int ne, int rows_per_split) {
    __shared__ float part[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    float s = 0.f;
    if (col < ne)
        for (int r = r0 + sub; r < r1; r += 4) s += E[(size_t)r * ne + col];
    part[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && col < ne)
        slab[(size_t)blockIdx.y * ne + col] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// dst[i] (=|+=) sign * sum_k slab[k][i], fixed order -> bitwise reproducible
__global__ void mcpc_reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ dst, size_t n, int ksplit,
                                         float sign, int accumulate) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < ksplit; ++k) s += slab[(size_t)k * n + idx];
        dst[idx] = accumulate ? dst[idx] + sign * s : sign * s;
    }
}

// Linear 0 (constant input): G_W0[u][k] = -sum_chain esum[chain][u] * inputs[chain][k];  G_b0[u] = -sum_chain esum[chain][u]
__global__ void mcpc_dw0_kernel(const float* __restrict__ esum, const float* __restrict__ inputs,
                                float* __restrict__ gW, float* __restrict__ gb, int B, int n1, int npad1, int n_in,
                                int gw_ld, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n1 * n_in) {
        const int u = idx / n_in, k = idx % n_in;
        float s = 0.f;
        if (inputs != nullptr)
            for (int ch = 0; ch < B; ++ch) s = fmaf(esum[(size_t)ch * npad1 + u], inputs[(size_t)ch * n_in + k], s);
        float* d = gW + (size_t)u * gw_ld + k;
        *d = accumulate ? *d - s : -s;
    }
    if (idx < n1) {
        float s = 0.f;
        for (int ch = 0; ch < B; ++ch) s += esum[(size_t)ch * npad1 + idx];
        gb[idx] = accumulate ? gb[idx] - s : -s;
    }
}

// energies_out[row][:] = {loss, E_1..E_L, 0.., overall} from per-workgroup partials (fixed order, fp64)
__global__ void mcpc_energy_reduce_kernel(const double* __restrict__ epart, double* __restrict__ out, int rows, int nwg, int L) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double loss = 0.0, overall = 0.0;
    double* o = out + (size_t)row * kEnergyCols;
    for (int l = 0; l <= kMaxLatent; ++l) {
        double s = 0.0;
        for (int w = 0; w < nwg; ++w) s += epart[((size_t)row * nwg + w) * (kMaxLatent + 1) + l];
        if (l < kMaxLatent) { o[1 + l] = s; overall += s; } else { loss = s; }
    }
    o[0] = loss;
    o[kEnergyCols - 1] = overall + loss;
}

// out[scale * G] with un-padding: dst[u][i] (=|+=) scale * G[u][i_pad]
__global__ void mcpc_export_grad_kernel(const float* __restrict__ G, float* __restrict__ dst, int n_out, int n_in, int ld,
                                        float scale, int accumulate) {
    const size_t total = (size_t)n_out * n_in;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int u = idx / n_in, i = idx % n_in;
        const float v = scale * G[(size_t)u * ld + i];
        dst[idx] = accumulate ? dst[idx] + v : v;
    }
}

__global__ void mcpc_philox_kernel(uint64_t seed, uint64_t step, int layer, uint64_t chain_base, int batch, int n_units,
                                   float* __restrict__ out, int raw) {
    const int groups = (n_units + 3) / 4;
    const size_t total = (size_t)batch * groups;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int chain = idx / groups, g = idx % groups;
        float v[4];
        if (raw) {
            uint32_t r[4];
            philox4x32_10((uint32_t)(chain_base + chain), ((uint32_t)layer << 24) | (uint32_t)g, (uint32_t)step,
                          (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
            for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(r[k]);
        } else {
            const f32x4 z = normals4(seed, step, (uint32_t)layer, (uint32_t)(chain_base + chain), (uint32_t)g);
            v[0] = z.x; v[1] = z.y; v[2] = z.z; v[3] = z.w;
        }
        for (int k = 0; k < 4; ++k)
            if (4 * g + k < n_units) out[(size_t)chain * n_units + 4 * g + k] = v[k];
    }
}

}  // namespace mcpc
