// K3 -- Hebbian sums  G[u][i] = sum_r E[r][u] * A[r][i]  over the spilled (error, activation) rows, for the wide Linears.
// Replaces the parameter part of overall.backward() (reference predictive_coding/pc_trainer.py:862).
//
// A plain fp32 GEMM with a tiny output (ne x na, at most 784 x 256 at cfg-M) and a huge K (rows = steps x chains, 385 k per
// flush of 64 steps): split-K over the grid, every split writes its partial tile to a slab, the slabs are summed in a fixed
// order by mcpc_reduce_jobs_kernel (bitwise reproducible; no float atomics).
//
// Workgroup = 8 waves (two per SIMD), tile = TE x TA MFMA tiles of 16 x 16 (TE <= 17 error tiles x TA = 8 RA activation
// tiles: 272 x 256 outputs for the read-out Linear), so that a spilled byte is read once per workgroup column and every
// k-row of the tile costs (TE + TA) x 64 B for TE x TA x 256 MACs: 8.2 B per CU-cycle at the full fp32 MFMA rate, which
// HBM + L2 deliver (the first kernel read its operands straight into registers, 64 x 64 per wave: 32 B per CU-cycle).
//   * operands travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no staging registers),
//     32 rows per stage, two stages; the DMA of stage s+1 is issued before the MFMAs of stage s (17 k cycles per SIMD);
//   * LDS images are the row-major panels themselves ([32][16 TE] and [32][16 TA] floats, lane-linear as LDS-DMA requires);
//     MFMA operand (tile t, k-step ks) = one ds_read_b32 at [4 ks + lane/16][16 t + lane%16];
//   * wave w owns activation tiles RA w .. RA w + RA - 1 against all TE error tiles: TE x RA accumulators (136 VGPRs at
//     17 x 2), TE + RA LDS reads per TE x RA MFMAs;
//   * bias sums (column sums of E) are taken from the LDS panel by the VALU (one thread per column), not by extra MFMAs.
#pragma once

namespace mcpc {

constexpr int kHebKB = 32;           // spilled rows per stage
constexpr int kHebThreads = 512;

struct HebArgs {
    const float* E;                  // [rows][ne]   spilled errors (row stride ne: padded width)
    const float* A;                  // [rows][na]   spilled activations
    float* slab;                     // [ksplit][ne][na]
    float* slab_b;                   // [ksplit][ne]
    int rows, ne, na;                // rows % 32 == 0; ne, na multiples of 16
    int rows_per_split;              // multiple of 32
    int n_mt, n_nt, ksplit;          // error-tile groups, activation-tile groups, K splits: grid = n_mt * n_nt * ksplit workgroups
    int e_col_base;                  // first E column of this launch (a Linear may be covered by launches of different TE)
};

#ifndef MCPC_HEB_LD_AUX
#define MCPC_HEB_LD_AUX 0          // cache policy of the spill reads (0 plain, 1 sc0, 2 nt, 16 sc1, 18 nt sc1): the learning call runs
                                   // within 0.2 us per step of itself with any of them (scripts/lib_ab.sh), so the reads are not what
                                   // costs the step kernel beside it its L2 hits (0.82 against 0.95-0.99 without a flush)
#endif
#ifndef MCPC_HEB_LDS_PAD
#define MCPC_HEB_LDS_PAD 16        // floats added to a 256-float panel row in LDS (0: the linear image of round 2, for A/B runs)
#endif
constexpr int heb_lds_stride(int row_floats) { return row_floats == 256 ? row_floats + MCPC_HEB_LDS_PAD : row_floats; }

__device__ __forceinline__ void heb_glds16(const float* gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, MCPC_HEB_LD_AUX);
}

// SWAPPED: the launch computes the TRANSPOSED product for a Linear with a narrow input (the E slot holds its activations,
// the A slot its errors), so that the narrow operand takes the TE slot; the bias sums (column sums of the errors) then
// come from the A panel, and the reduction writes the slab back transposed.
template <int TE, int RA, bool SWAPPED = false>
__global__ __launch_bounds__(kHebThreads, 2) void mcpc_heb_kernel(const HebArgs P) {
    constexpr int TA = 8 * RA;
    constexpr int LDE = 16 * TE, LDA = 16 * TA;               // panel row lengths (floats)
    // LDS row strides.  An MFMA operand is one ds_read_b32 at [4 ks + lane / 16][16 t + lane % 16]: banks are (a / 4) mod 32 and
    // lanes 0-31 (k-rows kq = 0, 1) are served together, so a row stride that is a multiple of 32 floats puts both k-rows on the
    // same 16 banks (2-way conflict on every read: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.47 for <16, 2>, 0.08 for <17, 2>
    // whose 272-float E rows are clear of it by themselves).  A panel whose row is exactly one 1 KiB LDS-DMA piece (256 floats)
    // can be spread: piece r lands at r * (256 + 16) floats.  (Rows of 128 floats share a piece with their neighbour and stay.)
    constexpr int LDE_S = heb_lds_stride(LDE), LDA_S = heb_lds_stride(LDA);
    constexpr int E_PIECES = kHebKB * LDE / 256;               // 1 KiB pieces per stage
    constexpr int A_PIECES = kHebKB * LDA / 256;
    constexpr int PIECES = E_PIECES + A_PIECES;
    constexpr int PPW = (PIECES + 7) / 8;                      // pieces per wave
    constexpr int STAGE = kHebKB * (LDE_S + LDA_S);            // floats per stage
    static_assert(kHebKB * LDE % 256 == 0 && kHebKB * LDA % 256 == 0, "panels must be whole 1 KiB pieces");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;

    // workgroup -> (split, activation group, error group).  Consecutive LOGICAL ids share a K split (the same rows of A and,
    // across activation groups, of E), so they are put on one XCD (hardware workgroup id % 8) and meet in its L2.
    const int total = P.n_mt * P.n_nt * P.ksplit;
    int id = blockIdx.x;
    if (total % 8 == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int per_split = P.n_mt * P.n_nt;
    const int split = id / per_split, rem = id - split * per_split;
    const int nt = rem / P.n_mt, mt = rem - nt * P.n_mt;
    const int e_col0 = P.e_col_base + mt * LDE, a_col0 = nt * LDA;
    const int r0 = split * P.rows_per_split;
    const int r1 = min(P.rows, r0 + P.rows_per_split);
    const int n_stage = (r1 - r0) / kHebKB;

    // masked panel columns (tiles past ne / na) are never written by the DMA: they stay zero
    for (int i = tid; i < 2 * STAGE / 4; i += kHebThreads) reinterpret_cast<f32x4*>(lds)[i] = splat(0.f);
    __syncthreads();

    // this wave's pieces: (global offset relative to the stage's first row, LDS offset inside a stage); fixed for the kernel
    int g_off[PPW], l_off[PPW];
    bool p_on[PPW], p_is_a[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int p = w + 8 * k;
        const bool is_a = p >= E_PIECES;
        const int c = (is_a ? p - E_PIECES : p) * 64 + lane;            // 16-byte chunk of the panel
        const int cpr = (is_a ? LDA : LDE) / 4;                         // chunks per row
        const int row = c / cpr, col = 4 * (c - row * cpr);
        const int gcol = (is_a ? a_col0 : e_col0) + col;
        p_is_a[k] = is_a;
        p_on[k] = p < PIECES && gcol < (is_a ? P.na : P.ne);
        g_off[k] = row * (is_a ? P.na : P.ne) + gcol;
        // (one piece per row when the row stride is padded; otherwise the panel is one linear image)
        l_off[k] = is_a ? kHebKB * LDE_S + (LDA_S != LDA ? (p - E_PIECES) * LDA_S : (p - E_PIECES) * 256)
                        : (LDE_S != LDE ? p * LDE_S : p * 256);
    }
    auto issue_stage = [&](int s, int buf) {
        const size_t rbase = (size_t)(r0 + s * kHebKB);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const float* src = (p_is_a[k] ? P.A + rbase * P.na : P.E + rbase * P.ne) + g_off[k];
            if (p_on[k]) heb_glds16(src, lds + buf * STAGE + l_off[k]);
        }
    };

    f32x4 acc[TE][RA];
#pragma unroll
    for (int i = 0; i < TE; ++i)
#pragma unroll
        for (int j = 0; j < RA; ++j) acc[i][j] = splat(0.f);
    // column sum of the errors: column e_col0 + tid of the E panel (threads < LDE, nt == 0) or, SWAPPED, column
    // a_col0 + tid of the A panel (threads < LDA, mt == 0)
    float bsum = 0.f;
    const bool does_bias = SWAPPED ? (mt == 0 && tid < LDA) : (nt == 0 && tid < LDE);

    if (n_stage > 0) issue_stage(0, 0);
    __syncthreads();                                          // (drains the DMA: hipcc waits vmcnt(0) in front of the barrier)
    const int e_rd = kq * LDE_S + m;                          // lane's operand offset inside a k-step's four rows
    const int a_rd = kHebKB * LDE_S + kq * LDA_S + 16 * RA * w + m;
    for (int s = 0; s < n_stage; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_stage) issue_stage(s + 1, buf ^ 1);
        const float* Ep = lds + buf * STAGE + e_rd;
        const float* Ap = lds + buf * STAGE + a_rd;
        float eA[TE], aA[RA], eB[TE], aB[RA];
#define HEB_LOAD(e_, a_, ks_)                                                                   \
        do {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < TE; ++i) e_[i] = Ep[(ks_) * 4 * LDE_S + 16 * i]; \
            _Pragma("unroll") for (int j = 0; j < RA; ++j) a_[j] = Ap[(ks_) * 4 * LDA_S + 16 * j]; \
        } while (0)
#define HEB_MFMA(e_, a_)                                                                        \
        do {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < TE; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < RA; ++j) acc[i][j] = mfma16(e_[i], a_[j], acc[i][j]); \
        } while (0)
        // operands of k-step ks+1 are requested before the MFMAs of k-step ks; the sched_barriers keep hipcc from sinking
        // every read next to its first use (it then waited out an LDS round trip in front of every fourth MFMA)
        HEB_LOAD(eA, aA, 0);
#pragma unroll
        for (int ks = 0; ks < kHebKB / 4; ks += 2) {
            __builtin_amdgcn_sched_barrier(0);
            HEB_LOAD(eB, aB, ks + 1);
            HEB_MFMA(eA, aA);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < kHebKB / 4) HEB_LOAD(eA, aA, ks + 2);
            HEB_MFMA(eB, aB);
        }
        __builtin_amdgcn_sched_barrier(0);
#undef HEB_LOAD
#undef HEB_MFMA
        if (does_bias) {
            const float* col = lds + buf * STAGE + (SWAPPED ? kHebKB * LDE_S : 0) + tid;
#pragma unroll 8
            for (int r = 0; r < kHebKB; ++r) bsum += col[r * (SWAPPED ? LDA_S : LDE_S)];
        }
        __syncthreads();      // stage s+1 has landed (vmcnt(0) in front of the barrier); every wave is done with stage s
    }

    // C layout of tile (i, j): row 4 kq + reg -> error unit, column m -> activation unit
    float* out = P.slab + (size_t)split * P.ne * P.na;
#pragma unroll
    for (int i = 0; i < TE; ++i) {
        const int u0 = e_col0 + 16 * i + 4 * kq;
        if (u0 >= P.ne) continue;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int a = a_col0 + 16 * (RA * w + j) + m;
            if (a >= P.na) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) out[(size_t)(u0 + reg) * P.na + a] = acc[i][j][reg];
        }
    }
    if constexpr (SWAPPED) {
        if (does_bias && a_col0 + tid < P.na) P.slab_b[(size_t)split * P.na + a_col0 + tid] = bsum;
    } else {
        if (does_bias && e_col0 + tid < P.ne) P.slab_b[(size_t)split * P.ne + e_col0 + tid] = bsum;
    }
}

constexpr int kHeb6KB = 32;
// bf16 elements between the rows of two units in an LDS plane: 32 of data + 8 of padding.  With rows of exactly 64 B the sixteen lanes
// of a ds_read_b128 group (unit m = 0..15, same k chunk) start at banks 16 m mod 64 -- four lanes per bank quartet -- and the
// ds_write_b128s of the split pass collide four ways as well (SQ_LDS_BANK_CONFLICT = 0.50 of the LDS-active cycles, round 3 PMC); with
// 80 B they start at 20 m mod 64: sixteen different quartets, both ways.
constexpr int kHeb6LD = 40;            // spilled rows per stage = K of one bf16 MFMA

// ---- bf16x6 form of the tiled kernel (the default; tuning heb_fp32=1 selects the fp32-MFMA kernel above) ---------------------------
// The same GEMM with the spilled fp32 operands split into three bf16 pieces (the six products whose magnitude is above
// 2^-24 of the leading one, fp32 accumulation, small terms first; mcpc_bf16x6.h) on v_mfma_f32_16x16x32_bf16: against an fp64 sum its error is that of
// the fp32-MFMA kernel (max 2.7e-7 / rms 3.1e-8 of sum|terms| against 2.1e-7 / 2.7e-8 at K = 4096, scripts/heb_bf16_ubench.hip), at
// 1.66 x its rate on the same shapes (0.785 ms against 1.30 ms for the 784 x 256 flush of 64 steps on the whole chip).
//   global -> registers (one float4 per row and thread, a stage ahead) -> v_cvt_pk_bf16_f32 -> LDS planes -> MFMA operands:
// TE error tiles x 8 RA activation tiles per workgroup, 8 waves, wave w owns activation tiles RA w .. RA w + RA - 1.
// LDS: three bf16 planes of the stage's operands, TRANSPOSED: plane[p][unit][r], 32 r = 64 B of data per unit in rows of 80 B (kHeb6LD),
// so that the MFMA operand of lane (m, g) -- unit m, k = 8g..8g+7 -- is ONE conflict-free ds_read_b128.
// Split pass: thread (ug, c) takes four consecutive units x the 8-row chunk c: 8 float4 loads (a chunk's 16 lanes = 256 contiguous bytes
// of a spilled row), one ds_write_b128 per unit and plane.  512 units = 512 tasks = one per thread; a 17th error tile (TE = 17:
// 784 = 17 + 16 + 16 tiles) is 512 more elements = ONE per thread (row tid / 16, unit 16 TE' + tid % 16), written with ds_write_b16.
template <int TE, int RA>
__global__ __launch_bounds__(kHebThreads, 2) void mcpc_heb6_kernel(const HebArgs P) {
    constexpr int TA = 8 * RA;
    constexpr int TEM = TE >= 16 ? 16 : TE;             // error tiles handled by the float4 tasks
    constexpr bool XT = TE == 17;                       // one extra error tile handled element-wise
    static_assert(TE <= 17 && 4 * 4 * (TEM + TA) <= kHebThreads, "one task per thread");
    constexpr int NU = 16 * (TE + TA);                  // units (columns) per stage: E panel then A panel
    constexpr int PLANE = NU * kHeb6LD;                   // bf16 elements per plane
    extern __shared__ __attribute__((aligned(16))) unsigned short lds6[];       // [3][NU][32] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int total = P.n_mt * P.n_nt * P.ksplit;
    int id = blockIdx.x;
    if (total % 8 == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int per_split = P.n_mt * P.n_nt;
    const int split = id / per_split, rem = id - split * per_split;
    const int nt = rem / P.n_mt, mt = rem - nt * P.n_mt;
    const int e_col0 = P.e_col_base + mt * 16 * TE, a_col0 = nt * 16 * TA;
    const int r0 = split * P.rows_per_split;
    const int r1 = min(P.rows, r0 + P.rows_per_split);
    const int n_stage = (r1 - r0) / kHeb6KB;

    // the float4 task of this thread; LDS unit index: E tiles 0 .. TEM-1, [the extra tile TEM], then the A tiles
    const int ug = tid >> 2, c = tid & 3;
    const bool mine = ug < 4 * (TEM + TA);
    const bool is_a = 4 * ug >= 16 * TEM;
    const int col = is_a ? a_col0 + 4 * ug - 16 * TEM : e_col0 + 4 * ug;
    const int width = is_a ? P.na : P.ne;
    const bool on = mine && col < width;                  // (widths are multiples of 16: a group of four is in or out as a whole)
    const float* const src = (is_a ? P.A : P.E) + (size_t)(r0 + 8 * c) * width + (on ? col : 0);
    const int lunit = 4 * ug + ((XT && is_a) ? 16 : 0);
    const int loff = (mine ? lunit : 0) * kHeb6LD + 8 * c;  // element offset of the group's first unit inside a plane
    // the extra tile: element (row tid / 16, unit tid % 16)
    const int xr = tid >> 4, xu = tid & 15;
    const bool xon = XT && e_col0 + 16 * TEM + xu < P.ne;
    const float* const xsrc = P.E + (size_t)(r0 + xr) * P.ne + (xon ? e_col0 + 16 * TEM + xu : 0);
    const int xoff = (16 * TEM + xu) * kHeb6LD + xr;
    f32x4 v[8];
    float xv = 0.f;
    auto load_stage = [&](int s) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(s * kHeb6KB + j) * width));
        if constexpr (XT) xv = __builtin_nontemporal_load(xsrc + (size_t)s * kHeb6KB * P.ne);
    };
    f32x4 bsum = splat(0.f);
    float xbsum = 0.f;
    auto split_store = [&]() {
        if (mine) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {                // unit 4 ug + u: rows 8 c .. 8 c + 7 are v[0..7][u]
                u32x4 hi, mid, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = on ? v[2 * j][u] : 0.f, b = on ? v[2 * j + 1][u] : 0.f;
                    bsum[u] += a + b;
                    unsigned h, mm, ll;
                    split3_pair_fast(f32x2{a, b}, h, mm, ll);       // (9 instructions per pair instead of 13: mcpc_bf16x6.h)
                    hi[j] = h; mid[j] = mm; lo[j] = ll;
                }
                *reinterpret_cast<u32x4*>(lds6 + 0 * PLANE + loff + u * kHeb6LD) = hi;
                *reinterpret_cast<u32x4*>(lds6 + 1 * PLANE + loff + u * kHeb6LD) = mid;
                *reinterpret_cast<u32x4*>(lds6 + 2 * PLANE + loff + u * kHeb6LD) = lo;
            }
        }
        if constexpr (XT) {
            const float a = xon ? xv : 0.f;
            xbsum += a;
            const unsigned h = pk_bf16(a, 0.f);
            const float ra = a - bf16lo_f32(h);
            const unsigned mm = pk_bf16(ra, 0.f);
            const float sa = ra - bf16lo_f32(mm);
            lds6[0 * PLANE + xoff] = (unsigned short)h;
            lds6[1 * PLANE + xoff] = (unsigned short)mm;
            lds6[2 * PLANE + xoff] = (unsigned short)pk_bf16(sa, 0.f);
        }
    };

    f32x4 acc[TE][RA];
#pragma unroll
    for (int i = 0; i < TE; ++i)
#pragma unroll
        for (int j = 0; j < RA; ++j) acc[i][j] = splat(0.f);

    if (n_stage > 0) load_stage(0);
    const unsigned short* const base = lds6 + (size_t)m * kHeb6LD + 8 * g;       // lane (m, g) of tile t reads plane[p][16 t + m][8 g .. 8 g + 7]
    struct Op { u32x4 h, m, l; };
    auto ld_op = [&](int tile) {
        const unsigned short* p = base + (size_t)(16 * tile) * kHeb6LD;
        Op o;
        o.h = *reinterpret_cast<const u32x4*>(p);
        o.m = *reinterpret_cast<const u32x4*>(p + PLANE);
        o.l = *reinterpret_cast<const u32x4*>(p + 2 * PLANE);
        return o;
    };
    for (int s = 0; s < n_stage; ++s) {
        split_store();                                    // stage s: registers -> three bf16 planes in LDS
        __syncthreads();
        if (s + 1 < n_stage) load_stage(s + 1);            // travels during the MFMAs
        Op ao[RA];
#pragma unroll
        for (int j = 0; j < RA; ++j) ao[j] = ld_op(TE + RA * w + j);
        Op e = ld_op(0);
#pragma unroll
        for (int i = 0; i < TE; ++i) {
            // the next error tile's operands are requested before this tile's MFMAs (pinned: left alone hipcc sinks the reads)
            Op en = e;
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < TE) en = ld_op(i + 1);
            // six products per accumulator, small terms first, the accumulators of the tile alternating
#define H6(ep_, ap_) _Pragma("unroll") for (int j = 0; j < RA; ++j) acc[i][j] = mfma6(e.ep_, ao[j].ap_, acc[i][j])
            H6(m, m); H6(l, h); H6(h, l); H6(m, h); H6(h, m); H6(h, h);
#undef H6
            __builtin_amdgcn_sched_barrier(0);
            e = en;
        }
        __syncthreads();                                  // every wave is done with the planes of stage s
    }
    // C layout of tile (i, j): row 4 g + reg -> error unit, column m -> activation unit
    float* out = P.slab + (size_t)split * P.ne * P.na;
#pragma unroll
    for (int i = 0; i < TE; ++i) {
        const int u0 = e_col0 + 16 * i + 4 * g;
        if (u0 >= P.ne) continue;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int a = a_col0 + 16 * (RA * w + j) + m;
            if (a >= P.na) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) out[(size_t)(u0 + reg) * P.na + a] = acc[i][j][reg];
        }
    }
    // bias sums (column sums of E, workgroups of the first activation group only).  float4 tasks: the four chunk lanes of a unit
    // group are adjacent lanes.  Extra tile: 32 rows spread over tid / 16 -> through LDS.
    if (nt == 0) {
        f32x4 bb = bsum;
#pragma unroll
        for (int u = 0; u < 4; ++u) { float t = bb[u]; t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); bb[u] = t; }
        if (c == 0 && mine && !is_a && col < P.ne) *reinterpret_cast<f32x4*>(P.slab_b + (size_t)split * P.ne + col) = bb;
        if constexpr (XT) {
            float* red = reinterpret_cast<float*>(lds6);
            red[tid] = xbsum;
            __syncthreads();
            if (tid < 16 && e_col0 + 16 * TEM + tid < P.ne) {
                float t = 0.f;
                for (int r = 0; r < 32; ++r) t += red[16 * r + tid];
                P.slab_b[(size_t)split * P.ne + e_col0 + 16 * TEM + tid] = t;
            }
        }
    }
}

// ---- fixed-order slab reduction for every Linear of a flush in ONE launch ---------------------------------------------
struct ReduceJob {
    const float* slab;     // [ksplit][n]
    float* dst;            // [n]
    int n, ksplit;
    float sign;
    int tr_cols, tr_ld;    // tr_cols > 0: the slab holds the transposed matrix [n / tr_cols][tr_cols]; element idx goes to
                           // dst[(idx % tr_cols) * tr_ld + idx / tr_cols]
};
constexpr int kMaxReduceJobs = 2 * (kMaxLatent + 1);
struct ReduceJobs {
    ReduceJob job[kMaxReduceJobs];
    int n_jobs;
};

// dst[i] += sign * sum_k slab[k][i]; k ascending in groups of 8 independent loads -> bitwise reproducible.  blockIdx.y = job.
// Short vectors with many splits (bias slabs) take the wide form: the block's 4 waves each sum every 4th split of the same
// 64 elements, partial sums added in wave order through LDS.
__global__ __launch_bounds__(256) void mcpc_reduce_jobs_kernel(const ReduceJobs J) {
    __shared__ float part[4][64];
    const ReduceJob jb = J.job[blockIdx.y];
    const size_t n = (size_t)jb.n;
    if (jb.n <= 4096) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int base = blockIdx.x * 64; base < jb.n; base += gridDim.x * 64) {
            const int idx = base + lane;
            float s = 0.f;
            if (idx < jb.n)
                for (int k = wv; k < jb.ksplit; k += 4) s += __builtin_nontemporal_load(jb.slab + (size_t)k * n + idx);
            part[wv][lane] = s;
            __syncthreads();
            if (wv == 0 && idx < jb.n) {
                const int di = jb.tr_cols > 0 ? (idx % jb.tr_cols) * jb.tr_ld + idx / jb.tr_cols : idx;
                jb.dst[di] += jb.sign * ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
            }
            __syncthreads();
        }
        return;
    }
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= jb.ksplit; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(jb.slab + (size_t)(k + j) * n + idx);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < jb.ksplit; ++k) s += __builtin_nontemporal_load(jb.slab + (size_t)k * n + idx);
        const size_t di = jb.tr_cols > 0 ? (idx % jb.tr_cols) * (size_t)jb.tr_ld + idx / jb.tr_cols : idx;
        jb.dst[di] += jb.sign * s;
    }
}

}  // namespace mcpc
