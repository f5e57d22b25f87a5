// K3 -- Hebbian sums  G[u][i] = sum_r E[r][u] * A[r][i]  over the spilled (error, activation) rows, for the wide Linears.
// Replaces the parameter part of overall.backward() (reference predictive_coding/pc_trainer.py:862).
//
// A plain fp32 GEMM with a tiny output (ne x na, at most 784 x 256 at cfg-M) and a huge K (rows = steps x chains, 385 k per
// flush of 64 steps): split-K over the grid, every split writes its partial tile to a slab, the slabs are summed in a fixed
// order by mcpc_reduce_jobs_kernel (bitwise reproducible; no float atomics).
//
// (fp32-MFMA kernel mcpc_heb_kernel, rounds 2-3: row-major spill; kept behind tuning heb_fp32=1 as the independent form the bf16x6
// kernel mcpc_heb7_kernel below is checked against.)
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
    const unsigned* e_max;           // mcpc_heb7_kernel: bit pattern of the largest |value| in the E image of this segment, and in the A image:
    const unsigned* a_max;           //   the powers of two its fp16 operands are scaled by (written by the step kernels: KParams::spillmax)
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

// ---- fp16 form of the tiled kernel: mcpc_heb7_kernel (the default; tuning heb_fp32=1 selects the fp32-MFMA kernel above) ---------------------
// Round 4 ran this GEMM with the spilled fp32 operands split into three bf16 pieces and six v_mfma_f32_16x16x32_bf16 per product.  Round 5:
// TWO fp16 pieces per operand (22 significant bits) and THREE v_mfma_f32_16x16x32_f16 -- a_m b_h + a_h b_m + a_h b_h; the a_m b_m term,
// 2^-22 of the leading one, changes neither the maximal nor the rms error of a sum over this many rows (profiles/r05_f16x4_study.txt) --
// as in the step kernels' GEMM core (mcpc_gemm_f16.h), with the operands SCALED BY POWERS OF TWO so that they fit fp16's five exponent bits.
// Here the sum runs over the spilled ROWS, so a scale must be the same for every row: one exponent per operand IMAGE, taken from the largest
// |value| the step kernel spilled of that tensor in this segment (KParams::spillmax: every epilogue wave keeps the maximum of what it
// spills; HebArgs::e_max / a_max).  Values within 2^16 of their image's maximum keep all 22 bits; smaller ones keep an absolute error below
// 2^-39 of the maximum -- invisible in a sum that holds terms of the maximum's size.  The accumulators are un-scaled (exact) when the
// partial sums are written.  Against an fp64 sum the error is that of the fp32-MFMA kernel and of round 4's bf16x6 form (the tests'
// fp64 windows hold at unchanged tolerances).
//
// Round 3's form (mcpc_heb6_kernel) kept the planes of BOTH panels of a 32-row stage in LDS (126 KB: no second buffer), so every stage
// was "all threads split -> barrier -> all waves MFMA -> barrier": MFMA busy 58 %, LDS bank conflicts 0.41 of the LDS-active cycles
// (profiles/r03_pmc_summary.json), 17 us of whole-chip time per accumulating step at cfg-M for 9.4 us of MFMA work.  This form:
//   * reads a TILE-MAJOR spill (KLayer::spill_tm: [row tile of 16][unit tile of 16][q = unit quad][c = row][4 units], the layout in
//     which a step-kernel epilogue wave's float4-per-lane store is ONE contiguous KiB -- its row-major stores were sixteen 64-byte pieces
//     of sixteen rows, about 100 cycles of the CU's vector-memory path each, the path the step kernel is bound by);
//   * the ACTIVATION operand never touches LDS: wave w owns activation tiles RA w .. RA w + RA - 1 and no other wave needs them, so
//     lane (m, g) gathers its eight values (unit m, rows 4g..4g+3 of both row tiles of the stage) straight from global memory
//     (8 dwords per tile, the same 16 cache lines for all eight) and splits them in registers;
//   * the ERROR panel (all waves need all TE tiles) goes through LDS as two fp16 planes [unit][32 rows] in rows of 80 B,
//     DOUBLE-BUFFERED (2 x 43 KB at TE = 17): the split of stage s+1 and the MFMAs of stage s run in the same barrier interval, waves
//     0-3 splitting first and waves 4-7 multiplying first, so that on every SIMD one wave's conversion sits beside the other's MFMAs;
//   * a thread of the split pass holds (row c, row 16 + c) x 4 units -- the same lane of the two row tiles of one unit tile, two
//     coalesced float4 loads -- and writes one dword per unit and plane: k-slot pair (2c, 2c+1) = rows (c, 16 + c).  The MFMA's k order
//     is free as long as both operands use it, and with the 80-byte unit stride the 32 lanes of a ds_write_b32 group hit 32 different
//     banks (no conflicts), as do the 16 lanes of the operand's ds_read_b128.
constexpr int kHeb7KB = 32;            // spilled rows per stage = K of one bf16 MFMA = two row tiles
constexpr int kHeb7LD = 40;            // bf16 elements between the rows of two units in an LDS plane: 32 of data + 8 of padding (80 B)

// SWAPPED: the launch computes the TRANSPOSED product for a Linear with a narrow input (the E slot holds its activations, the A slot
// its errors); the bias sums (column sums of the errors) then come from the A operand, and the reduction writes the slab back transposed.
template <int TE, int RA, bool SWAPPED = false>
__global__ __launch_bounds__(kHebThreads, 2) void mcpc_heb7_kernel(const HebArgs P) {
    constexpr int TPW = (TE + 7) / 8;                   // error tiles a wave converts per stage (tile j = w + 8 i)
    constexpr int PLANE = 16 * TE * kHeb7LD;            // bf16 elements per plane
    constexpr int NPL = 2;                              // planes per buffer: the two fp16 pieces
    extern __shared__ __attribute__((aligned(16))) unsigned short lds7[];       // [2 buffers][2 planes][16 TE units][40] fp16
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;             // MFMA view: unit m of a tile, k group g
    const int c = lane & 15, q = lane >> 4;             // split view: row c of a row tile, unit quad q
    const int total = P.n_mt * P.n_nt * P.ksplit;
    int id = blockIdx.x;
    if (total % 8 == 0) id = (id & 7) * (total >> 3) + (id >> 3);      // consecutive logical ids (one K split) on one XCD
    const int per_split = P.n_mt * P.n_nt;
    const int split = id / per_split, rem = id - split * per_split;
    const int nt = rem / P.n_mt, mt = rem - nt * P.n_mt;
    const int e_col0 = P.e_col_base + mt * 16 * TE, a_col0 = nt * 16 * 8 * RA;
    const int net = P.ne / 16, nat = P.na / 16;         // unit tiles per row tile of the two images
    const int e_tile0 = e_col0 / 16, a_tile0 = a_col0 / 16 + RA * w;
    const int r0 = split * P.rows_per_split;
    const int r1 = min(P.rows, r0 + P.rows_per_split);
    const int n_stage = (r1 - r0) / kHeb7KB;
    const size_t rt_base = (size_t)r0 / 16;             // first row tile of this split (rows_per_split is a multiple of 32)
    // the powers of two the two operand images are scaled by: from the largest |value| the step kernels spilled of each in this segment
    const int e_exp = scale_exp_for_max(__uint_as_float(*P.e_max)), a_exp = scale_exp_for_max(__uint_as_float(*P.a_max));
    const float e_scale = pow2i(e_exp), a_scale = pow2i(a_exp);

    // ---- what this thread loads per stage ------------------------------------------------------------------------------------
    // Addresses = a wave-uniform 64-bit base (scalar registers, advanced by the scalar ALU) + ONE 32-bit byte offset per lane that never
    // changes: with per-lane 64-bit pointers the address arithmetic alone took ~45 VGPRs, and <16, 2> / <17, 2> (128 / 136 accumulators)
    // spilled 300-550 of them.
    typedef const char __attribute__((address_space(1)))* gb_t;
    bool e_on[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) e_on[i] = (w + 8 * i) < TE && e_tile0 + w + 8 * i < net;
    bool a_on[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) a_on[j] = a_tile0 + j < nat;
    const uint32_t e_lane = 16u * (uint32_t)lane;                                             // float4 `lane` of a tile
    const size_t e_rt_bytes = (size_t)net * 1024, a_rt_bytes = (size_t)nat * 1024;             // one row tile of the images
    f32x4 eraw[TPW][2];                                 // E: (row c | row 16 + c) x units 4q..4q+3 of tile w + 8 i
    f32x4 araw[RA][2];                                  // A: the same float4s of this wave's own tiles a_tile0 + j
    // The activation operand of lane (m, g) is unit m x rows (4g+i, 16+4g+i): a 16 x 16 transpose away from what a coalesced load
    // gives a lane (row c x 4 units).  It goes through a scratch of this WAVE's own in LDS (no other wave reads it: no barrier, only the
    // wave's own lgkmcnt): written as float4s, read back as 8 dwords per tile.  Quad q's 16 rows start 17 float4 slots apart, so
    // that the 32 lanes of a ds_read_b32 group ((q, g mod 2, m mod 4): dword 68 q + 16 g + 4 i + m mod 4) hit 32 different banks.
    // (Gathering the 8 dwords straight from global memory cost 7 % of the kernel: each such wave instruction touches 16 cache lines.)
    float* const a_scr = reinterpret_cast<float*>(lds7 + 2 * NPL * PLANE) + 16 * TE + (size_t)w * (2 * 272);     // (one tile's two row tiles: reused tile after tile)
    const int a_wr = 4 * (17 * q + c);                                                          // float offset of this lane's float4 in a tile image
    const int a_rd = 4 * (17 * (m >> 2) + 4 * g) + (m & 3);                                     // ... of (unit m, row 4 g)
    // (no load under a branch: hipcc joins the paths behind s_waitcnt vmcnt(0) -- a stage past the end re-reads the last one, a tile
    // past the image the first one, and what they return is dropped by a select)
    const int s_last = n_stage > 0 ? n_stage - 1 : 0;
    auto load_e = [&](int s_) {
#if defined(MCPC_HEB_EXP) && MCPC_HEB_EXP == 4      // timing experiment only: no global loads
        if (P.rows >= 0) return;
#endif
        const int s = s_ < s_last ? s_ : s_last;
        const gb_t base = (gb_t)P.E + (rt_base + 2 * (size_t)s) * e_rt_bytes;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const gb_t tb = base + (size_t)(e_on[i] ? e_tile0 + w + 8 * i : e_tile0) * 1024;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                // (the tile-past-the-image select is applied where the value is USED: a select here would make the wave wait for the load at once)
                eraw[i][b] = __builtin_nontemporal_load(reinterpret_cast<const gf32x4*>(tb + b * e_rt_bytes + e_lane));
            }
        }
    };
    auto load_a = [&](int s_) {
#if defined(MCPC_HEB_EXP) && MCPC_HEB_EXP == 4
        if (P.rows >= 0) return;
#endif
        const int s = s_ < s_last ? s_ : s_last;
        const gb_t base = (gb_t)P.A + (rt_base + 2 * (size_t)s) * a_rt_bytes;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const gb_t tb = base + (size_t)(a_on[j] ? a_tile0 + j : 0) * 1024;
#pragma unroll
            for (int b = 0; b < 2; ++b)
                araw[j][b] = __builtin_nontemporal_load(reinterpret_cast<const gf32x4*>(tb + b * a_rt_bytes + e_lane));
        }
    };
    // ---- bias sums: column sums of the errors (E panel; SWAPPED: the A operand) ----------------------------------------------
    // (E panel: per stage the 16 rows-pairs of a unit quad are summed across their DPP row and ONE lane adds the quad's four sums to a
    // running fp32 value in LDS -- a fixed lane, stage after stage: the same order in every run -- instead of 4 TPW accumulator registers
    // per thread in a kernel that has none to spare)
    float* const bias_lds = reinterpret_cast<float*>(lds7 + 2 * NPL * PLANE);          // [16 TE] behind the two plane buffers
    float asum[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) asum[j] = 0.f;
    const bool e_bias = !SWAPPED && nt == 0, a_bias = SWAPPED && mt == 0;

    // registers -> planes of buffer `buf`: tile j = w + 8 i, unit 16 j + 4 q + u, k-slot pair c
    auto split_store = [&](int buf, bool real) {        // real: a stage of the split (not the re-conversion past its end): its rows count for the bias
#if defined(MCPC_HEB_EXP) && MCPC_HEB_EXP == 2      // timing experiment only (wrong sums): no conversion, no plane stores
        if (P.rows >= 0) return;
#endif
        unsigned short* const base = lds7 + (size_t)buf * NPL * PLANE + (size_t)(4 * q) * kHeb7LD + 2 * c;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if ((w + 8 * i) >= TE) continue;
            const f32x4 v0 = e_on[i] ? eraw[i][0] : splat(0.f), v1 = e_on[i] ? eraw[i][1] : splat(0.f);
            if (e_bias && real) {
                f32x4 t = v0 + v1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v = t[u];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    t[u] = v;
                }
                if (c == 0) {
                    f32x4* const bp = reinterpret_cast<f32x4*>(bias_lds + 16 * (w + 8 * i) + 4 * q);
                    *bp = *bp + t;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned h, mm;
                split2_pair(f32x2{v0[u], v1[u]}, e_scale, h, mm);
                unsigned short* const dst = base + (size_t)(16 * (w + 8 * i) + u) * kHeb7LD;
                *reinterpret_cast<unsigned*>(dst) = h;
                *reinterpret_cast<unsigned*>(dst + PLANE) = mm;
            }
        }
    };
    struct Op { u32x4 h, m; };
    auto make_a = [&](int j) {
        // registers -> the wave's scratch -> (unit m, rows 4g+i of both row tiles)
#pragma unroll
        for (int b = 0; b < 2; ++b) *reinterpret_cast<f32x4*>(a_scr + b * 272 + a_wr) = a_on[j] ? araw[j][b] : splat(0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Op o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h, mm;
            const float a0 = a_scr[a_rd + 4 * i], a1 = a_scr[272 + a_rd + 4 * i];
            split2_pair(f32x2{a0, a1}, a_scale, h, mm);
            o.h[i] = h; o.m[i] = mm;
            if (a_bias) asum[j] += a0 + a1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the next tile's stores come after these reads)
        __builtin_amdgcn_wave_barrier();
        return o;
    };

    f32x4 acc[TE][RA];
#pragma unroll
    for (int i = 0; i < TE; ++i)
#pragma unroll
        for (int j = 0; j < RA; ++j) acc[i][j] = splat(0.f);

    // The three products of an error tile, small terms first: (m h) (h m) (h h).  Both planes of the NEXT tile are requested at the head of
    // this one (a second register set of 8): their LDS round trip hides behind this tile's 3 RA MFMAs and the partner wave's work.
    auto mfma_stage = [&](int buf, const Op (&ao)[RA]) {
#if defined(MCPC_HEB_EXP) && MCPC_HEB_EXP == 5      // timing experiment only: no MFMA phase
        if (P.rows >= 0) return;
#endif
        const unsigned short* const rd = lds7 + (size_t)buf * NPL * PLANE + (size_t)m * kHeb7LD + 8 * g;     // lane (m, g): plane[p][16 t + m][8 g ..]
        u32x4 eh = *reinterpret_cast<const u32x4*>(rd), em = *reinterpret_cast<const u32x4*>(rd + PLANE);
#pragma unroll
        for (int i = 0; i < TE; ++i) {
            const unsigned short* const nx = rd + (size_t)(16 * (i + 1 < TE ? i + 1 : i)) * kHeb7LD;
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 ehn = *reinterpret_cast<const u32x4*>(nx), emn = *reinterpret_cast<const u32x4*>(nx + PLANE);
#define H7(e_, ap_) _Pragma("unroll") for (int j = 0; j < RA; ++j) acc[i][j] = mfma4(e_, ao[j].ap_, acc[i][j])
            H7(em, h); H7(eh, m); H7(eh, h);
#undef H7
            __builtin_amdgcn_sched_barrier(0);
            eh = ehn; em = emn;
        }
    };

    // ---- pipeline: stage s multiplies out of buffer s & 1 while stage s + 1 is split into the other one ------------------------
    if (tid < 16 * TE) bias_lds[tid] = 0.f;
    __syncthreads();
    if (n_stage == 0) return;
    load_e(0);
    load_a(0);
    split_store(0, true);
    load_e(1);
    __syncthreads();
    // One wave of every SIMD converts while the other multiplies: waves 0-3 split stage s + 1 first and multiply stage s afterwards,
    // waves 4-7 the other way round (two copies of the loop, the same barrier count).  The conversion of stage s + 1 into the last
    // buffer past the end is harmless (it re-converts the last stage into the buffer nobody reads any more).
#if defined(MCPC_HEB_EXP) && MCPC_HEB_EXP == 3      // timing experiment: no stagger, every wave splits first
    if (w < 8) {
#else
    if (w < 4) {
#endif
        for (int s = 0; s < n_stage; ++s) {
            split_store((s + 1) & 1, s + 1 < n_stage);
            load_e(s + 2);
            Op ao[RA];
#pragma unroll
            for (int j = 0; j < RA; ++j) ao[j] = make_a(j);    // stage s (loaded one stage ago)
            load_a(s + 1);
            mfma_stage(s & 1, ao);
            __syncthreads();          // buffer (s + 1) & 1 is complete; every wave is done reading buffer s & 1
        }
    } else {
        for (int s = 0; s < n_stage; ++s) {
            Op ao[RA];
#pragma unroll
            for (int j = 0; j < RA; ++j) ao[j] = make_a(j);
            load_a(s + 1);
            mfma_stage(s & 1, ao);
            split_store((s + 1) & 1, s + 1 < n_stage);
            load_e(s + 2);
            __syncthreads();
        }
    }

    // C layout of tile (i, j): row 4 g + reg -> error unit, column m -> activation unit; the sums leave their scaled units here (exact)
    const float unscale = pow2i(-e_exp) * pow2i(-a_exp);
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
            for (int reg = 0; reg < 4; ++reg) out[(size_t)(u0 + reg) * P.na + a] = acc[i][j][reg] * unscale;
        }
    }
    if (e_bias) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if ((w + 8 * i) >= TE || !e_on[i]) continue;
            if (c == 0) *reinterpret_cast<f32x4*>(P.slab_b + (size_t)split * P.ne + 16 * (e_tile0 + w + 8 * i) + 4 * q) =
                            *reinterpret_cast<const f32x4*>(bias_lds + 16 * (w + 8 * i) + 4 * q);
        }
    }
    if (a_bias) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            float v = asum[j];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (g == 0 && a_on[j]) P.slab_b[(size_t)split * P.na + 16 * (a_tile0 + j) + m] = v;
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
