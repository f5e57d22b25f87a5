// Round-5 go / no-go measurement for the step kernel's decomposition (NOT used by the product).
//
// Question: does a 32-chain workgroup -- every weight fragment serving TWO chain tiles, 4 GEMM waves + 8 epilogue waves -- bring a
// 6000-chain step of cfg-M to <= 40 us (today: 16-chain workgroups, 4 G + 4 E waves, 52-54 us = 1.5 rounds of a 35 us workgroup-step)?
// The kernel below is a MODEL of the step kernel's two roles with the product's own arithmetic pieces (the packed three-plane fragment
// stream out of L2, fp32 B rows in LDS split in the GEMM wave with mcpc::split3_pair_fast, six v_mfma_f32_16x16x32_bf16 per tile and
// k-block, accumulator blocks handed over through LDS with progress counters) and cfg-M's table: per step and workgroup the read-out in
// four chunks (HF: 49 unit tiles x 8 k-blocks; HB: 16 x 24.5), FWD_2 (16 x 8), FWD_1 (16 x 1), BWD_2 (16 x 8), BWD_1 (2 x 8) = 3.24 MB
// of fragments.  E waves wait for an entry, run EW VALU operations per float4 of it (a mix of fp32 / integer / transcendental work like
// the epilogues': errors, energies, loss, Philox + Box-Muller, the x update) and publish; a G wave stores a block only when the E waves
// are at most RING entries behind and starts a step only when the previous step's x update (entry BWD_2) is through.
//   CT = chain tiles per fragment (1: today's form, 2: the candidate), NE = E waves per SIMD, PRE = B operand pre-split by the producer
//   (three ds_read_b128 per chain tile, no split in the GEMM wave).
// Calibration: CT = 1, NE = 1 must land near the product's measured 33-36 us per workgroup-step (and its NOEPI / NOGEMM builds near
// 34.8 / 19.6: profiles/r04_k1_bounds.txt) for the CT = 2 lines to mean anything -- EW is chosen for that.
//   hipcc --offload-arch=gfx950 -O3 -Imontecarlopredictivecoding_amd/csrc -Iscripts scripts/k1_decomp_ubench.hip -o scripts/bin/k1_decomp_ubench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "mcpc_bf16x6.h"
using namespace mcpc;
// F16 = 1 (default since the product moved to the fp16 pipe in the second half of round 5): two fp16 planes per fragment, a 24-instruction
// split, three v_mfma_f32_16x16x32_f16 per tile and k-block;  -DF16=0: the bf16x6 arithmetic the go / no-go of profiles/r05_k1_decomp.txt
// was measured with (three planes, six MFMAs).
#ifndef F16
#define F16 1
#endif
constexpr int kPlanes = F16 ? 2 : 3;
constexpr int kFragStride = 64 * kPlanes;                        // u32x4 units per (tile, k-block)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2_pair_f16(f32x2 x, float sc, unsigned& h, unsigned& m) {
    const f32x2 xs = x * sc;
    const f16x2_t hh = __builtin_convertvector(xs, f16x2_t);
    const f32x2 r = xs - __builtin_convertvector(hh, f32x2);
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
__device__ __forceinline__ f32x4 mfma3(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

struct Entry { int base; int nkb; int tiles; int items; };      // base: u32x4 offset of the entry's fragments; items: float4s per chain tile
constexpr int kEntries = 12;
struct Table { Entry e[kEntries]; int xupd; };                   // xupd: the entry whose epilogue the NEXT step's first GEMM waits for
constexpr int LDB = 264;                                          // B row stride (floats)
constexpr int RING = 3;

struct Frag { u32x4 h, m, l; };
__device__ __forceinline__ Frag split8(f32x4 x0, f32x4 x1) {
    unsigned h[4], m[4], l[4];
#if F16
    const float sc = 4096.f;                                      // (the product takes the row's scale from a word its producers keep)
    split2_pair_f16(f32x2{x0.x, x0.y}, sc, h[0], m[0]); split2_pair_f16(f32x2{x0.z, x0.w}, sc, h[1], m[1]);
    split2_pair_f16(f32x2{x1.x, x1.y}, sc, h[2], m[2]); split2_pair_f16(f32x2{x1.z, x1.w}, sc, h[3], m[3]);
    l[0] = l[1] = l[2] = l[3] = 0u;
    Frag f16f;
    f16f.h = u32x4{h[0], h[1], h[2], h[3]}; f16f.m = u32x4{m[0], m[1], m[2], m[3]}; f16f.l = u32x4{0u, 0u, 0u, 0u};
    return f16f;
#endif
    split3_pair_fast(f32x2{x0.x, x0.y}, h[0], m[0], l[0]);
    split3_pair_fast(f32x2{x0.z, x0.w}, h[1], m[1], l[1]);
    split3_pair_fast(f32x2{x1.x, x1.y}, h[2], m[2], l[2]);
    split3_pair_fast(f32x2{x1.z, x1.w}, h[3], m[3], l[3]);
    Frag f;
    f.h = u32x4{h[0], h[1], h[2], h[3]}; f.m = u32x4{m[0], m[1], m[2], m[3]}; f.l = u32x4{l[0], l[1], l[2], l[3]};
    return f;
}

__device__ __forceinline__ int lds_load_i(const volatile int* p) { return *p; }
__device__ __forceinline__ void spin_until(const volatile int* p, int want) {
    int guard = 0;
    while (lds_load_i(p) < want && ++guard < (1 << 22)) __builtin_amdgcn_s_sleep(1);
}

// one entry's GEMM for NT tiles: fragments one k-block ahead in two register sets
template <int NT, int CT, bool PRE>
__device__ __forceinline__ void gemm_entry(f32x4 (&acc)[4][CT], const gu32x4* A, int base, int tile0, int nkb, const float* lds_b, int lane) {
    const int c = lane & 15, g = lane >> 4;
    Frag aP[NT], aQ[NT];
    f32x4 bP[CT][3], bQ[CT][3];
    int aoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) aoff[t] = base + (tile0 + t) * nkb * kFragStride + lane;
#define LOADSET(a_, b_, k_)                                                                                    \
    do {                                                                                                       \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                       \
            a_[t].h = A[aoff[t] + (k_) * kFragStride]; a_[t].m = A[aoff[t] + (k_) * kFragStride + 64];         \
            if (!F16) a_[t].l = A[aoff[t] + (k_) * kFragStride + 128]; }                                       \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                    \
            const float* r_ = lds_b + (16 * ct + c) * LDB + ((k_) & 7) * 32 + 8 * g;                           \
            b_[ct][0] = *(const f32x4*)r_; b_[ct][1] = *(const f32x4*)(r_ + 4);                                \
            if (PRE) b_[ct][2] = *(const f32x4*)(r_ + 16 * CT * LDB); }                                        \
    } while (0)
#define BLOCK(a_, b_)                                                                                          \
    do {                                                                                                       \
        Frag B_[CT];                                                                                           \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                    \
            if (PRE) { B_[ct].h = __builtin_bit_cast(u32x4, b_[ct][0]); B_[ct].m = __builtin_bit_cast(u32x4, b_[ct][1]);      \
                       B_[ct].l = __builtin_bit_cast(u32x4, b_[ct][2]); }                                      \
            else B_[ct] = split8(b_[ct][0], b_[ct][1]); }                                                      \
        if (F16) {                                                                                             \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma3(a_[t].m, B_[ct].h, acc[t][ct]); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma3(a_[t].h, B_[ct].m, acc[t][ct]); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma3(a_[t].h, B_[ct].h, acc[t][ct]); \
        } else {                                                                                               \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma6(a_[t].m, B_[ct].m, acc[t][ct]); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma6(a_[t].l, B_[ct].h, acc[t][ct]); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma6(a_[t].h, B_[ct].l, acc[t][ct]); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma6(a_[t].m, B_[ct].h, acc[t][ct]); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma6(a_[t].h, B_[ct].m, acc[t][ct]); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma6(a_[t].h, B_[ct].h, acc[t][ct]); \
        }                                                                                                      \
    } while (0)
    LOADSET(aP, bP, 0);
    int kb = 0;
    for (; kb + 2 <= nkb; kb += 2) {
        __builtin_amdgcn_sched_barrier(0);
        LOADSET(aQ, bQ, kb + 1); BLOCK(aP, bP);
        __builtin_amdgcn_sched_barrier(0);
        LOADSET(aP, bP, (kb + 2 < nkb ? kb + 2 : kb + 1)); BLOCK(aQ, bQ);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kb < nkb) BLOCK(aP, bP);
#undef LOADSET
#undef BLOCK
}

// MODE bit 0: G waves run their GEMMs; bit 1: E waves run their arithmetic
template <int CT, int NE, bool PRE, int MODE, int NG = 1>
__global__ __launch_bounds__(256 * NG + 256 * NE, 1) void kmodel(const u32x4* __restrict__ Ag, Table tab, float* out, int steps, int ew) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_b = lds;                                             // B rows (fp32; PRE: + a third 16-byte image behind them)
    float* lds_c = lds + 16 * CT * LDB * (PRE ? 2 : 1);             // accumulator blocks handed over: [16 tiles][CT][64 lanes][4]
    volatile int* prog = (volatile int*)(lds_c + 16 * CT * 256);    // [0]: G blocks published, [1]: E entries finished, [2]: E arrivals
    const gu32x4* A = (const gu32x4*)Ag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * CT * LDB * (PRE ? 2 : 1); i += blockDim.x) lds_b[i] = (i % 97) * 1e-2f - 0.3f;
    if (threadIdx.x < 3) prog[threadIdx.x] = 0;
    __syncthreads();
    const int total = steps * kEntries;
    constexpr int GW = 4 * NG;                                        // G waves (NG per SIMD)
    if (wave < GW) {                                                  // ---- G role
        f32x4 sink = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0, n = 0; s < steps; ++s) {
            for (int p = 0; p < kEntries; ++p, ++n) {
                const Entry en = tab.e[p];
                const int nt = en.tiles / GW + (wave < en.tiles % GW ? 1 : 0);            // 13 tiles over 4 waves -> 4 3 3 3
                const int tile0 = wave * (en.tiles / GW) + (wave < en.tiles % GW ? wave : en.tiles % GW);
                f32x4 acc[4][CT];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p == 0 && s > 0) spin_until(prog + 1, (s - 1) * kEntries + tab.xupd + 1);     // FX of the previous step's update
                if (MODE & 1) {
                    if (nt == 4) gemm_entry<4, CT, PRE>(acc, A, en.base, tile0, en.nkb, lds_b, lane);
                    else if (nt == 3) gemm_entry<3, CT, PRE>(acc, A, en.base, tile0, en.nkb, lds_b, lane);
                    else if (nt == 2) gemm_entry<2, CT, PRE>(acc, A, en.base, tile0, en.nkb, lds_b, lane);
                    else if (nt == 1) gemm_entry<1, CT, PRE>(acc, A, en.base, tile0, en.nkb, lds_b, lane);
                }
                spin_until(prog + 1, n - RING + 1);                  // the ring slot this block goes to has been consumed
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        if (t < nt) *(f32x4*)(lds_c + ((((wave * 4 + t) & 15) * CT + ct) * 64 + lane) * 4) = acc[t][ct];
                __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
                if (lane == 0) __hip_atomic_fetch_add((int*)prog, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                sink += acc[0][0];
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = sink.x;
        return;
    }
    // ---- E role: entry n is ready when all four G waves have published it
    const int ew_id = wave - GW, n_e = 4 * NE;
    float keep = 0.f;
    unsigned ctr = threadIdx.x * 2654435761u;
    for (int n = 0; n < total; ++n) {
        const Entry en = tab.e[n % kEntries];
        spin_until(prog, GW * (n + 1));
        if (MODE & 2) {
            const int items = en.items * CT;                         // float4s of this entry
            for (int it = ew_id * 64 + lane; it < items; it += n_e * 64) {
                f32x4 v = *(f32x4*)(lds_c + (it % (16 * CT * 64)) * 4);
                // the epilogues' instruction mix: per 8 operations 4 fp32 (fma / mul / add), 3 integer (Philox-like mad_u64 + xor), 1 transcendental
                for (int r = 0; r < ew; r += 8) {
                    v.x = __builtin_fmaf(v.x, 0.999f, v.y); v.y = v.y * 1.0001f + 0.5f; v.z = __builtin_fmaf(v.z, v.x, 0.25f); v.w = v.w - v.z * 0.125f;
                    const unsigned long long pr = (unsigned long long)ctr * 0xD2511F53u;
                    ctr = (unsigned)(pr >> 32) ^ (unsigned)pr ^ 0x9E3779B9u;
                    v.w += __builtin_amdgcn_exp2f(-__builtin_fabsf(v.y)) * (float)(ctr & 1u);
                }
                *(f32x4*)(lds_b + (it % (16 * CT * (LDB / 4))) * 4) = v;          // the next GEMM's B rows
                keep += v.x;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // all E waves of the workgroup finish the entry together: the last one to arrive publishes
        if (lane == 0) {
            const int arrived = __hip_atomic_fetch_add((int*)(prog + 2), 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
            if (arrived == n_e * (n + 1)) __hip_atomic_store((int*)(prog + 1), n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}

static Table make_table(size_t* frag_units) {
    // (unit tiles, k-blocks) of cfg-M's table, dealt to the four G waves; items = float4s of a 16-chain epilogue of that entry
    struct { int tiles, nkb; } spec[kEntries] = {{13, 8}, {12, 8}, {12, 8}, {16, 7}, {12, 8}, {16, 6}, {16, 6}, {16, 6},     // HF0 HF1 HF2 HB0 HF3 HB1 HB2 HB3
                                                 {16, 8}, {16, 1}, {16, 8}, {2, 8}};                                       // FWD_2 FWD_1 BWD_2 BWD_1
    Table t;
    size_t off = 0;
    for (int p = 0; p < kEntries; ++p) {
        t.e[p].base = (int)off; t.e[p].nkb = spec[p].nkb;
        t.e[p].tiles = spec[p].tiles;
        t.e[p].items = spec[p].tiles * 64;                            // 16 x 16 outputs per tile = 64 float4
        off += (size_t)spec[p].tiles * spec[p].nkb * kFragStride;
    }
    // HB entries carry no epilogue of their own beyond the hand-off (the back-projection stays in registers): a token amount
    for (int p : {3, 5, 6, 7}) t.e[p].items = 64;
    t.xupd = 10;
    *frag_units = off;
    return t;
}

template <int CT, int NE, bool PRE, int MODE, int NG = 1> double run(const char* name, int nwg, int ew, const u32x4* A, const Table& tab, float* out) {
    const int steps = 200;
    const size_t lds = (size_t)(16 * CT * LDB * (PRE ? 2 : 1) + 16 * CT * 256 + 16) * 4;
    hipFuncSetAttribute((const void*)kmodel<CT, NE, PRE, MODE, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kmodel<CT, NE, PRE, MODE, NG>), dim3(nwg), dim3(256 * NG + 256 * NE), lds, 0, A, tab, out, 10, ew);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kmodel<CT, NE, PRE, MODE, NG>), dim3(nwg), dim3(256 * NG + 256 * NE), lds, 0, A, tab, out, steps, ew);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return 0; }
    const double us = best * 1e3 / steps;
    printf("%-78s WGs=%3d chains/WG=%2d E waves=%d EW=%3d  %6.2f us per workgroup-step  = %6.2f us per 16 chains\n", name, nwg, 16 * CT, 4 * NE, ew, us, us / CT);
    fflush(stdout);
    return us;
}

// Round 6: the MIXED shard -- 32-chain and 16-chain units of ONE 6000-chain shard side by side on the chip, more units
// than CUs in total under a round schedule.  Both forms run here at the same time on disjoint CUs (two streams; u32 + u16 = 256 resident
// workgroups, step counts chosen so that both kernels take about the same time), which gives each form's cost per unit-step WITH the other
// beside it (shared L2s, shared clock); a schedule that keeps all 256 CUs busy then needs (n32 c32 + n16 c16) / 256 per step of the shard.
template <int NE32>
static void run_mixed(int u32, int u16, int ew, const u32x4* A, const Table& tab, float* out, double calib) {
    const int s32 = 140, s16 = 200;
    const size_t lds32 = (size_t)(16 * 2 * LDB + 16 * 2 * 256 + 16) * 4, lds16 = (size_t)(16 * LDB + 16 * 256 + 16) * 4;
    hipFuncSetAttribute((const void*)kmodel<2, NE32, false, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32);
    hipFuncSetAttribute((const void*)kmodel<1, 1, false, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    float* out2 = out + (size_t)128 * 1024;
    double best32 = 1e30, best16 = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(a0, sa);
        hipLaunchKernelGGL((kmodel<2, NE32, false, 3, 1>), dim3(u32), dim3(256 + 256 * NE32), lds32, sa, A, tab, out, rep ? s32 : 10, ew);
        hipEventRecord(a1, sa);
        hipEventRecord(b0, sb);
        hipLaunchKernelGGL((kmodel<1, 1, false, 3, 1>), dim3(u16), dim3(512), lds16, sb, A, tab, out2, rep ? s16 : 10, ew);
        hipEventRecord(b1, sb);
        hipEventSynchronize(a1); hipEventSynchronize(b1);
        if (!rep) continue;
        float ma, mb; hipEventElapsedTime(&ma, a0, a1); hipEventElapsedTime(&mb, b0, b1);
        best32 = std::min(best32, ma * 1e3 / s32); best16 = std::min(best16, mb * 1e3 / s16);
    }
    printf("mixed launch: %3d x 32 chains (4 G + %d E) beside %3d x 16 chains, EW=%3d: %6.2f us per 32-chain unit-step, %6.2f per 16-chain unit-step (alone: see above)\n",
           u32, 4 * NE32, u16, ew, best32, best16);
    for (int n32 : {0, 60, 100, 140, 188}) {
        const int n16 = (6000 - 32 * n32 + 15) / 16;
        const double per_step = (n32 * best32 + n16 * best16) / 256.0;
        printf("   6000 chains as %3d x 32 + %3d x 16 units (%3d units), every CU busy: %6.2f us per step in the model = %6.2f at the product's calibration (x %.3f)\n",
               n32, n16, n32 + n16, per_step, per_step * calib, calib);
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    size_t units;
    const Table tab = make_table(&units);
    u32x4* A; float* out;
    hipMalloc(&A, units * 16); hipMemset(A, 0x3c, units * 16);
    hipMalloc(&out, (size_t)256 * 1024 * 4);
    printf("F16 = %d: fragment stream per workgroup-step: %.2f MB (product: %s MB)\n", F16, units * 16 / 1e6, F16 ? "2.19" : "3.29");
    // --- calibration: today's form.  The product measures 35.2 (all), 34.8 (NOEPI), 19.6 (NOGEMM), 4.8 (neither) us per workgroup-step
    for (int ew : {96, 160, 224}) {
        run<1, 1, false, 3>("16 chains, 4 G + 4 E (today's form)", 256, ew, A, tab, out);
        run<1, 1, false, 2>("16 chains, E waves alone (product's NOGEMM: 19.6)", 256, ew, A, tab, out);
    }
    run<1, 1, false, 1>("16 chains, G waves alone (product's NOEPI: 34.8)", 256, 0, A, tab, out);
    // (skeleton only -- neither role computes: 5.09 us in the first run of this file, product 4.8; hipcc 7.2 rejects that instantiation
    //  after the NT = 2 case was added: "Operand has incorrect register class" on the constant accumulator stores)
    run<1, 1, true, 1>("16 chains, G alone, B pre-split by the producer", 256, 0, A, tab, out);
    // --- the candidate: every fragment serves two chain tiles
    for (int ew : {96, 160, 224}) {
        run<2, 2, false, 3>("32 chains, 4 G + 8 E", 188, ew, A, tab, out);
        run<2, 1, false, 3>("32 chains, 4 G + 4 E", 188, ew, A, tab, out);
        run<2, 2, true, 3>("32 chains, 4 G + 8 E, B pre-split by the producer", 188, ew, A, tab, out);
    }
    run<2, 2, false, 1>("32 chains, G waves alone", 188, 0, A, tab, out);
    run<2, 2, true, 1>("32 chains, G waves alone, B pre-split", 188, 0, A, tab, out);
    run<2, 2, false, 1>("32 chains, G waves alone, all 256 CUs", 256, 0, A, tab, out);
    run<2, 2, false, 2>("32 chains, 8 E waves alone", 188, 160, A, tab, out);
    // --- today's 16 chains, but TWO G waves per SIMD (half the tiles each) with the B operand pre-split by the producer: the combination
    //     of round 4's prototypes (vi) and (ix), which were only measured apart (EW + 18: the producer's split per float4)
    run<1, 1, true, 1, 2>("16 chains, 8 G alone, B pre-split", 256, 0, A, tab, out);
    run<1, 1, false, 1, 2>("16 chains, 8 G alone, split in the G waves", 256, 0, A, tab, out);
    for (int ew : {96, 160}) {
        run<1, 1, true, 3, 2>("16 chains, 8 G + 4 E, B pre-split", 256, ew + 18, A, tab, out);
        run<1, 1, true, 3, 1>("16 chains, 4 G + 4 E, B pre-split", 256, ew + 18, A, tab, out);
        run<1, 1, false, 3, 2>("16 chains, 8 G + 4 E, split in the G waves (round 4's (vi))", 256, ew, A, tab, out);
    }
    run<2, 1, true, 3, 2>("32 chains, 8 G + 4 E, B pre-split", 188, 96 + 18, A, tab, out);
    // --- round 6: the mixed shard (calibration: the product's 26.9 us per 16-chain unit-step over the model's figure at EW = 96)
    {
        const double m16 = run<1, 1, false, 3>("16 chains, 4 G + 4 E (today's form), calibration of the mixed shard", 256, 96, A, tab, out);
        run_mixed<2>(100, 156, 96, A, tab, out, 26.9 / m16);
        run_mixed<2>(128, 128, 96, A, tab, out, 26.9 / m16);
    }
    return 0;
}
