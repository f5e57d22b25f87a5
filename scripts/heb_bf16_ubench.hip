// Round-3 study: the Hebbian GEMM  G[u][i] = sum_r E[r][u] A[r][i]  (K3, csrc/mcpc_hebbian.h) with fp32 operands split into three
// bf16 pieces and six v_mfma_f32_16x16x32_bf16 per 32-deep block ("bf16x6": hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; what is
// dropped is below 2^-24 relative), against the shipped fp32-MFMA kernel on the same shapes, and both against an fp64 host sum.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/heb_bf16_ubench.hip -o scripts/bin/heb_bf16_ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../montecarlopredictivecoding_amd/csrc/mcpc_device.h"
namespace mcpc {
constexpr int kMaxLatent = 6;
__device__ __forceinline__ f32x4 splat(float v) { f32x4 r = {v, v, v, v}; return r; }
}
#include "../montecarlopredictivecoding_amd/csrc/mcpc_hebbian.h"
using namespace mcpc;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kH6KB = 32;            // spilled rows per stage = K of one bf16 MFMA
constexpr int kH6Threads = 512;

// pack two floats into bf16x2 (round to nearest even), low half = a
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo_to_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// TE error tiles x 16 activation tiles per workgroup, 8 waves, wave w owns activation tiles 2w, 2w+1.
// LDS: three bf16 planes of the stage's operands, TRANSPOSED: plane[p][unit][r], 32 r = 64 B per unit, so that the MFMA operand of
// lane (m, g) -- unit m, k = 8g..8g+7 -- is ONE ds_read_b128 and a wave reads 1 KiB linearly (conflict-free).
// Split pass: thread (ug, c) takes four consecutive units x the 8-row chunk c: 8 float4 loads (a chunk's 16 lanes = 256 contiguous bytes
// of a spilled row), one ds_write_b128 per unit and plane.  512 units = 512 tasks = one per thread; a 17th error tile (TE = 17:
// 784 = 17 + 16 + 16 tiles) is 512 more elements = ONE per thread (row tid / 16, unit 16 TE' + tid % 16), written with ds_write_b16.
template <int TE>
__global__ __launch_bounds__(kH6Threads, 2) void heb6_kernel(const HebArgs P) {
    constexpr int TA = 16, RA = 2;
    constexpr int TEM = TE >= 16 ? 16 : TE;             // error tiles handled by the float4 tasks
    constexpr bool XT = TE == 17;                       // one extra error tile handled element-wise
    static_assert(TE <= 17 && 4 * 4 * (TEM + TA) <= kH6Threads, "one task per thread");
    constexpr int NU = 16 * (TE + TA);                  // units (columns) per stage: E panel then A panel
    constexpr int PLANE = NU * kH6KB;                   // bf16 elements per plane
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];       // [3][NU][32] bf16
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
    const int n_stage = (r1 - r0) / kH6KB;

    // the float4 task of this thread; LDS unit index: E tiles 0 .. TEM-1, [the extra tile TEM], then the A tiles
    const int ug = tid >> 2, c = tid & 3;
    const bool mine = ug < 4 * (TEM + TA);
    const bool is_a = 4 * ug >= 16 * TEM;
    const int col = is_a ? a_col0 + 4 * ug - 16 * TEM : e_col0 + 4 * ug;
    const int width = is_a ? P.na : P.ne;
    const bool on = mine && col < width;                  // (widths are multiples of 16: a group of four is in or out as a whole)
    const float* const src = (is_a ? P.A : P.E) + (size_t)(r0 + 8 * c) * width + (on ? col : 0);
    const int lunit = 4 * ug + ((XT && is_a) ? 16 : 0);
    const int loff = (mine ? lunit : 0) * kH6KB + 8 * c;  // element offset of the group's first unit inside a plane
    // the extra tile: element (row tid / 16, unit tid % 16)
    const int xr = tid >> 4, xu = tid & 15;
    const bool xon = XT && e_col0 + 16 * TEM + xu < P.ne;
    const float* const xsrc = P.E + (size_t)(r0 + xr) * P.ne + (xon ? e_col0 + 16 * TEM + xu : 0);
    const int xoff = (16 * TEM + xu) * kH6KB + xr;
    f32x4 v[8];
    float xv = 0.f;
    auto load_stage = [&](int s) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(s * kH6KB + j) * width));
        if constexpr (XT) xv = __builtin_nontemporal_load(xsrc + (size_t)s * kH6KB * P.ne);
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
                    const unsigned h = pk_bf16(a, b);
                    const float ra = a - bf16_lo_to_f32(h), rb = b - bf16_hi_to_f32(h);
                    const unsigned mm = pk_bf16(ra, rb);
                    const float sa = ra - bf16_lo_to_f32(mm), sb = rb - bf16_hi_to_f32(mm);
                    hi[j] = h; mid[j] = mm; lo[j] = pk_bf16(sa, sb);
                }
                *reinterpret_cast<u32x4*>(lds + 0 * PLANE + loff + u * kH6KB) = hi;
                *reinterpret_cast<u32x4*>(lds + 1 * PLANE + loff + u * kH6KB) = mid;
                *reinterpret_cast<u32x4*>(lds + 2 * PLANE + loff + u * kH6KB) = lo;
            }
        }
        if constexpr (XT) {
            const float a = xon ? xv : 0.f;
            xbsum += a;
            const unsigned h = pk_bf16(a, 0.f);
            const float ra = a - bf16_lo_to_f32(h);
            const unsigned mm = pk_bf16(ra, 0.f);
            const float sa = ra - bf16_lo_to_f32(mm);
            lds[0 * PLANE + xoff] = (unsigned short)h;
            lds[1 * PLANE + xoff] = (unsigned short)mm;
            lds[2 * PLANE + xoff] = (unsigned short)pk_bf16(sa, 0.f);
        }
    };

    f32x4 acc[TE][RA];
#pragma unroll
    for (int i = 0; i < TE; ++i)
#pragma unroll
        for (int j = 0; j < RA; ++j) acc[i][j] = splat(0.f);

    if (n_stage > 0) load_stage(0);
    const unsigned short* const base = lds + (size_t)m * kH6KB + 8 * g;       // lane (m, g) of tile t reads plane[p][16 t + m][8 g .. 8 g + 7]
    struct Op { bf16x8 h, m, l; };
    auto ld_op = [&](int tile) {
        const unsigned short* p = base + (size_t)(16 * tile) * kH6KB;
        Op o;
        o.h = *reinterpret_cast<const bf16x8*>(p);
        o.m = *reinterpret_cast<const bf16x8*>(p + PLANE);
        o.l = *reinterpret_cast<const bf16x8*>(p + 2 * PLANE);
        return o;
    };
    for (int s = 0; s < n_stage; ++s) {
#ifndef H6_NOSPLIT
        split_store();                                    // stage s: registers -> three bf16 planes in LDS
#endif
        __syncthreads();
        if (s + 1 < n_stage) load_stage(s + 1);            // travels during the MFMAs
        const Op a0 = ld_op(TE + RA * w), a1 = ld_op(TE + RA * w + 1);
        Op e = ld_op(0);
#ifndef H6_NOMFMA
#pragma unroll
        for (int i = 0; i < TE; ++i) {
            // the next error tile's operands are requested before this tile's MFMAs (pinned: left alone hipcc sinks the reads)
            Op en = e;
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < TE) en = ld_op(i + 1);
            // six products per accumulator, small terms first, the two accumulators alternating
#define H6(eo_, ao_, j_) acc[i][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(eo_, ao_, acc[i][j_], 0, 0, 0)
            H6(e.m, a0.m, 0); H6(e.m, a1.m, 1);
            H6(e.l, a0.h, 0); H6(e.l, a1.h, 1);
            H6(e.h, a0.l, 0); H6(e.h, a1.l, 1);
            H6(e.m, a0.h, 0); H6(e.m, a1.h, 1);
            H6(e.h, a0.m, 0); H6(e.h, a1.m, 1);
            H6(e.h, a0.h, 0); H6(e.h, a1.h, 1);
#undef H6
            __builtin_amdgcn_sched_barrier(0);
            e = en;
        }
#else
        acc[0][0] += __builtin_bit_cast(f32x4, a0.h) + __builtin_bit_cast(f32x4, a1.l) + __builtin_bit_cast(f32x4, e.m);
#endif
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
            float* red = reinterpret_cast<float*>(lds);
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

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }

template <int TE>
void launch6(const HebArgs& a) {
    const int lds_bytes = 3 * 16 * (TE + 16) * kH6KB * 2;
    hipFuncSetAttribute((const void*)heb6_kernel<TE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL((heb6_kernel<TE>), dim3(a.n_mt * a.n_nt * a.ksplit), dim3(kH6Threads), lds_bytes, 0, a);
}
template <int TE>
void launch32(const HebArgs& a) {
    const int lds_bytes = 2 * kHebKB * (heb_lds_stride(16 * TE) + heb_lds_stride(256)) * 4;
    hipFuncSetAttribute((const void*)mcpc_heb_kernel<TE, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL((mcpc_heb_kernel<TE, 2, false>), dim3(a.n_mt * a.n_nt * a.ksplit), dim3(kHebThreads), lds_bytes, 0, a);
}

int main(int argc, char** argv) {
    const int ne = 784, na = 256;
    srand(7);
    // ---- accuracy on a small K --------------------------------------------------------------------------------
    {
        const int rows = 4096;
        std::vector<float> E((size_t)rows * ne), A((size_t)rows * na);
        for (auto& x : E) x = (float)(nrand() * 0.3);                                   // errors: signed
        for (auto& x : A) { const double v = nrand() * 3.0; x = (float)(v > 0 ? v : 0); } // activations: relu
        float *dE, *dA, *slab, *slab_b;
        hipMalloc(&dE, E.size() * 4); hipMalloc(&dA, A.size() * 4);
        hipMalloc(&slab, (size_t)ne * na * 4); hipMalloc(&slab_b, ne * 4);
        hipMemcpy(dE, E.data(), E.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        std::vector<double> ref((size_t)64 * na, 0.0), mag((size_t)64 * na, 0.0), refb(64, 0.0);
        for (int r = 0; r < rows; ++r)
            for (int u = 0; u < 64; ++u) {
                const double e = E[(size_t)r * ne + (u * 12 + 5)];
                refb[u] += e;
                for (int i = 0; i < na; ++i) { ref[u * na + i] += e * A[(size_t)r * na + i]; mag[u * na + i] += std::fabs(e * A[(size_t)r * na + i]); }
            }
        for (int which = 0; which < 2; ++which) {
            hipMemset(slab, 0, (size_t)ne * na * 4); hipMemset(slab_b, 0, ne * 4);
            HebArgs a17{dE, dA, slab, slab_b, rows, ne, na, rows, 1, 1, 1, 0}, a16{dE, dA, slab, slab_b, rows, ne, na, rows, 2, 1, 1, 17 * 16};
            if (which == 0) { launch32<17>(a17); launch32<16>(a16); } else { launch6<17>(a17); launch6<16>(a16); }
            hipDeviceSynchronize();
            std::vector<float> h((size_t)ne * na), hb(ne);
            hipMemcpy(h.data(), slab, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), slab_b, ne * 4, hipMemcpyDeviceToHost);
            double mx = 0, rms = 0, mxb = 0;
            for (int u = 0; u < 64; ++u) {
                mxb = std::max(mxb, std::fabs(hb[u * 12 + 5] - refb[u]));
                for (int i = 0; i < na; ++i) {
                    const double e = std::fabs((double)h[(size_t)(u * 12 + 5) * na + i] - ref[u * na + i]) / mag[u * na + i];
                    mx = std::max(mx, e); rms += e * e;
                }
            }
            printf("%-28s K = %d: error / sum|terms|  max %.3e  rms %.3e   (bias sums: max abs err %.3e)   [%s]\n", which ? "bf16x6 (heb6_kernel)" : "fp32 MFMA (shipped kernel)", rows, mx,
                   std::sqrt(rms / (64.0 * na)), mxb, hipGetErrorString(hipGetLastError()));
        }
        hipFree(dE); hipFree(dA); hipFree(slab); hipFree(slab_b);
    }
    // ---- rate at the flush's size: 64 steps x 6016 chains ---------------------------------------------------------
    {
        const int rows = 64 * 6016, ksplit = rows / (48 * 32), rps = ((rows + ksplit - 1) / ksplit + 31) / 32 * 32;
        const int ks = (rows + rps - 1) / rps;
        float *dE, *dA, *slab, *slab_b;
        hipMalloc(&dE, (size_t)rows * ne * 4); hipMalloc(&dA, (size_t)rows * na * 4);
        hipMalloc(&slab, (size_t)ks * ne * na * 4); hipMalloc(&slab_b, (size_t)ks * ne * 4);
        hipMemset(dE, 0x3c, (size_t)rows * ne * 4); hipMemset(dA, 0x3c, (size_t)rows * na * 4);      // 0x3c3c3c3c = 0.0115: plain finite data
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int which = 0; which < 2; ++which)
            for (int rep = 0; rep < 3; ++rep) {
                HebArgs a17{dE, dA, slab, slab_b, rows, ne, na, rps, 1, 1, ks, 0}, a16{dE, dA, slab, slab_b, rows, ne, na, rps, 2, 1, ks, 17 * 16};
                hipEventRecord(e0);
                if (which == 0) { launch32<17>(a17); launch32<16>(a16); } else { launch6<17>(a17); launch6<16>(a16); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flop = 2.0 * rows * ne * na;
                printf("%-28s 784 x 256 over %d rows (%d splits): %.3f ms  = %.1f TFLOP/s of fp32 work   [%s]\n", which ? "bf16x6 (heb6_kernel)" : "fp32 MFMA (shipped kernel)", rows, ks, ms,
                       flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
            }
    }
    return 0;
}
