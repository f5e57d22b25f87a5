// Round-3 feasibility study (NOT used by the product): the step kernel's GEMM inner loop with fp32 emulated on the bf16 pipe.
// Per wave and 32-deep k-block: 4 unit tiles x 2 chain tiles, i.e. 48 v_mfma_f32_16x16x32_bf16 (six products per tile pair) against
//   A: pre-split weights, three bf16 planes in MFMA fragment order: 12 x global_load_dwordx4 (1 KiB each) from an L2-resident stream
//      of the size the real kernel would have (3.3 MB per workgroup and step at cfg-M),
//   B: fp32 activations from LDS rows [chain][k]: 4 x ds_read_b128, split into hi / mid / lo by the GEMM wave itself (VALU),
// fragments requested ONE block ahead (two register sets of 48 VGPRs).  Optionally waves 4-7 (one more per SIMD) run an epilogue-like
// VALU stream.  Prints cycles per k-block (ideal: 48 x 16 = 768) and the aggregate fragment stream.
//   hipcc --offload-arch=gfx950 -O3 scripts/bf16x6_stream_ubench.hip -o scripts/bin/bf16x6_stream_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
struct Planes { u32x4 h, m, l; };
__device__ __forceinline__ Planes split8(f32x4 x0, f32x4 x1) {       // 8 floats -> three bf16x8
    Planes p;
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[2 * j], b = v[2 * j + 1];
        const unsigned h = pk_bf16(a, b);
        const float ra = a - lo_f(h), rb = b - hi_f(h);
        const unsigned m = pk_bf16(ra, rb);
        p.h[j] = h; p.m[j] = m; p.l[j] = pk_bf16(ra - lo_f(m), rb - hi_f(m));
    }
    return p;
}
#define BF(x_) __builtin_bit_cast(bf16x8, x_)

// MODE bit 0: A loads, bit 1: B reads + split, bit 2: E-like VALU waves
template <int MODE>
__global__ __launch_bounds__(512, 2) void kstream(const u32x4* __restrict__ Ag, float* out, unsigned long long* cyc, int nkb, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[32 * 264];
    const gu32x4* A = (const gu32x4*)Ag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 264; i += 512) lds[i] = (i % 97) * 1e-2f - 0.3f;
    __syncthreads();
    if (wave >= 4) {                                                    // epilogue-like VALU stream (transcendental + fma mix)
        if (!(MODE & 4)) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = lane * 1e-3f + i;
        const unsigned long long m0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps * nkb * 6; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = __builtin_fmaf(v[i], 0.999f, 0.01f); v[i] = v[i] > 4.f ? __builtin_amdgcn_exp2f(-v[i]) : v[i] * 1.01f; }
        }
        const unsigned long long m1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) cyc[blockIdx.x * 8 + wave] = m1 - m0;
        return;
    }
    const int c = lane & 15, g = lane >> 4;
    const float* bp = lds + c * 264 + 8 * g;
    f32x4 acc[4][2];
    for (int t = 0; t < 4; ++t) for (int ct = 0; ct < 2; ++ct) acc[t][ct] = {0.f, 0.f, 0.f, 0.f};
    int aoff[4];
    for (int t = 0; t < 4; ++t) aoff[t] = ((wave * 4 + t) * nkb) * 3 * 64;           // [tile][kb][plane][lane]
    u32x4 aP[4][3], aQ[4][3];
    f32x4 bP[2][2], bQ[2][2];
    for (int t = 0; t < 4; ++t) for (int p = 0; p < 3; ++p) { aP[t][p] = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; aQ[t][p] = aP[t][p]; }
    for (int ct = 0; ct < 2; ++ct) for (int h = 0; h < 2; ++h) { bP[ct][h] = {1.f, 0.5f, 0.25f, 2.f}; bQ[ct][h] = bP[ct][h]; }
#define LOADSET(a_, b_, k_) do { \
    if (MODE & 1) { _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int p = 0; p < 3; ++p) a_[t][p] = A[aoff[t] + ((k_) * 3 + p) * 64 + lane]; } \
    if (MODE & 2) { _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) _Pragma("unroll") for (int h = 0; h < 2; ++h) \
        b_[ct][h] = *(const f32x4*)(bp + ct * 16 * 264 + ((k_) & 7) * 32 + 4 * h); } } while (0)
#define BLOCK(a_, b_) do { \
    Planes B0, B1; \
    if (MODE & 2) { B0 = split8(b_[0][0], b_[0][1]); B1 = split8(b_[1][0], b_[1][1]); } \
    else { B0.h = B0.m = B0.l = __builtin_bit_cast(u32x4, b_[0][0]); B1.h = B1.m = B1.l = __builtin_bit_cast(u32x4, b_[1][0]); } \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][1]), BF(B0.m), acc[t][0], 0, 0, 0); \
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][1]), BF(B1.m), acc[t][1], 0, 0, 0); } \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][2]), BF(B0.h), acc[t][0], 0, 0, 0); \
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][2]), BF(B1.h), acc[t][1], 0, 0, 0); } \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][0]), BF(B0.l), acc[t][0], 0, 0, 0); \
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][0]), BF(B1.l), acc[t][1], 0, 0, 0); } \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][1]), BF(B0.h), acc[t][0], 0, 0, 0); \
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][1]), BF(B1.h), acc[t][1], 0, 0, 0); } \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][0]), BF(B0.m), acc[t][0], 0, 0, 0); \
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][0]), BF(B1.m), acc[t][1], 0, 0, 0); } \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][0]), BF(B0.h), acc[t][0], 0, 0, 0); \
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(a_[t][0]), BF(B1.h), acc[t][1], 0, 0, 0); } } while (0)
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        LOADSET(aP, bP, 0);
        int kb = 0;
        for (; kb + 2 <= nkb; kb += 2) {
            __builtin_amdgcn_sched_barrier(0);
            LOADSET(aQ, bQ, kb + 1); BLOCK(aP, bP);
            __builtin_amdgcn_sched_barrier(0);
            LOADSET(aP, bP, (kb + 2 < nkb ? kb + 2 : 0)); BLOCK(aQ, bQ);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0][0];
    for (int t = 0; t < 4; ++t) for (int ct = 0; ct < 2; ++ct) s += acc[t][ct];
    out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = m1 - m0;
}

template <int MODE> void run(const char* name, int nblocks, int nkb) {
    const int reps = 40;
    u32x4* A; float* out; unsigned long long* cyc;
    const size_t nA = (size_t)16 * nkb * 3 * 64;      // 16 tiles x nkb x 3 planes x 64 lanes, shared by all blocks
    hipMalloc(&A, nA * 16); hipMemset(A, 0x3c, nA * 16);
    hipMalloc(&out, (size_t)nblocks * 512 * 4); hipMalloc(&cyc, (size_t)nblocks * 8 * 8); hipMemset(cyc, 0, (size_t)nblocks * 64);
    hipLaunchKernelGGL((kstream<MODE>), dim3(nblocks), dim3(512), 0, 0, A, out, cyc, nkb, 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstream<MODE>), dim3(nblocks), dim3(512), 0, 0, A, out, cyc, nkb, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double g = 0; for (int b = 0; b < nblocks; ++b) for (int w = 0; w < 4; ++w) g += (double)h[b * 8 + w];
    g /= nblocks * 4.0;
    const double blocks = (double)reps * nkb;
    const double bytes = (double)nblocks * nA * 16 * reps;
    printf("%-52s WGs=%3d nkb=%3d  %7.1f ticks per k32-block (ideal 768 cycles)  %8.3f ms  -> %.2f us per block, fragment stream %.1f TB/s, fp32-equivalent %.0f TFLOP/s\n",
           name, nblocks, nkb, g / blocks, ms, ms * 1e3 / blocks, (MODE & 1) ? bytes / ms / 1e9 : 0.0, nblocks * 4.0 * blocks * 8 * 16 * 16 * 32 * 2 / ms / 1e9);
    hipFree(A); hipFree(out); hipFree(cyc);
}

int main() {
    for (int nb : {188, 256}) {
        run<0>("MFMA only (register operands)", nb, 70);
        run<1>("MFMA + 3-plane fragment loads", nb, 70);
        run<2>("MFMA + LDS fp32 B + split in the GEMM wave", nb, 70);
        run<3>("MFMA + fragment loads + B split", nb, 70);
        run<7>("... + an epilogue-like VALU wave per SIMD", nb, 70);
    }
    return 0;
}
