// Developer micro-benchmark (round 3): does VALU work of one wave of a SIMD progress beside the OTHER wave's bf16 MFMAs
// (v_mfma_f32_16x16x32_bf16), the way it does NOT beside fp32 MFMAs (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in the step kernel)?
// 512-thread workgroups, one per CU: waves 0-3 ("G", one per SIMD) issue MFMAs back to back, waves 4-7 ("E") issue VALU
// work.  Each role is timed alone and together (wall_clock64 ticks per wave, 100 MHz) and converted with the event time.
//   MKIND 0: v_mfma_f32_16x16x4_f32 (what ships), 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_32x32x16_bf16
//   VKIND 0: independent v_fma_f32, 1: v_exp_f32, 2: integer xor/add, 3: v_mad_u64_u32 (Philox's multiply)
// hipcc --offload-arch=gfx950 -O3 scripts/bf16_overlap_ubench.hip -o scripts/bin/bf16_overlap_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MKIND, int VKIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int mfma_iters, int valu_iters, int prio) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    unsigned long long m0 = 0, m1 = 0;
    float res = 0.f;
    if (wave < 4) {
        if (prio == 2) __builtin_amdgcn_s_setprio(2);
        bf16x8 ab, bb;
        for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(a + j); bb[j] = (__bf16)(b + j); }
        if constexpr (MKIND == 2) {
            f32x16 acc[2];
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
            m0 = wall_clock64();
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
            }
            m1 = wall_clock64();
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) res += acc[i][j];
        } else {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
            m0 = wall_clock64();
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if constexpr (MKIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
                    }
            }
            m1 = wall_clock64();
            for (int i = 0; i < 8; ++i) res += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        }
    } else {
        if (prio == 1) __builtin_amdgcn_s_setprio(2);
        float v[8];
        unsigned long long w[8];
        for (int i = 0; i < 8; ++i) { v[i] = a + i; w[i] = threadIdx.x + i; }
        m0 = wall_clock64();
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (VKIND == 0) v[i] = __builtin_fmaf(v[i], b, a);
                    else if (VKIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
                    else if (VKIND == 2) v[i] = __uint_as_float((__float_as_uint(v[i]) ^ 0x9e3779b9u) + 0x7f4a7c15u);
                    else w[i] = (unsigned long long)(unsigned)w[i] * 0xD2511F53u + (w[i] >> 32);
                }
        }
        m1 = wall_clock64();
        for (int i = 0; i < 8; ++i) res += v[i] + (float)w[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = m1 - m0;
}

template <int MKIND, int VKIND>
void run(const char* name, float* out, unsigned long long* cyc, int mi, int vi, int prio) {
    hipMemset(cyc, 0, 256 * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MKIND, VKIND>), dim3(256), dim3(512), 0, 0, out, cyc, mi, vi, prio);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double g = 0, e = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? g : e) += (double)h[b * 8 + w];
    g /= 1024; e /= 1024;     // 100 MHz ticks = 10 ns
    printf("%-44s mi=%5d vi=%5d prio=%d : G %8.1f us (%6.2f ns per MFMA)   E %8.1f us (%6.3f ns per VALU op)   kernel %.1f us\n", name, mi, vi, prio,
           g * 0.01, mi ? g * 10.0 / (mi * 32.0) : 0.0, e * 0.01, vi ? e * 10.0 / (vi * 32.0) : 0.0, ms * 1e3);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        printf("--- rep %d\n", rep);
        run<0, 0>("fp32 mfma alone", out, cyc, 4000, 0, 0);
        run<1, 0>("bf16 16x16x32 alone", out, cyc, 8000, 0, 0);
        run<2, 0>("bf16 32x32x16 alone", out, cyc, 4000, 0, 0);
        run<0, 0>("v_fma alone", out, cyc, 0, 16000, 0);
        run<0, 1>("v_exp alone", out, cyc, 0, 4000, 0);
        run<0, 2>("int alone", out, cyc, 0, 16000, 0);
        run<0, 3>("mad_u64 alone", out, cyc, 0, 8000, 0);
        run<0, 0>("fp32 mfma + v_fma", out, cyc, 4000, 16000, 0);
        run<1, 0>("bf16 16x16x32 + v_fma", out, cyc, 8000, 16000, 0);
        run<1, 0>("bf16 16x16x32 + v_fma (E prio)", out, cyc, 8000, 16000, 1);
        run<1, 0>("bf16 16x16x32 + v_fma (G prio)", out, cyc, 8000, 16000, 2);
        run<2, 0>("bf16 32x32x16 + v_fma", out, cyc, 4000, 16000, 0);
        run<1, 1>("bf16 16x16x32 + v_exp", out, cyc, 8000, 4000, 0);
        run<1, 2>("bf16 16x16x32 + int", out, cyc, 8000, 16000, 0);
        run<1, 3>("bf16 16x16x32 + mad_u64", out, cyc, 8000, 8000, 0);
        run<0, 3>("fp32 mfma + mad_u64", out, cyc, 4000, 8000, 0);
    }
    return 0;
}
