// Developer micro-benchmark: does plain VALU work of one wave overlap with the fp32 MFMAs of the other wave of a SIMD?
// 512-thread workgroups, one per CU: waves 0-3 issue v_mfma_f32_16x16x4_f32 back to back, waves 4-7 issue independent
// v_fma_f32 (or v_exp_f32).  Each role is timed alone and together (s_memtime per wave).
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o scripts/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VKIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int mfma_iters, int valu_iters, int prio) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    unsigned long long m0 = 0, m1 = 0;
    float res = 0.f;
    if (wave < 4) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
        m0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        m1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) res += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        if (prio) __builtin_amdgcn_s_setprio(2);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        m0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (VKIND == 0) v[i] = __builtin_fmaf(v[i], b, a);
                    else if (VKIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
                    else v[i] = __uint_as_float((__float_as_uint(v[i]) ^ 0x9e3779b9u) + 0x7f4a7c15u);   // integer ops
                }
        }
        m1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) res += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = m1 - m0;
}

template <int VKIND>
void run(const char* name, float* out, unsigned long long* cyc, int mi, int vi, int prio) {
    hipMemset(cyc, 0, 256 * 8 * 8);
    hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, out, cyc, mi, vi, prio);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double g = 0, e = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? g : e) += (double)h[b * 8 + w];
    g /= 1024; e /= 1024;
    printf("%-34s mfma_iters=%5d valu_iters=%5d prio=%d : G %9.0f ticks (%.2f per MFMA)   E %9.0f ticks (%.2f per VALU op)\n", name, mi, vi, prio, g,
           mi ? g / (mi * 32.0) : 0.0, e, vi ? e / (vi * 32.0) : 0.0);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("mfma alone", out, cyc, 2000, 0, 0);
        run<0>("v_fma alone", out, cyc, 0, 8000, 0);
        run<0>("mfma + v_fma", out, cyc, 2000, 8000, 0);
        run<0>("mfma + v_fma (E prio)", out, cyc, 2000, 8000, 1);
        run<1>("v_exp alone", out, cyc, 0, 2000, 0);
        run<1>("mfma + v_exp", out, cyc, 2000, 2000, 0);
        run<2>("int alone", out, cyc, 0, 8000, 0);
        run<2>("mfma + int", out, cyc, 2000, 8000, 0);
    }
    printf("(s_memtime ticks at 100 MHz: 1 tick = ~24 shader cycles at 2.4 GHz)\n");
    return 0;
}
