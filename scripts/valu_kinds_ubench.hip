// Developer micro-benchmark: which kinds of VALU code make progress beside a wave that streams fp32 MFMAs back to back
// on the same SIMD?  512-thread workgroups (one per CU): waves 0-3 issue v_mfma_f32_16x16x4_f32 for the whole run, waves
// 4-7 run `iters` rounds of one op kind with ILP independent chains per lane.  Reported: E cycles per op, alone and beside
// the MFMA stream, at equal priority and with the E waves raised (s_setprio 2).
// hipcc --offload-arch=gfx950 -O3 scripts/valu_kinds_ubench.hip -o scripts/bin/valu_kinds_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int ILP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int mfma_iters, int iters, int prio) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    unsigned long long m0 = 0, m1 = 0;
    float res = 0.f;
    if (wave < 4) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) res += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        if (prio) __builtin_amdgcn_s_setprio(2);
        float v[ILP];
        for (int i = 0; i < ILP; ++i) v[i] = a + 0.37f * i;
        m0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 32 / ILP; ++r)
#pragma unroll
                for (int i = 0; i < ILP; ++i) {
                    if (KIND == 0) v[i] = __builtin_fmaf(v[i], b, a);                               // v_fma
                    else if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);                        // transcendental
                    else if (KIND == 2) v[i] = v[i] > b ? v[i] - 1.0f : v[i] + a;                   // v_cmp + v_cndmask (VCC/SGPR round trip)
                    else if (KIND == 3) v[i] = fmaxf(v[i], 0.0f) - fabsf(v[i]) * b;                 // modifiers, v_max
                    else if (KIND == 4) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-fabsf(v[i])));   // sigmoid core
                    else if (KIND == 5) v[i] = __builtin_amdgcn_logf(1.0f + v[i] * v[i]);           // v_log
                }
        }
        m1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < ILP; ++i) res += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = m1 - m0;
}

template <int KIND, int ILP>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 500;
    double r[3];
    int cfg[3][2] = {{0, 0}, {40000, 0}, {40000, 1}};
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL((k<KIND, ILP>), dim3(256), dim3(512), 0, 0, out, cyc, cfg[c][0], iters, cfg[c][1]);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double e = 0;
        for (int b = 0; b < 256; ++b) for (int w = 4; w < 8; ++w) e += (double)h[b * 8 + w];
        r[c] = e / 1024 / (iters * 32.0);
    }
    printf("%-28s ILP %2d : alone %6.2f   beside MFMA %7.2f   beside MFMA, E prio %7.2f   (ticks per op)\n", name, ILP, r[0], r[1], r[2]);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 1>("v_fma"); run<0, 4>("v_fma"); run<0, 8>("v_fma");
    run<1, 1>("v_exp"); run<1, 4>("v_exp");
    run<2, 1>("cmp+cndmask+add"); run<2, 4>("cmp+cndmask+add"); run<2, 8>("cmp+cndmask+add");
    run<3, 1>("max/abs/fma"); run<3, 4>("max/abs/fma");
    run<4, 1>("exp+add+rcp"); run<4, 4>("exp+add+rcp"); run<4, 8>("exp+add+rcp");
    run<5, 1>("fma+log"); run<5, 4>("fma+log");
    return 0;
}
