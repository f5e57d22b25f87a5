// Developer micro-benchmark: cadence of v_mfma_f32_16x16x4_f32 / 32x32x2 in the shapes the steps kernel uses.
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_ubench.hip -o /tmp/mfma_ubench && /tmp/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int DIST>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = m1 - m0;
    (void)t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = m1 - m0;
}


// 4x4x1 (16 blocks): the fine-grained shape -- 4 chains x 64 units per instruction, A broadcast from block `abid`
template <int NACC, int BCAST>
__global__ __launch_bounds__(256) void k4(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define K4_ROUND(ABID) _Pragma("unroll") for (int i = 0; i < NACC; ++i) { \
            if (BCAST) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, ABID, 0); \
            else acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0); }
#pragma unroll
        for (int r = 0; r < 96 / NACC / 4; ++r) { K4_ROUND(0) K4_ROUND(5) K4_ROUND(10) K4_ROUND(15) }
        a += 1e-6f;
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = m1 - m0;
}

template <typename F> void run(const char* name, F launch, int per_iter, double flop_per_mfma) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 2000;
    launch(out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    double mf = (double)iters * per_iter;
    printf("%-34s %7.2f memtime-ticks/MFMA   %8.3f ms   %7.1f TFLOP/s (1024 waves)\n", name, mean / mf, ms,
           1024.0 * mf * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    run("16x16x4 f32, 8 acc round-robin", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k16<8, 0>), dim3(256), dim3(256), 0, 0, o, c, n); }, 32, 2048.0);
    run("16x16x4 f32, 4 acc round-robin", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k16<4, 0>), dim3(256), dim3(256), 0, 0, o, c, n); }, 32, 2048.0);
    run("16x16x4 f32, 2 acc alternating", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k16<2, 0>), dim3(256), dim3(256), 0, 0, o, c, n); }, 32, 2048.0);
    run("16x16x4 f32, 1 acc chain", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k16<1, 0>), dim3(256), dim3(256), 0, 0, o, c, n); }, 32, 2048.0);
    run("32x32x2 f32, 4 acc round-robin", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k32<4>), dim3(256), dim3(256), 0, 0, o, c, n); }, 16, 4096.0);
    run("32x32x2 f32, 2 acc alternating", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k32<2>), dim3(256), dim3(256), 0, 0, o, c, n); }, 16, 4096.0);
    run("32x32x2 f32, 1 acc chain", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k32<1>), dim3(256), dim3(256), 0, 0, o, c, n); }, 16, 4096.0);
    run("4x4x1 16B f32, 6 acc, no bcast", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k4<6, 0>), dim3(256), dim3(256), 0, 0, o, c, n); }, 96, 512.0);
    run("4x4x1 16B f32, 6 acc, cbsz=4 abid", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k4<6, 1>), dim3(256), dim3(256), 0, 0, o, c, n); }, 96, 512.0);
    run("4x4x1 16B f32, 3 acc, cbsz=4 abid", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k4<3, 1>), dim3(256), dim3(256), 0, 0, o, c, n); }, 96, 512.0);
    run("4x4x1 16B f32, 12 acc, cbsz=4 abid", [](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL((k4<12, 1>), dim3(256), dim3(256), 0, 0, o, c, n); }, 96, 512.0);
    return 0;
}
