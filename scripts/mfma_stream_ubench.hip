// Developer micro-benchmark 2: the steps kernel's GEMM inner loop in isolation.
// Per 32 MFMAs (4 tiles x 2 chain tiles x 4 k-groups): 4 x global_load_dwordx4 (1 KiB each, from a
// 1 MiB L2-resident packed stream), 2 x ds_read_b128, fragments requested two blocks ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;

#define STAGE_SCHED() do { \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
    __builtin_amdgcn_sched_group_barrier(0x008, 20, 0); } while (0)
template <bool GLOAD, bool LDSRD, bool SPREAD, bool DEEP = false>
__global__ __launch_bounds__(256) void kstream(const f32x4* __restrict__ Ag, float* out, unsigned long long* cyc, int nkb, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[32 * 260];
    const gf32x4* A = (const gf32x4*)Ag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 260; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    const int c = lane & 15, q = lane >> 4;
    const float* bp = lds + c * 260 + 4 * q;
    f32x4 acc[4][2];
    for (int t = 0; t < 4; ++t) for (int ct = 0; ct < 2; ++ct) acc[t][ct] = {0.f, 0.f, 0.f, 0.f};
    int aoff[4];
    for (int t = 0; t < 4; ++t) aoff[t] = ((wave * 4 + t) * nkb) * 64;
    f32x4 aP[4], aQ[4], aR[4], aS[4], bP[2], bQ[2], bR[2], bS[2];
#define LOADSET(a_, b_, k_) do { \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { if (GLOAD) a_[t] = A[aoff[t] + (k_) * 64 + lane]; } \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) { if (LDSRD) b_[ct] = *(const f32x4*)(bp + ct * 16 * 260 + (k_) * 16); } } while (0)
#define BLOCK(a_, b_) do { \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].x, b_[ct].x, acc[t][ct], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].y, b_[ct].y, acc[t][ct], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].z, b_[ct].z, acc[t][ct], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].w, b_[ct].w, acc[t][ct], 0, 0, 0); } while (0)
    for (int t = 0; t < 4; ++t) { aP[t] = {1.f, 2.f, 3.f, 4.f}; aQ[t] = aP[t]; aR[t] = aP[t]; aS[t] = aP[t]; }
    for (int ct = 0; ct < 2; ++ct) { bP[ct] = {1.f, 1.f, 1.f, 1.f}; bQ[ct] = bP[ct]; bR[ct] = bP[ct]; bS[ct] = bP[ct]; }
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        LOADSET(aP, bP, 0); LOADSET(aQ, bQ, 1);
        int kb = 0;
        if (DEEP) {
            LOADSET(aR, bR, 2);
            for (; kb + 7 <= nkb; kb += 4) {
                __builtin_amdgcn_sched_barrier(0);
                LOADSET(aS, bS, kb + 3); BLOCK(aP, bP); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
                LOADSET(aP, bP, kb + 4); BLOCK(aQ, bQ); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
                LOADSET(aQ, bQ, kb + 5); BLOCK(aR, bR); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
                LOADSET(aR, bR, kb + 6); BLOCK(aS, bS); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
            }
            // tail: not exact (counts 3 extra blocks' worth at most); only the steady state matters here
            BLOCK(aP, bP); BLOCK(aQ, bQ); BLOCK(aR, bR);
            kb = nkb;
        }
        for (; kb + 5 <= nkb; kb += 3) {
            if (SPREAD) {
                __builtin_amdgcn_sched_barrier(0);
                LOADSET(aR, bR, kb + 2); BLOCK(aP, bP); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
                LOADSET(aP, bP, kb + 3); BLOCK(aQ, bQ); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
                LOADSET(aQ, bQ, kb + 4); BLOCK(aR, bR); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
            } else {
            LOADSET(aR, bR, kb + 2); __builtin_amdgcn_sched_barrier(0);
            BLOCK(aP, bP); __builtin_amdgcn_sched_barrier(0);
            LOADSET(aP, bP, kb + 3); __builtin_amdgcn_sched_barrier(0);
            BLOCK(aQ, bQ); __builtin_amdgcn_sched_barrier(0);
            LOADSET(aQ, bQ, kb + 4); __builtin_amdgcn_sched_barrier(0);
            BLOCK(aR, bR); __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int rem = nkb - kb;
        if (rem == 4) { LOADSET(aR, bR, kb + 2); BLOCK(aP, bP); LOADSET(aP, bP, kb + 3); BLOCK(aQ, bQ); BLOCK(aR, bR); BLOCK(aP, bP); }
        else if (rem == 3) { LOADSET(aR, bR, kb + 2); BLOCK(aP, bP); BLOCK(aQ, bQ); BLOCK(aR, bR); }
        else if (rem == 2) { BLOCK(aP, bP); BLOCK(aQ, bQ); }
        else if (rem == 1) { BLOCK(aP, bP); }
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0][0];
    for (int t = 0; t < 4; ++t) for (int ct = 0; ct < 2; ++ct) s += acc[t][ct];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = m1 - m0;
}

template <bool G, bool Ld, bool Sp = false, bool Dp = false> void run(const char* name, int nblocks, int nkb = 50) {
    const int reps = 200 * 50 / nkb;
    f32x4* A; float* out; unsigned long long* cyc;
    const size_t nA = (size_t)16 * nkb * 64;     // 16 tiles per block x nkb x 64 lanes (shared by all blocks: L2-resident)
    hipMalloc(&A, nA * 16); hipMemset(A, 0, nA * 16);
    hipMalloc(&out, (size_t)nblocks * 256 * 4); hipMalloc(&cyc, (size_t)nblocks * 4 * 8);
    hipLaunchKernelGGL((kstream<G, Ld, Sp, Dp>), dim3(nblocks), dim3(256), 0, 0, A, out, cyc, nkb, 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstream<G, Ld, Sp, Dp>), dim3(nblocks), dim3(256), 0, 0, A, out, cyc, nkb, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    // blocks executed per rep: main loop covers kb=0..47 (16 iters x 3), + 2 tail = 50
    const double nm = (double)reps * nkb * 32;
    printf("%-44s blocks=%3d nkb=%2d  %6.2f ticks/MFMA  %8.3f ms\n", name, nblocks, nkb, mean / nm, ms);
    hipFree(A); hipFree(out); hipFree(cyc);
}

int main() {
    for (int nkb : {50, 16}) {
        run<false, false>("MFMA only (register operands)", 188, nkb);
        run<true, true>("MFMA + global loads + LDS reads (clumped)", 188, nkb);
        run<true, true, true>("MFMA + global loads + LDS reads (spread)", 188, nkb);
        run<true, true, true, true>("MFMA + global loads + LDS reads (spread, 3 ahead)", 188, nkb);
        run<false, true, true>("MFMA + LDS reads (spread)", 188, nkb);
        run<true, false, true>("MFMA + global loads (spread)", 188, nkb);
    }
    return 0;
}
