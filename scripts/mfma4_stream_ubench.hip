// Developer micro-benchmark 3: GEMM inner loop for 24-chain workgroups built on v_mfma_f32_4x4x1_16B_f32.
//   out[unit 64 x NT tiles][chain 24] += W[unit][k] * act[chain][k]
//   B operand (per lane = unit): weights, one global_load_dwordx4 per (tile, 4 k): lane holds W[u0+lane][k..k+3]
//   A operand: act[4 chains][16 k] in ONE VGPR per chain group (lane = 4*kk + chain), broadcast to all 16 blocks with
//   cbsz=4, abid=kk -> one ds_read_b32 per chain group per 16 k-steps.
// Per 16 k-steps and wave: 4*NT weight loads, 6 LDS reads, 96*NT MFMAs (8 cycles each).
// hipcc --offload-arch=gfx950 -O3 scripts/mfma4_stream_ubench.hip -o scripts/bin/mfma4_stream_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;

constexpr int kCG = 6;        // chain groups of 4 -> 24 chains
constexpr int kLd = 260;      // LDS row stride (floats) of act[chain][k], K = 256

#define MFMA4(a_, b_, c_, abid_) __builtin_amdgcn_mfma_f32_4x4x1f32(a_, b_, c_, 4, abid_, 0)

template <int NT, bool GLOAD, bool LDSRD>
__global__ __launch_bounds__(256) void kstream(const f32x4* __restrict__ Wg, float* out, unsigned long long* cyc, int nkb, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[24 * kLd];
    const gf32x4* W = (const gf32x4*)Wg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 24 * kLd; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    // A operand address: lane = 4*kk + r reads act[chain 4g + r][16 kb + kk]
    const float* ap = lds + (lane & 3) * kLd + (lane >> 2);
    f32x4 acc[NT][kCG];
    for (int t = 0; t < NT; ++t) for (int g = 0; g < kCG; ++g) acc[t][g] = {0.f, 0.f, 0.f, 0.f};
    // packed weights: [tile][k/4][lane] f32x4; this wave's tiles: wave*NT + t
    int woff[NT];
    for (int t = 0; t < NT; ++t) woff[t] = ((wave * NT + t) * nkb * 4) * 64 + lane;
    f32x4 wP[NT][4], wQ[NT][4], wR[NT][4];
    float aP[kCG], aQ[kCG], aR[kCG];
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 4; ++j) { wP[t][j] = {1.f, 2.f, 3.f, 4.f}; wQ[t][j] = wP[t][j]; wR[t][j] = wP[t][j]; }
    for (int g = 0; g < kCG; ++g) { aP[g] = 1.f; aQ[g] = 1.f; aR[g] = 1.f; }
#define LOADSET(w_, a_, kb_) do { \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int j = 0; j < 4; ++j) { if (GLOAD) w_[t][j] = W[woff[t] + ((kb_) * 4 + j) * 64]; } \
    _Pragma("unroll") for (int g = 0; g < kCG; ++g) { if (LDSRD) a_[g] = ap[g * 4 * kLd + (kb_) * 16]; } } while (0)
    // 16 k-steps: k = 4 j + r  (weight register j component r), abid = k
#define KSTEP(w_, a_, j_, r_, comp_) \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int g = 0; g < kCG; ++g) acc[t][g] = MFMA4(a_[g], w_[t][j_].comp_, acc[t][g], 4 * j_ + r_);
#define BLOCK(w_, a_) do { \
    KSTEP(w_, a_, 0, 0, x) KSTEP(w_, a_, 0, 1, y) KSTEP(w_, a_, 0, 2, z) KSTEP(w_, a_, 0, 3, w) \
    KSTEP(w_, a_, 1, 0, x) KSTEP(w_, a_, 1, 1, y) KSTEP(w_, a_, 1, 2, z) KSTEP(w_, a_, 1, 3, w) \
    KSTEP(w_, a_, 2, 0, x) KSTEP(w_, a_, 2, 1, y) KSTEP(w_, a_, 2, 2, z) KSTEP(w_, a_, 2, 3, w) \
    KSTEP(w_, a_, 3, 0, x) KSTEP(w_, a_, 3, 1, y) KSTEP(w_, a_, 3, 2, z) KSTEP(w_, a_, 3, 3, w) } while (0)
    // spread: one VMEM read behind every 6*NT... MFMAs (4*NT loads per 96*NT MFMAs), LDS reads likewise
#define STAGE_SCHED() do { \
    _Pragma("unroll") for (int i = 0; i < 4 * NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } \
    _Pragma("unroll") for (int g = 0; g < kCG; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 6 * NT, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
    __builtin_amdgcn_sched_group_barrier(0x008, 96 * NT - 48 * NT - 36 * NT, 0); } while (0)
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        LOADSET(wP, aP, 0); LOADSET(wQ, aQ, 1);
        int kb = 0;
        for (; kb + 5 <= nkb; kb += 3) {
            __builtin_amdgcn_sched_barrier(0);
            LOADSET(wR, aR, kb + 2); BLOCK(wP, aP); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
            LOADSET(wP, aP, kb + 3); BLOCK(wQ, aQ); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
            LOADSET(wQ, aQ, kb + 4); BLOCK(wR, aR); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
        }
        const int rem = nkb - kb;
        if (rem == 4) { LOADSET(wR, aR, kb + 2); BLOCK(wP, aP); LOADSET(wP, aP, kb + 3); BLOCK(wQ, aQ); BLOCK(wR, aR); BLOCK(wP, aP); }
        else if (rem == 3) { LOADSET(wR, aR, kb + 2); BLOCK(wP, aP); BLOCK(wQ, aQ); BLOCK(wR, aR); }
        else if (rem == 2) { BLOCK(wP, aP); BLOCK(wQ, aQ); }
        else if (rem == 1) { BLOCK(wP, aP); }
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0][0];
    for (int t = 0; t < NT; ++t) for (int g = 0; g < kCG; ++g) s += acc[t][g];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = m1 - m0;
}

template <int NT, bool G, bool Ld> void run(const char* name, int nblocks, int nkb) {
    const int reps = 200 * 50 / nkb;
    f32x4* W; float* out; unsigned long long* cyc;
    const size_t nW = (size_t)4 * NT * nkb * 4 * 64;     // 4 waves x NT tiles x nkb x 4 loads x 64 lanes (shared by all blocks: L2-resident)
    hipMalloc(&W, nW * 16); hipMemset(W, 0, nW * 16);
    hipMalloc(&out, (size_t)nblocks * 256 * 4); hipMalloc(&cyc, (size_t)nblocks * 4 * 8);
    hipLaunchKernelGGL((kstream<NT, G, Ld>), dim3(nblocks), dim3(256), 0, 0, W, out, cyc, nkb, 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstream<NT, G, Ld>), dim3(nblocks), dim3(256), 0, 0, W, out, cyc, nkb, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double nm = (double)reps * nkb * 96 * NT;
    const double macs = nm * 256.0 * 4 * nblocks;        // per MFMA 256 MACs, 4 waves
    printf("%-40s NT=%d blocks=%3d nkb=%2d  %6.2f ticks/MFMA (ideal 8)  %8.3f ms  %6.1f TFLOP/s\n", name, NT, nblocks, nkb, mean / nm, ms, 2 * macs / (ms * 1e-3) / 1e12);
    hipFree(W); hipFree(out); hipFree(cyc);
}

int main() {
    for (int nkb : {50, 16}) {
        run<1, false, false>("4x4x1 MFMA only", 250, nkb);
        run<1, true, true>("4x4x1 + weight stream + LDS operand", 250, nkb);
        run<2, false, false>("4x4x1 MFMA only", 250, nkb);
        run<2, true, true>("4x4x1 + weight stream + LDS operand", 250, nkb);
        run<2, true, false>("4x4x1 + weight stream", 250, nkb);
        run<2, false, true>("4x4x1 + LDS operand", 250, nkb);
    }
    return 0;
}
