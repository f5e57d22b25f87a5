// Developer micro-benchmark: does the ADDRESS PATTERN of the weight-fragment loads matter to a wave that streams MFMAs?
// The step kernel's GEMM loop (32 MFMAs : 4 global_load_dwordx4 : 2 ds_read_b128, fragments two k-blocks ahead) with
//   MAP 0  one contiguous 1 KiB per wave-instruction (8 full 128-B lines; every line is touched by two lane quads)
//   MAP 1  16 lines per instruction, 64 B of each (k-blocks 2p and 2p+1 interleaved inside every line)
//   MAP 2  32 lines per instruction, 32 B of each (four k-blocks interleaved)
// and load flavours plain / nontemporal.  The PMC counters of the step kernel show the vector L1 stalled on pending
// data for ~45 % of its cycles; this asks whether hits on pending lines are what it waits for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;

#define STAGE_SCHED() do { \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
    __builtin_amdgcn_sched_group_barrier(0x008, 20, 0); } while (0)

template <int MAP> __device__ __forceinline__ int lane_off(int lane) {     // f32x4 units inside a tile's stream
    if (MAP == 0) return lane;
    if (MAP == 1) return (lane >> 2) * 8 + (lane & 3);                     // 128-B line per quad, first 64 B
    return (lane >> 1) * 8 + (lane & 1);                                   // 128-B line per lane pair, first 32 B
}
template <int MAP> __device__ __forceinline__ int k_off(int k) {           // f32x4 units
    if (MAP == 0) return k * 64;
    if (MAP == 1) return (k >> 1) * 128 + (k & 1) * 4;
    return (k >> 2) * 256 + (k & 3) * 2;
}

template <int MAP, bool NT>
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
    const int lo = lane_off<MAP>(lane);
    for (int t = 0; t < 4; ++t) aoff[t] = ((wave * 4 + t) * (nkb + 4)) * 64 + lo;
    f32x4 aP[4], aQ[4], aR[4], bP[2], bQ[2], bR[2];
#define LD(p_) (NT ? __builtin_nontemporal_load(p_) : *(p_))
#define LOADSET(a_, b_, k_) do { \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) a_[t] = LD(A + aoff[t] + k_off<MAP>(k_)); \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) b_[ct] = *(const f32x4*)(bp + ct * 16 * 260 + (k_) * 16); } while (0)
#define BLOCK(a_, b_) do { \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].x, b_[ct].x, acc[t][ct], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].y, b_[ct].y, acc[t][ct], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].z, b_[ct].z, acc[t][ct], 0, 0, 0); \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[t].w, b_[ct].w, acc[t][ct], 0, 0, 0); } while (0)
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        LOADSET(aP, bP, 0); LOADSET(aQ, bQ, 1);
        int kb = 0;
        for (; kb + 5 <= nkb; kb += 3) {
            __builtin_amdgcn_sched_barrier(0);
            LOADSET(aR, bR, kb + 2); BLOCK(aP, bP); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
            LOADSET(aP, bP, kb + 3); BLOCK(aQ, bQ); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
            LOADSET(aQ, bQ, kb + 4); BLOCK(aR, bR); STAGE_SCHED(); __builtin_amdgcn_sched_barrier(0);
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

template <int MAP, bool NT> void run(const char* name, int nblocks, int nkb) {
    const int reps = 400 * 50 / nkb;
    f32x4* A; float* out; unsigned long long* cyc;
    const size_t nA = (size_t)16 * (nkb + 4) * 64 + 4096;  // 16 tile streams per block (shared by all blocks: L2-resident, far beyond a 32 KiB L1)
    hipMalloc(&A, nA * 16); hipMemset(A, 0, nA * 16);
    hipMalloc(&out, (size_t)nblocks * 256 * 4); hipMalloc(&cyc, (size_t)nblocks * 4 * 8);
    hipLaunchKernelGGL((kstream<MAP, NT>), dim3(nblocks), dim3(256), 0, 0, A, out, cyc, nkb, 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstream<MAP, NT>), dim3(nblocks), dim3(256), 0, 0, A, out, cyc, nkb, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nblocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double nm = (double)reps * nkb * 32;
    printf("%-52s blocks=%3d nkb=%3d  %6.2f ticks/MFMA  %8.3f ms\n", name, nblocks, nkb, mean / nm, ms);
    hipFree(A); hipFree(out); hipFree(cyc);
}

int main() {
    for (int nkb : {16, 64, 200}) {          // 16 x nkb KiB per block: 256 KiB, 1 MiB, 3.2 MiB (the kernel streams 2.2 MB per step)
        for (int nb : {188, 256}) {
            run<0, false>("contiguous 1 KiB per instruction", nb, nkb);
            run<0, true>("contiguous 1 KiB, nontemporal", nb, nkb);
            run<1, false>("16 lines x 64 B per instruction", nb, nkb);
            run<1, true>("16 lines x 64 B, nontemporal", nb, nkb);
            run<2, false>("32 lines x 32 B per instruction", nb, nkb);
        }
    }
    return 0;
}
