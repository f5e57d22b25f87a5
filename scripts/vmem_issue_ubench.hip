// Developer micro-benchmark: what does one vector-memory instruction cost a wave that streams fp32 MFMAs?
// One wave per SIMD, 32 MFMAs (v_mfma_f32_16x16x4_f32) per block, NLD loads per block issued two blocks ahead and spread
// between the MFMAs; the load form varies.  All loads hit L2/L1 (a 64 KiB window).  Reported: cycles per MFMA.
// hipcc --offload-arch=gfx950 -O3 scripts/vmem_issue_ubench.hip -o scripts/bin/vmem_issue_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;
typedef __attribute__((address_space(1))) f32x2 gf32x2;
typedef __attribute__((address_space(1))) float gf32;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// FORM 0: no loads. 1: global_load_dwordx4 (64-bit vaddr). 2: raw buffer load b128 (SGPR rsrc + 32-bit voffset).
//      3: global dwordx2 x2. 4: global dword x4. 5: ds_read_b128 x4 (LDS). 6: global_load_dwordx4, saddr + 32-bit voffset (asm)
template <int FORM>
__global__ __launch_bounds__(256) void k(const float* __restrict__ Ag, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    f32x4 a[3][4];
    for (int s = 0; s < 3; ++s) for (int t = 0; t < 4; ++t) a[s][t] = {1.f, 2.f, 3.f, 4.f};
    const float b = 1.0f + lane;
    const gf32x4* A4 = (const gf32x4*)Ag;
    const char __attribute__((address_space(1)))* Ab = (const char __attribute__((address_space(1)))*)Ag;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, 1 << 20, 0x27000);
    const uint32_t voff = (uint32_t)(wave * 4096 + lane * 16);
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
#define LOADS(dst, it) do { _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
        const uint32_t o = voff + (uint32_t)(((it) & 3) * 16384 + t * 1024); \
        if (FORM == 1) dst[t] = A4[o / 16]; \
        else if (FORM == 2) { i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0); dst[t] = __builtin_bit_cast(f32x4, v); } \
        else if (FORM == 3) { f32x2 lo = *(const gf32x2*)(Ab + o), hi = *(const gf32x2*)(Ab + o + 32768); dst[t] = {lo.x, lo.y, hi.x, hi.y}; } \
        else if (FORM == 4) { dst[t] = {*(const gf32*)(Ab + o), *(const gf32*)(Ab + o + 32768), *(const gf32*)(Ab + o + 65536), *(const gf32*)(Ab + o + 98304)}; } \
        else if (FORM == 5) dst[t] = *(const f32x4*)(lds + ((o / 4) & 4095)); \
        else if (FORM == 6) { const gf32x4* p = (const gf32x4*)(Ab + (size_t)(((it) & 3) * 16384 + t * 1024)); dst[t] = *(const gf32x4*)((const char __attribute__((address_space(1)))*)p + voff); } \
    } } while (0)
#define BLOCK(src) do { _Pragma("unroll") for (int r = 0; r < 4; ++r) { const float av[4] = {src[r].x, src[r].y, src[r].z, src[r].w}; \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 3], b, acc[i], 0, 0, 0); } } while (0)
#define SCHED() do { if (FORM != 0) { _Pragma("unroll") for (int t = 0; t < 4; ++t) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); \
        __builtin_amdgcn_sched_group_barrier(FORM == 5 ? 0x100 : 0x020, FORM == 3 ? 2 : FORM == 4 ? 4 : 1, 0); } \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0); } } while (0)
    for (int it = 0; it < iters; it += 3) {
        __builtin_amdgcn_sched_barrier(0);
        LOADS(a[2], it); BLOCK(a[0]); SCHED(); __builtin_amdgcn_sched_barrier(0);
        LOADS(a[0], it + 1); BLOCK(a[1]); SCHED(); __builtin_amdgcn_sched_barrier(0);
        LOADS(a[1], it + 2); BLOCK(a[2]); SCHED(); __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = m1 - m0;
}

template <int FORM> void run(const char* name) {
    float *A, *out; unsigned long long* cyc;
    hipMalloc(&A, 1 << 20); hipMemset(A, 0, 1 << 20);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 3000;
    hipLaunchKernelGGL((k<FORM>), dim3(188), dim3(256), 0, 0, A, out, cyc, 30);
    hipLaunchKernelGGL((k<FORM>), dim3(188), dim3(256), 0, 0, A, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(188 * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    printf("%-58s %6.2f cycles/MFMA  (+%.1f cycles per block of 32)\n", name, mean / (iters * 32.0), mean / iters - 32 * 32.46);
    hipFree(A); hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("no loads");
    run<1>("4 x global_load_dwordx4 (64-bit vaddr)");
    run<6>("4 x global_load_dwordx4 (uniform base + 32-bit voffset)");
    run<2>("4 x buffer_load_dwordx4 (rsrc + 32-bit voffset)");
    run<3>("8 x global_load_dwordx2");
    run<4>("16 x global_load_dword");
    run<5>("4 x ds_read_b128");
    return 0;
}
