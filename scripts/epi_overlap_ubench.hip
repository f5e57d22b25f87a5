// Developer micro-benchmark: what slows an epilogue-like wave down when its SIMD partner streams fp32 MFMAs?
// E kinds: 0 = sigmoid/BCE math only, 1 = + LDS quad write/read per quad, 2 = + global nt load/store per quad,
//          3 = + a scalar (kernel-argument) reload per quad.   G kinds: 0 idle, 1 MFMA only, 2 MFMA + L2 fragment stream + LDS reads.
// hipcc --offload-arch=gfx950 -O3 scripts/epi_overlap_ubench.hip -o scripts/bin/epi_overlap_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gf32x4;

__device__ __forceinline__ void sigmoid_bce(float o, float y, float& sig, float& bce) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(o) * 1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    sig = (o >= 0.0f ? 1.0f : e) * r;
    bce = fmaxf(o, 0.0f) - o * y + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e);
}

struct Args { const int* sc; const f32x4* A; float* xg; float* out; unsigned long long* cyc; int g_iters, e_iters; int n, mask; float scale; int prio; int pad[32]; };

template <int GK, int EK>
__global__ __launch_bounds__(512) void k(const Args a) {
    __shared__ __attribute__((aligned(16))) float lds[32 * 260 + 8 * 64 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 260; i += 512) lds[i] = i * 1e-4f;
    __syncthreads();
    unsigned long long m0 = 0, m1 = 0;
    float res = 0.f;
    if (wave < 4) {
        if (GK == 0) { a.out[blockIdx.x * 512 + threadIdx.x] = 0.f; return; }
        if (a.prio == -1) __builtin_amdgcn_s_setprio(2);
        const gf32x4* A = (const gf32x4*)a.A;
        const int c = lane & 15, q = lane >> 4;
        const float* bp = lds + c * 260 + 4 * q;
        f32x4 acc[2][2];
        for (int t = 0; t < 2; ++t) for (int ct = 0; ct < 2; ++ct) acc[t][ct] = {0.f, 0.f, 0.f, 0.f};
        f32x4 fa[2] = {{1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}}, fb[2] = {{1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}};
        f32x4 na[2], nb[2];
        int off = (blockIdx.x & 7) * 4096 + wave * 2048;
        m0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < a.g_iters; ++it) {
            if (GK == 2) {
                const int kb = it & 15;
                na[0] = A[off + kb * 64 + lane]; na[1] = A[off + 1024 + kb * 64 + lane];
                nb[0] = *(const f32x4*)(bp + kb * 16); nb[1] = *(const f32x4*)(bp + 16 * 260 + kb * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t][r], fb[ct][r], acc[t][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (GK == 2) { fa[0] = na[0]; fa[1] = na[1]; fb[0] = nb[0]; fb[1] = nb[1]; }
        }
        m1 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < 2; ++t) for (int ct = 0; ct < 2; ++ct) res += acc[t][ct].x + acc[t][ct].y + acc[t][ct].z + acc[t][ct].w;
    } else {
        if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
        float* stage = lds + 32 * 260 + (wave - 4) * 256 * 2;
        float* xg = a.xg + ((size_t)blockIdx.x * 4 + (wave - 4)) * 64 * 4 * 64;
        f32x4 o = {0.1f * lane, -0.2f * lane, 0.05f * lane, 0.3f}, y = {0.f, 1.f, 0.f, 1.f};
        float lsum = 0.f;
        m0 = __builtin_amdgcn_s_memtime();
        if (EK == 4) {
            // same arithmetic, 4 independent quads per iteration written stage by stage (16 exps, then 16 rcps, ...)
            f32x4 oq[4] = {o, o * 1.1f, o * 0.9f, o * 1.2f};
            for (int it = 0; it < a.e_iters / 4; ++it) {
                float ov[16], ex[16], rc[16], lg[16], ev[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) ov[j] = oq[j >> 2][j & 3];
#pragma unroll
                for (int j = 0; j < 16; ++j) ex[j] = __builtin_amdgcn_exp2f(-fabsf(ov[j]) * 1.4426950408889634f);
#pragma unroll
                for (int j = 0; j < 16; ++j) rc[j] = __builtin_amdgcn_rcpf(1.0f + ex[j]);
#pragma unroll
                for (int j = 0; j < 16; ++j) lg[j] = __builtin_amdgcn_logf(1.0f + ex[j]);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float yv = y[j & 3];
                    const bool on = (4 * lane + (j & 3)) >= a.mask && (4 * lane + (j & 3)) < a.n;
                    const float sg = (ov[j] >= 0.0f ? 1.0f : ex[j]) * rc[j];
                    const float bc = fmaxf(ov[j], 0.0f) - ov[j] * yv + 0.6931471805599453f * lg[j];
                    ev[j] = on ? sg - yv : 0.f;
                    lsum += on ? bc : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) oq[j >> 2][j & 3] += ev[j] * a.scale;
            }
            o = oq[0] + oq[1] + oq[2] + oq[3];
        } else
        for (int it = 0; it < a.e_iters; ++it) {
            f32x4 yy = y;
            if (EK == 2) yy = __builtin_nontemporal_load((const f32x4*)(xg + ((it & 63) * 64 + lane) * 4));
            int n = a.n, mask = a.mask;
            if (EK == 3) {   // a scalar reload per quad, as hipcc emits under SGPR pressure
                typedef __attribute__((address_space(4))) const int cint;
                cint* sc = (cint*)a.sc;
                asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x4\n\ts_waitcnt lgkmcnt(0)" : "=s"(n), "=s"(mask) : "s"(sc) : "memory");
            }
            const float ov[4] = {o.x, o.y, o.z, o.w}, yv[4] = {yy.x, yy.y, yy.z, yy.w};
            float ev[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on = (4 * lane + r) >= mask && (4 * lane + r) < n;
                float sg, bc;
                sigmoid_bce(ov[r], yv[r], sg, bc);
                ev[r] = on ? sg - yv[r] : 0.f;
                lsum += on ? bc : 0.f;
            }
            f32x4 e = {ev[0], ev[1], ev[2], ev[3]};
            if (EK == 1) { *(f32x4*)(stage + lane * 4) = e; e = *(const f32x4*)(stage + ((lane + 1) & 63) * 4); }
            if (EK == 2) __builtin_nontemporal_store(e, (f32x4*)(xg + ((it & 63) * 64 + lane) * 4));
            o = o + e * a.scale;
        }
        m1 = __builtin_amdgcn_s_memtime();
        res = lsum + o.x + o.y + o.z + o.w;
    }
    a.out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0) a.cyc[blockIdx.x * 8 + wave] = m1 - m0;
}

template <int GK, int EK>
void run(Args a, int gi, int ei) {
    a.g_iters = gi; a.e_iters = ei;
    hipMemset(a.cyc, 0, 256 * 8 * 8);
    hipLaunchKernelGGL((k<GK, EK>), dim3(256), dim3(512), 0, 0, a);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), a.cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double g = 0, e = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? g : e) += (double)h[b * 8 + w];
    printf("prio %2d  G kind %d, E kind %d:  G %6.1f cycles per MFMA   E %7.1f cycles per quad\n", a.prio, GK, EK, gi ? g / 1024 / (gi * 16.0) : 0.0, ei ? e / 1024 / ei : 0.0);
}

int main() {
    Args a{};
    f32x4* A; hipMalloc(&A, 8 * 4096 * 16 * 2); hipMemset(A, 0, 8 * 4096 * 16 * 2); a.A = A;
    hipMalloc(&a.xg, (size_t)256 * 4 * 64 * 4 * 64 * 4); hipMemset(a.xg, 0, (size_t)256 * 4 * 64 * 4 * 64 * 4);
    hipMalloc(&a.out, 256 * 512 * 4); hipMalloc(&a.cyc, 256 * 8 * 8);
    a.n = 250; a.mask = 2; a.scale = 1e-3f;
    { int* sc; hipMalloc(&sc, 64); int h[2] = {250, 2}; hipMemcpy(sc, h, 8, hipMemcpyHostToDevice); a.sc = sc; }
    const int GI = 4000, EI = 400;
    for (int prio : {0, 1, 2, -1}) {       // 0: both default, 1/2: E raised, -1: G raised
        a.prio = prio;
        run<0, 0>(a, 0, EI); run<1, 0>(a, GI, EI); run<2, 0>(a, GI, EI);
        run<1, 1>(a, GI, EI); run<2, 1>(a, GI, EI);
        run<1, 2>(a, GI, EI); run<2, 2>(a, GI, EI);
        run<0, 4>(a, 0, EI); run<1, 4>(a, GI, EI); run<2, 4>(a, GI, EI);
    }
    return 0;
}
