// Study for a later round (NOT used by the product): fp32 GEMM emulated with bf16 MFMAs on split operands.
//   x = hi + mid + lo  (three bf16 values, exact for almost every fp32 x);   a*b ~ sum of the leading cross terms
//   x3: hi*hi + hi*mid + mid*hi                      (16-bit operands)
//   x6: x3 + hi*lo + lo*hi + mid*mid                 (drops terms below 2^-24 relative)
//   x9: all nine terms
// Part 1 (accuracy): out[16 units][16 chains] = W[16][K] . act[16][K]^T, K = 256 and 784, data shaped like the step kernel's
//   (weights U(-1/sqrt(K), 1/sqrt(K)), activations relu(N(0,1))), against an fp64 host reference; the fp32 MFMA the product
//   uses today is measured the same way.
// Part 2 (rate): cycles per v_mfma_f32_16x16x32_bf16 back to back, next to v_mfma_f32_16x16x4_f32.
// hipcc --offload-arch=gfx950 -O3 scripts/bf16split_study.hip -o scripts/bin/bf16split_study
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }      // round to nearest even

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = to_bf16(x);
    const float r1 = x - (float)h;
    m = to_bf16(r1);
    const float r2 = r1 - (float)m;
    l = to_bf16(r2);
}

// one wave: 16 x 16 outputs.  A operand lane (m = lane & 15, g = lane >> 4) holds W[m][32 kb + 8 g .. + 7];
// B operand lane (n = lane & 15, g) holds act[n][32 kb + 8 g .. + 7];  C lane (n, q) regs r: unit 4 q + r, chain n.
__device__ __forceinline__ void split3_trunc(float x, __bf16& h, __bf16& m, __bf16& l) {
    // truncation: each piece keeps the next 8 significant bits, the pieces add up to x exactly; bf16 = upper half of the fp32 word
    const uint32_t xb = __float_as_uint(x), hb = xb & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);
    const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const uint32_t lb = __float_as_uint(r2) & 0xffff0000u;
    h = __builtin_bit_cast(__bf16, (unsigned short)(hb >> 16)); m = __builtin_bit_cast(__bf16, (unsigned short)(mb >> 16));
    l = __builtin_bit_cast(__bf16, (unsigned short)(lb >> 16));
}

__global__ void accuracy(const float* __restrict__ W, const float* __restrict__ X, int K, float* out32, float* out3, float* out6, float* out9, float* out6t) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, g = lane >> 4;
    f32x4 c32 = {0, 0, 0, 0}, c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, ct = {0, 0, 0, 0};
    for (int kb = 0; kb < K / 32; ++kb) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            __bf16 h, mm, l;
            split3(W[m * K + 32 * kb + 8 * g + j], h, mm, l); ah[j] = h; am[j] = mm; al[j] = l;
            split3(X[m * K + 32 * kb + 8 * g + j], h, mm, l); bh[j] = h; bm[j] = mm; bl[j] = l;
        }
        {   // the form a kernel would use: truncation split, ONE accumulator, small terms first within the block
            bf16x8 th, tm, tl, uh, um, ul;
            for (int j = 0; j < 8; ++j) {
                __bf16 h, mm, l;
                split3_trunc(W[m * K + 32 * kb + 8 * g + j], h, mm, l); th[j] = h; tm[j] = mm; tl[j] = l;
                split3_trunc(X[m * K + 32 * kb + 8 * g + j], h, mm, l); uh[j] = h; um[j] = mm; ul[j] = l;
            }
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tm, um, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl, uh, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, ul, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tm, uh, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, um, ct, 0, 0, 0);
            ct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, uh, ct, 0, 0, 0);
        }
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bl, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, c3, 0, 0, 0);
        // fp32 MFMA over the same 32 k: 8 x 16x16x4, lane (m, g) feeds k = 32 kb + 4 j + g
        for (int j = 0; j < 8; ++j)
            c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[m * K + 32 * kb + 4 * j + g], X[m * K + 32 * kb + 4 * j + g], c32, 0, 0, 0);
    }
    // C layout: lane (n = lane & 15, q = lane >> 4), reg r -> out[unit 4 q + r][chain n]
    for (int r = 0; r < 4; ++r) {
        const int u = 4 * g + r, n = m;
        out32[u * 16 + n] = c32[r];
        out3[u * 16 + n] = c0[r] + c1[r];
        out6[u * 16 + n] = c0[r] + (c1[r] + c2[r]);
        out9[u * 16 + n] = c0[r] + (c1[r] + (c2[r] + c3[r]));
        out6t[u * 16 + n] = ct[r];
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.0f + j); }
    float fa = threadIdx.x * 1e-3f, fb = 1.5f;
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                else if (KIND == 2) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 1], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i], 0, 0, 0);
            }
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = m1 - m0;
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }

int main() {
    srand(1);
    for (int K : {256, 784 / 32 * 32}) {
        std::vector<float> W(16 * K), X(16 * K);
        for (auto& w : W) w = (float)((2 * urand() - 1) / std::sqrt((double)K));
        for (auto& x : X) { double v = nrand() * 3.0; x = (float)(v > 0 ? v : 0.0); }
        std::vector<double> ref(256);
        double scale = 0;
        for (int u = 0; u < 16; ++u) for (int n = 0; n < 16; ++n) {
            double s = 0, sa = 0;
            for (int k = 0; k < K; ++k) { s += (double)W[u * K + k] * (double)X[n * K + k]; sa += std::fabs((double)W[u * K + k] * (double)X[n * K + k]); }
            ref[u * 16 + n] = s; scale += sa;
        }
        scale /= 256;      // mean of sum |a_k b_k|: the natural scale of the rounding error of a dot product
        float *dW, *dX, *o[5];
        hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4);
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
        for (auto& p : o) hipMalloc(&p, 256 * 4);
        hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dW, dX, K, o[0], o[1], o[2], o[3], o[4]);
        hipDeviceSynchronize();
        const char* names[5] = {"fp32 MFMA (16x16x4_f32), as shipped", "bf16 x3 (hi,mid)", "bf16 x6", "bf16 x9", "bf16 x6, truncation split, one accumulator"};
        printf("K = %d   (errors relative to mean sum|a b| = %.3g; fp32 epsilon = 5.96e-8)\n", K, scale);
        for (int v = 0; v < 5; ++v) {
            std::vector<float> h(256);
            hipMemcpy(h.data(), o[v], 256 * 4, hipMemcpyDeviceToHost);
            double mx = 0, rms = 0;
            for (int i = 0; i < 256; ++i) { const double e = std::fabs((double)h[i] - ref[i]) / scale; mx = std::max(mx, e); rms += e * e; }
            printf("  %-40s max %.3e   rms %.3e\n", names[v], mx, std::sqrt(rms / 256));
        }
        hipFree(dW); hipFree(dX); for (auto& p : o) hipFree(p);
    }
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    for (int nb : {256, 32}) {
        for (int kind = 0; kind < 3; ++kind) {
            const int iters = 2000;
            auto launch = [&](int n) {
                if (kind == 0) hipLaunchKernelGGL((rate<0>), dim3(nb), dim3(256), 0, 0, out, cyc, n);
                else if (kind == 1) hipLaunchKernelGGL((rate<1>), dim3(nb), dim3(256), 0, 0, out, cyc, n);
                else hipLaunchKernelGGL((rate<2>), dim3(nb), dim3(256), 0, 0, out, cyc, n);
            };
            launch(10); hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(nb * 4);
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
            const double per = mean / (iters * 32.0);
            const double ns = ms * 1e6 / (iters * 32.0);
            if (kind == 1) printf("%3d workgroups  v_mfma_f32_16x16x4_f32:              %.2f ticks, %.2f ns each -> 32 k of a 16x16 tile: %.0f ns\n", nb, per, ns, 8 * ns);
            else printf("%3d workgroups  v_mfma_f32_16x16x32_bf16 (%d accum.): %.2f ticks, %.2f ns each -> 32 k of a 16x16 tile: x3 %.0f, x6 %.0f, x9 %.0f ns\n", nb, kind == 0 ? 8 : 2, per, ns, 3 * ns, 6 * ns, 9 * ns);
        }
    }
    return 0;
}
