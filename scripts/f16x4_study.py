"""Round 5: accuracy of fp32 dot products emulated on the fp16 matrix pipe (mcpc_gemm_f16.h) against the fp32 MFMA chain and round 3-4's
bf16x6, relative to sum |terms|, vs an fp64 reference (NumPy emulation; `max / rms` over 4000 rows).  f16x4: two fp16 pieces per operand,
four products; f16x3: without the m*m term.  Row scale: a power of two per B row (what the kernel does); tile scale: one per 16 rows."""
import numpy as np
rng = np.random.default_rng(0)
f32 = np.float32
def bf16_round(a):  # RNE to bf16, returned as float32
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split_bf16x3(a):
    h = bf16_round(a); r = (a - h).astype(f32); m = bf16_round(r); l = bf16_round((r - m).astype(f32)); return h, m, l
def split_f16x2(a, scale):
    s = (a * scale).astype(f32)
    h = s.astype(np.float16).astype(f32); m = (s - h).astype(f32).astype(np.float16).astype(f32)
    return h, m
def dot_chain_f32(terms_list):  # accumulate list of product arrays [K] in fp32 sequentially over k, terms order given
    acc = np.zeros(terms_list[0].shape[:-1], f32)
    K = terms_list[0].shape[-1]
    for k0 in range(0, K, 32):      # MFMA k-block: products of a block summed (exact-ish fp32 adds inside), then added to acc
        for t in terms_list:
            blk = t[..., k0:k0 + 32].astype(np.float64).sum(-1)  # inside one MFMA: treat as exact sum, one rounding into acc
            acc = (acc + blk.astype(f32)).astype(f32)
    return acc
def study(K, wscale, xgen, n=4000):
    W = (rng.uniform(-1, 1, (n, K)) * wscale).astype(f32); X = xgen((n, K)).astype(f32)
    exact = (W.astype(np.float64) * X.astype(np.float64)).sum(-1); denom = (np.abs(W.astype(np.float64) * X.astype(np.float64))).sum(-1) + 1e-300
    # fp32 MFMA chain: k-steps of 4, fp32 accumulate
    acc = np.zeros(n, f32)
    for k0 in range(0, K, 4):
        acc = (acc + (W[:, k0:k0+4].astype(np.float64) * X[:, k0:k0+4]).sum(-1).astype(f32)).astype(f32)
    e32 = np.abs(acc - exact) / denom
    wh, wm, wl = split_bf16x3(W); xh, xm, xl = split_bf16x3(X)
    p = lambda a, b: (a.astype(np.float64) * b)
    a6 = dot_chain_f32([p(wm, xm), p(wl, xh), p(wh, xl), p(wm, xh), p(wh, xm), p(wh, xh)])
    e6 = np.abs(a6 - exact) / denom
    sw = 2.0 ** (14 - np.ceil(np.log2(np.abs(W).max())))
    # per-row (chain) scale for X: power of two with max*s <= 2^14
    mx = np.abs(X).max(-1, keepdims=True); sx = 2.0 ** (14 - np.ceil(np.log2(np.maximum(mx, 1e-30))))
    w1, w2 = split_f16x2(W, sw); x1, x2 = split_f16x2(X, sx)
    a4 = dot_chain_f32([p(w2, x2), p(w2, x1), p(w1, x2), p(w1, x1)])
    a4 = a4.astype(np.float64) / sw / sx[:, 0]
    e4 = np.abs(a4 - exact) / denom
    # tile-wide scalar scale for X (max over 16 rows)
    mx16 = np.repeat(np.abs(X).reshape(n // 16, 16 * K).max(-1), 16)[:, None]; sx16 = 2.0 ** (14 - np.ceil(np.log2(np.maximum(mx16, 1e-30))))
    x1, x2 = split_f16x2(X, sx16)
    a4t = dot_chain_f32([p(w2, x2), p(w2, x1), p(w1, x2), p(w1, x1)]).astype(np.float64) / sw / sx16[:, 0]
    e4t = np.abs(a4t - exact) / denom
    a3 = dot_chain_f32([p(w2, x1), p(w1, x2), p(w1, x1)]).astype(np.float64) / sw / sx16[:, 0]
    e3 = np.abs(a3 - exact) / denom
    return [(e.max(), np.sqrt((e ** 2).mean())) for e in (e32, e6, e4, e4t, e3)]
gens = {
 "relu(N(0,3))": lambda s: np.maximum(rng.normal(0, 3, s), 0),
 "errors N(0,1)": lambda s: rng.normal(0, 1, s),
 "tiny errors N(0,1e-4)": lambda s: rng.normal(0, 1e-4, s),
 "mixed rows (row scale 10^U(-5,1))": lambda s: rng.normal(0, 1, s) * 10 ** rng.uniform(-5, 1, (s[0], 1)),
 "heavy tail within row": lambda s: rng.normal(0, 1, s) * 10 ** rng.uniform(-4, 0, s),
}
print("%-36s %5s | %-19s | %-19s | %-19s | %-19s | %-19s" % ("B operand", "K", "fp32 MFMA chain", "bf16x6", "f16x4 row-scale", "f16x4 tile-scale", "f16x3 (no m*m)"))
for name, g in gens.items():
    for K in (32, 64, 96, 128, 256, 784):
        r = study(K, 1 / np.sqrt(K), g)
        print("%-36s %5d | " % (name, K) + " | ".join("%.2e / %.2e" % x for x in r))
