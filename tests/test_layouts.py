"""CPU check of the MFMA fragment layouts the HIP kernels rely on.

A NumPy model of ``v_mfma_f32_16x16x4_f32`` (lane maps from the CDNA4 guide: A[i][k] in lane
i+16k, B[k][j] in lane j+16k, D[i][j] in lane j+16*(i//4), register i%4) is driven with the same
index formulas as ``csrc/mcpc_kernels.h`` (pack_kernel, gemm_tiles, dw_kernel).  If these tests
pass, the packing and the accumulator interpretation written in the kernels are consistent.
"""
import numpy as np

LANES = np.arange(64)


def mfma16(a, b, c):
    """a, b: [64]; c: [64, 4] -> d [64, 4]."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[LANES & 15, LANES >> 4] = a
    B[LANES >> 4, LANES & 15] = b
    C = np.zeros((16, 16))
    for reg in range(4):
        C[4 * (LANES >> 4) + reg, LANES & 15] = c[:, reg]
    D = A @ B + C
    d = np.zeros((64, 4))
    for reg in range(4):
        d[:, reg] = D[4 * (LANES >> 4) + reg, LANES & 15]
    return d


def pack_forward(W, out_tiles, in_tiles):
    n_out, n_in = W.shape
    Wf = np.zeros((out_tiles, in_tiles, 64, 4))
    for ut in range(out_tiles):
        for kb in range(in_tiles):
            for r in range(4):
                u = 16 * ut + (LANES & 15)
                k = 16 * kb + 4 * (LANES >> 4) + r
                ok = (u < n_out) & (k < n_in)
                Wf[ut, kb, ok, r] = W[u[ok], k[ok]]
    return Wf


def pack_backward(W, out_tiles, in_tiles):
    n_out, n_in = W.shape
    Wb = np.zeros((in_tiles, out_tiles, 64, 4))
    for it in range(in_tiles):
        for ub in range(out_tiles):
            for r in range(4):
                u = 16 * ub + 4 * (LANES >> 4) + r
                i = 16 * it + (LANES & 15)
                ok = (u < n_out) & (i < n_in)
                Wb[it, ub, ok, r] = W[u[ok], i[ok]]
    return Wb


def gemm_tile(Apacked_tile, Blds, nkb, chain_tile):
    """Apacked_tile: [nkb, 64, 4]; Blds: [chains, K] row-major; returns acc [64, 4] (C layout)."""
    c = LANES & 15
    q = LANES >> 4
    acc = np.zeros((64, 4))
    for kb in range(nkb):
        b = np.stack([Blds[16 * chain_tile + c, 16 * kb + 4 * q + r] for r in range(4)], axis=1)
        a = Apacked_tile[kb]
        for r in range(4):
            acc = mfma16(a[:, r], b[:, r], acc)
    return acc


def c_layout_to_matrix(acc):
    """acc [64, 4] of tile -> [16 units, 16 chains]."""
    M = np.zeros((16, 16))
    for reg in range(4):
        M[4 * (LANES >> 4) + reg, LANES & 15] = acc[:, reg]
    return M


def test_forward_tiles_reproduce_linear():
    rs = np.random.RandomState(0)
    n_out, n_in, chains = 37, 30, 32
    ot, it = 3, 2
    W = rs.randn(n_out, n_in)
    FX = np.zeros((chains, it * 16)); FX[:, :n_in] = rs.randn(chains, n_in)
    Wf = pack_forward(W, ot, it)
    ref = FX[:, :n_in] @ W.T                      # [chains, n_out]
    for ut in range(ot):
        for ct in range(2):
            M = c_layout_to_matrix(gemm_tile(Wf[ut], FX, it, ct))      # [unit, chain]
            for m in range(16):
                u = 16 * ut + m
                want = ref[16 * ct:16 * ct + 16, u] if u < n_out else np.zeros(16)
                np.testing.assert_allclose(M[m], want, atol=1e-12)


def test_backward_tiles_reproduce_error_backprojection():
    rs = np.random.RandomState(1)
    n_out, n_in, chains = 40, 21, 32
    ot, it = 3, 2
    W = rs.randn(n_out, n_in)
    E = np.zeros((chains, ot * 16)); E[:, :n_out] = rs.randn(chains, n_out)
    Wb = pack_backward(W, ot, it)
    ref = E[:, :n_out] @ W                        # [chains, n_in]
    for t in range(it):
        for ct in range(2):
            M = c_layout_to_matrix(gemm_tile(Wb[t], E, ot, ct))        # [in unit, chain]
            for m in range(16):
                i = 16 * t + m
                want = ref[16 * ct:16 * ct + 16, i] if i < n_in else np.zeros(16)
                np.testing.assert_allclose(M[m], want, atol=1e-12)


def test_backward_chunked_matches_full():
    """Read-out back-projection accumulated chunk by chunk (kb window) equals the full product."""
    rs = np.random.RandomState(2)
    n_out, n_in, chains = 80, 16, 32
    ot, it = 5, 1
    W = rs.randn(n_out, n_in)
    E = rs.randn(chains, ot * 16)
    Wb = pack_backward(W, ot, it)
    full = c_layout_to_matrix(gemm_tile(Wb[0], E, ot, 0))
    acc = np.zeros((64, 4))
    c = LANES & 15; q = LANES >> 4
    for tile0, ntc in ((0, 2), (2, 2), (4, 1)):
        chunk = E[:, 16 * tile0:16 * (tile0 + ntc)]       # what the kernel keeps in LDS
        for kb in range(ntc):
            b = np.stack([chunk[c, 16 * kb + 4 * q + r] for r in range(4)], axis=1)
            a = Wb[0, tile0 + kb]
            for r in range(4):
                acc = mfma16(a[:, r], b[:, r], acc)
    np.testing.assert_allclose(c_layout_to_matrix(acc), full, atol=1e-12)


def test_dw_kernel_strided_tiles():
    """mcpc_dw_kernel: component i/j of a 16-byte load selects the strided tile; epilogue index map."""
    rs = np.random.RandomState(3)
    rows, ne, na = 24, 80, 48          # padded widths (multiples of 16)
    E = rs.randn(rows, ne); A = rs.randn(rows, na)
    ref = E.T @ A
    G = np.zeros((ne, na))
    m = LANES & 15; q = LANES >> 4
    for te in range((ne + 63) // 64):
        for ta in range((na + 63) // 64):
            ue = 64 * te + 4 * m; ua = 64 * ta + 4 * m
            ve = ue < ne; va = ua < na
            acc = np.zeros((4, 4, 64, 4))
            for r in range(0, rows, 4):
                e = np.zeros((64, 4)); a = np.zeros((64, 4))
                for k in range(4):
                    e[ve, k] = E[r + q[ve], ue[ve] + k]
                    a[va, k] = A[r + q[va], ua[va] + k]
                for i in range(4):
                    for j in range(4):
                        acc[i, j] = mfma16(e[:, i], a[:, j], acc[i, j])
            for i in range(4):
                for reg in range(4):
                    u = 64 * te + 4 * (4 * q + reg) + i
                    for lane in range(64):
                        if u[lane] < ne and va[lane]:
                            G[u[lane], ua[lane]:ua[lane] + 4] = [acc[i, j, lane, reg] for j in range(4)]
    np.testing.assert_allclose(G, ref, atol=1e-10)
