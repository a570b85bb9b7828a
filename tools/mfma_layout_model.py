"""Lane-level numpy model of the MFMA register layouts used by csrc/deform.hip (development aid, CPU only).

Mirrors the index arithmetic of DenseTrunk / DenseIL / DenseT / store_il / the dW2 and dh1 blocks with an exact model of
v_mfma_f32_32x32x2_f32 (A[i][k] in lane i+32k, B[k][j] in lane j+32k, D[i][j] in lane j+32*hh register r with
i = rho(r,hh)), and checks every product against plain matrix algebra.  Run: python tools/mfma_layout_model.py
"""
import numpy as np

LANES = np.arange(64)
G = LANES & 31
H = LANES >> 5


def rho(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma32(a, b, c):
    """a, b: [64]; c: [64,16] -> new c."""
    A = np.stack([a[:32], a[32:]], 1)          # A[i][k]
    B = np.stack([b[:32], b[32:]], 0)          # B[k][j]
    D = A @ B                                  # [32 i][32 j]
    out = c.copy()
    for r in range(16):
        for hh in range(2):
            out[32 * hh:32 * hh + 32, r] += D[rho(r, hh), :]
    return out


def mfma4(a, b, c):
    """v_mfma_f32_4x4x1_16b_f32: 16 blocks; lane 4b+i holds A_b[i], lane 4b+j holds B_b[j], c[lane 4b+j, i] += A_b[i]*B_b[j]."""
    out = c.copy()
    for blk in range(16):
        A = a[4 * blk:4 * blk + 4]
        B = b[4 * blk:4 * blk + 4]
        out[4 * blk:4 * blk + 4, :] += np.outer(B, A)   # [lane j][reg i]
    return out


def small_head(W2, b2, k, X, KT):
    """k <= 4 outputs on the 4x4x1 MFMA (DenseIL::run4): returns out[lane, i]."""
    acc = np.zeros((64, 4))
    row = np.minimum(LANES & 3, k - 1)
    for s in range(16):
        for t in range(KT):
            a = W2[row, KT * rho(s, 0) + KT * 4 * H + t]
            acc = mfma4(a, X[t][:, s], acc)
    tot = acc + acc[LANES ^ 32]
    return tot + b2[np.minimum(np.arange(4), k - 1)][None, :]


def dense_trunk(W0, b0, feat, FCH, OT):
    """feat[tile][lane, reg] chunk layout -> hid[ot][lane, reg] interleaved rows (row = OT*i + ot)."""
    Y = [np.zeros((64, 16)) for _ in range(OT)]
    for ot in range(OT):
        for r in range(16):
            Y[ot][:, r] = b0[OT * rho(r, H) + ot]
    for j in range(FCH):
        for c in range(4):
            for ot in range(OT):
                a = W0[OT * G + ot, 8 * j + 4 * H + c]
                Y[ot] = mfma32(a, feat[j // 4][:, 4 * (j % 4) + c], Y[ot])
    return Y


def dense_il(Wm, bias, out_dim, X, KT, OT, row_il):
    Y = [np.zeros((64, 16)) for _ in range(OT)]
    for ot in range(OT):
        for r in range(16):
            row = OT * rho(r, H) + ot if row_il else 32 * ot + rho(r, H)
            Y[ot][:, r] = bias[np.minimum(row, out_dim - 1)]
    for s in range(16):
        for t in range(KT):
            for ot in range(OT):
                row = OT * G + ot if row_il else 32 * ot + G
                row = np.minimum(row, out_dim - 1)
                a = Wm[row, KT * rho(s, 0) + KT * 4 * H + t]
                Y[ot] = mfma32(a, X[t][:, s], Y[ot])
    return Y


def dense_T(Wm, in_valid, dY, YT, XT, col_il):
    dX = [np.zeros((64, 16)) for _ in range(XT)]
    for s in range(16 * YT):
        r, t = s // YT, s % YT
        wrow = YT * rho(r, 0) + t + YT * 4 * H
        b = dY[t][:, r]
        for xt in range(XT):
            col = XT * G + xt if col_il else np.minimum(32 * xt + G, in_valid - 1)
            dX[xt] = mfma32(Wm[wrow, col], b, dX[xt])
    return dX


def il_to_matrix(X, T):
    """interleaved tiles -> [32 gaussians, 32*T features]."""
    M = np.zeros((32, 32 * T))
    for t in range(T):
        for r in range(16):
            for hh in range(2):
                M[:, T * rho(r, hh) + t] = X[t][32 * hh:32 * hh + 32, r]
    return M


def tile_to_matrix(X, T):
    """tile layout (row = 32*t + rho) -> [32 gaussians, 32*T]."""
    M = np.zeros((32, 32 * T))
    for t in range(T):
        for r in range(16):
            for hh in range(2):
                M[:, 32 * t + rho(r, hh)] = X[t][32 * hh:32 * hh + 32, r]
    return M


def matrix_to_chunks(F, FCH):
    """[32, 8*FCH] -> chunk-layout tiles."""
    FT = (FCH + 3) // 4
    X = [np.zeros((64, 16)) for _ in range(FT)]
    for j in range(FCH):
        for c in range(4):
            X[j // 4][:, 4 * (j % 4) + c] = F[G, 8 * j + 4 * H + c]
    return X


def matrix_to_il(M, T):
    X = [np.zeros((64, 16)) for _ in range(T)]
    for t in range(T):
        for r in range(16):
            X[t][:, r] = M[G, T * rho(r, H) + t]
    return X


def check(WT, FCH, k, rng):
    W, F = 32 * WT, 8 * FCH
    feat = rng.standard_normal((32, F))
    W0, b0 = rng.standard_normal((W, F)), rng.standard_normal(W)
    W1, b1 = rng.standard_normal((W, W)), rng.standard_normal(W)
    W2, b2 = rng.standard_normal((k, W)), rng.standard_normal(k)
    hid_ref = np.maximum(feat @ W0.T + b0, 0)
    h1_ref = np.maximum(hid_ref @ W1.T + b1, 0)
    out_ref = h1_ref @ W2.T + b2
    # forward chain
    hid = dense_trunk(W0, b0, matrix_to_chunks(feat, FCH), FCH, WT)
    hid = [np.maximum(x, 0) for x in hid]
    assert np.allclose(il_to_matrix(hid, WT), hid_ref), "trunk"
    h1 = [np.maximum(x, 0) for x in dense_il(W1, b1, W, hid, WT, WT, True)]
    assert np.allclose(il_to_matrix(h1, WT), h1_ref), "L1"
    nt2 = 2 if k > 32 else 1
    outs = []
    for ot2 in range(nt2):
        kk = min(k - 32 * ot2, 32)
        o = dense_il(W2[32 * ot2:], b2[32 * ot2:], kk, h1, WT, 1, False)
        outs.append(tile_to_matrix(o, 1)[:, :kk])
    assert np.allclose(np.concatenate(outs, 1), out_ref), "L2"
    if k <= 4:
        o4 = small_head(W2, b2, k, h1, WT)
        assert np.allclose(o4[:32, :k], out_ref) and np.allclose(o4[32:, :k], out_ref), "L2 small-head form"
    # backward: dh1 = G W2 (masked), dhid += dh1 W1, dfeat = dhid W0
    Gm = rng.standard_normal((32, k))
    dh1_ref = (Gm @ W2) * (h1_ref > 0)
    dh1 = [np.zeros((64, 16)) for _ in range(WT)]
    for s in range((k + 1) // 2):
        o = 2 * s + H
        a_rows = np.minimum(o, k - 1)
        b = np.where(o < k, Gm[G, np.minimum(o, k - 1)], 0.0)
        for t in range(WT):
            dh1[t] = mfma32(W2[a_rows, WT * G + t], b, dh1[t])
    dh1 = [x * (hh > 0) for x, hh in zip(dh1, h1)]
    assert np.allclose(il_to_matrix(dh1, WT), dh1_ref), "dh1"
    dhid = dense_T(W1, W, dh1, WT, WT, True)
    dhid_ref = dh1_ref @ W1
    assert np.allclose(il_to_matrix(dhid, WT), dhid_ref), "dhid"
    FT = (FCH + 3) // 4
    dfeat = dense_T(W0, F, dhid, WT, FT, False)
    assert np.allclose(tile_to_matrix(dfeat, FT)[:, :F], dhid_ref @ W0), "dfeat"
    # dW2 through the transposed LDS tile: lds[g][feature] = il_to_matrix(h1)
    lds = il_to_matrix(h1, WT)
    dW2 = np.zeros((k, W))
    for ot2 in range(nt2):
        o = 32 * ot2 + G
        kk = k - 32 * ot2
        for tb in range(WT):
            acc = np.zeros((64, 16))
            for s in range(16):
                ga = np.where(o < k, Gm[2 * s + H, np.minimum(o, k - 1)], 0.0)
                acc = mfma32(ga, lds[2 * s + H, tb * 32 + G], acc)
            for r in range(16):
                for lane in range(64):
                    orow = rho(r, lane >> 5)
                    if orow < kk:
                        dW2[32 * ot2 + orow, tb * 32 + (lane & 31)] += acc[lane, r]
    assert np.allclose(dW2, Gm.T @ h1_ref), "dW2"
    if k <= 4:
        # register-resident 4x4x1 form (small_dw2_steps): A broadcast from block gq%16 of sa0/sa1, B = lds row gq
        sa = [np.where((LANES & 3) < k, Gm[16 * x + (LANES >> 2), np.minimum(LANES & 3, k - 1)], 0.0) for x in range(2)]
        NCH = W // 64
        acc = [np.zeros((64, 4)) for _ in range(NCH)]
        for gq in range(32):
            blk = gq & 15
            a_b = np.tile(sa[gq >> 4][4 * blk:4 * blk + 4], 16)          # CBSZ=4, ABID=blk: block blk of A feeds every block
            for u in range(NCH):
                acc[u] = mfma4(a_b, lds[gq, 64 * u + LANES], acc[u])
        dW2s = np.zeros((k, W))
        for u in range(NCH):
            for i in range(k):
                dW2s[i, 64 * u + LANES] = acc[u][:, i]
        assert np.allclose(dW2s, Gm.T @ h1_ref), "dW2 small-head form"
        sb = sa[0] + sa[1]
        db = np.array([sb[(LANES & 3) == i].sum() for i in range(k)])
        assert np.allclose(db, Gm.sum(0)), "db2 small-head form"
    return True


# ---------------------------------------------------------------------------------------------------------------------------------
# The 16-Gaussian forms of the forward kernel (csrc/deform_fwd16.h): one wave = 16 Gaussians on v_mfma_f32_16x16x4_f32
# (A[i][k] in lane i+16k, B[k][n] in lane n+16k, D[row][n] in lane n+16q register r with row = 4q + r).  Lane = (n, q).
# "IL16" activation layout with T = W/16 tiles: tile t, register r of lane (n, q) holds feature T*(4q+r) + t of Gaussian n.
N16 = LANES & 15
Q16 = LANES >> 4


def mfma16(a, b, c):
    """a, b: [64]; c: [64,4] -> new c."""
    A = a.reshape(4, 16).T          # A[i][k]
    B = b.reshape(4, 16)            # B[k][n]
    D = A @ B                       # [16 rows][16 n]
    out = c.copy()
    for r in range(4):
        for q in range(4):
            out[16 * q:16 * q + 16, r] += D[4 * q + r, :]
    return out


def matrix_to_il16(M, T):
    return [np.stack([M[N16, T * (4 * Q16 + r) + t] for r in range(4)], 1) for t in range(T)]


def il16_to_matrix(X, T):
    M = np.zeros((16, 16 * T))
    for t in range(T):
        for r in range(4):
            for lane in range(64):
                M[lane & 15, T * (4 * (lane >> 4) + r) + t] = X[t][lane, r]
    return M


def feat16(Fm):
    """[16, F] -> per 16-feature group u: registers c<4 of lane (n,q) hold features 16u + 4q + c."""
    return [np.stack([Fm[N16, 16 * u + 4 * Q16 + c] for c in range(4)], 1) for u in range(Fm.shape[1] // 16)]


def trunk16(W0, b0, feat, OT):
    Y = []
    for ot in range(OT):
        Y.append(mfma16(np.where(Q16 == 0, b0[OT * N16 + ot], 0.0), np.ones(64), np.zeros((64, 4))))
    for u in range(len(feat)):
        for c in range(4):
            for ot in range(OT):
                Y[ot] = mfma16(W0[OT * N16 + ot, 16 * u + 4 * Q16 + c], feat[u][:, c], Y[ot])
    return Y


def dense_il16(Wm, bias, X, KT, OT):
    Y = [mfma16(np.where(Q16 == 0, bias[OT * N16 + ot], 0.0), np.ones(64), np.zeros((64, 4))) for ot in range(OT)]
    for r in range(4):
        for t in range(KT):
            for ot in range(OT):
                Y[ot] = mfma16(Wm[OT * N16 + ot, KT * (4 * Q16 + r) + t], X[t][:, r], Y[ot])
    return Y


def small_head16(W2, b2, k, X, KT):
    acc = np.zeros((64, 4))
    row = np.minimum(LANES & 3, k - 1)
    for r in range(4):
        for t in range(KT):
            acc = mfma4(W2[row, KT * (4 * Q16 + r) + t], X[t][:, r], acc)
    x = acc + acc[LANES ^ 16]
    x = x + x[LANES ^ 32]
    return x + b2[np.minimum(np.arange(4), k - 1)][None, :]


def tall_head16(W2, b2, k, X, KT):
    """k = 16 * NT output rows in standard order: returns [16, k]."""
    NT = (k + 15) // 16
    out = np.zeros((16, 16 * NT))
    for ot in range(NT):
        rows = np.minimum(16 * ot + N16, k - 1)
        Y = mfma16(np.where(Q16 == 0, b2[rows], 0.0), np.ones(64), np.zeros((64, 4)))
        for r in range(4):
            for t in range(KT):
                Y = mfma16(W2[rows, KT * (4 * Q16 + r) + t], X[t][:, r], Y)
        for r in range(4):
            for lane in range(64):
                out[lane & 15, 16 * ot + 4 * (lane >> 4) + r] = Y[lane, r]
    return out[:, :k]


def hmask_words_from_rows(rh_rows, W):
    """What D2 reads (sv_hmask): for the 32-Gaussian tile with relu(hidden) rows `rh_rows` [32, W]: lane (g, h) word t bit r =
    rh[g, T32 * rho(r, h) + t] > 0 with T32 = W / 32."""
    T32 = W // 32
    words = np.zeros((64, 4), np.uint32)
    for lane in range(64):
        g, h = lane & 31, lane >> 5
        for t in range(T32):
            for r in range(16):
                if rh_rows[g, T32 * rho(r, h) + t] > 0:
                    words[lane, t] |= np.uint32(1 << r)
    return words


def check16(W, F, k, rng):
    T = W // 16
    feat = rng.standard_normal((16, F))
    W0, b0 = rng.standard_normal((W, F)), rng.standard_normal(W)
    W1, b1 = rng.standard_normal((W, W)), rng.standard_normal(W)
    W2, b2 = rng.standard_normal((k, W)), rng.standard_normal(k)
    hid_ref = np.maximum(feat @ W0.T + b0, 0)
    h1_ref = np.maximum(hid_ref @ W1.T + b1, 0)
    out_ref = h1_ref @ W2.T + b2
    hid = [np.maximum(x, 0) for x in trunk16(W0, b0, feat16(feat), T)]
    assert np.allclose(il16_to_matrix(hid, T), hid_ref), "trunk16"
    assert np.allclose(il16_to_matrix(matrix_to_il16(hid_ref, T), T), hid_ref)
    h1 = [np.maximum(x, 0) for x in dense_il16(W1, b1, hid, T, T)]
    assert np.allclose(il16_to_matrix(h1, T), h1_ref), "L1 16"
    if k <= 4:
        o4 = small_head16(W2, b2, k, h1, T)
        for q in range(4):
            assert np.allclose(o4[16 * q:16 * q + 16, :k], out_ref), "small head 16"
    else:
        assert np.allclose(tall_head16(W2, b2, k, h1, T), out_ref), "tall head 16"
    return True


if __name__ == "__main__":
    for W, F, k in [(128, 32, 3), (128, 32, 48), (64, 64, 4), (128, 48, 1), (64, 32, 48)]:
        check16(W, F, k, np.random.default_rng(1))
        print(f"16-Gaussian forms W={W} F={F} k={k}: layouts consistent")
    rng = np.random.default_rng(0)
    for WT, FCH, k in [(4, 4, 3), (4, 4, 48), (4, 6, 4), (2, 8, 1), (2, 16, 48), (4, 12, 3)]:
        check(WT, FCH, k, rng)
        print(f"WT={WT} FCH={FCH} k={k}: layouts consistent")


# ---------------------------------------------------------------------------------------------------------------------------------------------
