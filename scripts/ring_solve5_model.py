"""Lane-level NumPy model of k_ring_solve5 (cnmf_e_amd/csrc/ring_solve.hpp): one 64-lane wave solves the bordered
ridge system of one pixel with the trailing matrix held as transposed 16x16 tiles in the fp64 MFMA accumulator
layout.  Every register is an array over the 64 lanes, every LDS exchange an explicit array, the MFMA is
`acc += X1^T X2` on the accumulator layout -- so the index algebra of the kernel (tile roles, operand feeds,
partial-sum exchanges of the substitutions) is checked against numpy.linalg.solve without a GPU.

    python scripts/ring_solve5_model.py
"""
import numpy as np

L = np.arange(64)
C, RQ = L & 15, L >> 4
DS = 18


def tix(i, j):
    return i * (i + 1) // 2 + j


def to_mat(regs):
    """accumulator layout -> 16x16: lane (c, rq), register r holds M[rq + 4r][c]"""
    M = np.zeros((16, 16))
    for r in range(4):
        M[RQ + 4 * r, C] = regs[r]
    return M


def from_mat(M):
    return np.stack([M[RQ + 4 * r, C] for r in range(4)])


def mfma(xa, xb, acc):
    """four v_mfma_f64_16x16x4 calls, call r fed with register r of both operands: acc += Xa^T Xb"""
    return from_mat(to_mat(acc) + to_mat(xa).T @ to_mat(xb))


def diag_block(sblk):
    """R-layout (lane = row) Cholesky + in-place inverse, readlane broadcasts; returns rows of inv(L)"""
    i = np.arange(16)
    a = np.array([[sblk[ii * DS + c] for c in range(16)] for ii in range(16)])   # a[lane][c]
    mydinv = np.zeros(16)
    for j in range(16):
        piv = a[j, j]
        inv = 1.0 / np.sqrt(piv)
        a[:, j] *= inv
        mydinv = np.where(i == j, inv, mydinv)
        for c in range(j + 1, 16):
            lcj = a[c, j]
            a[:, c] = a[:, c] - a[:, j] * lcj
    for j in range(16):
        dj = mydinv[j]
        lij = np.where(i > j, a[:, j] * dj, 0.0)
        for cc in range(j):
            ejc = a[j, cc]
            a[:, cc] = a[:, cc] - lij * ejc
        a[:, j] = np.where(i > j, -lij, a[:, j])
    out = np.zeros((16, 16))
    for cc in range(16):
        out[:, cc] = np.where(i > cc, a[:, cc] * mydinv, np.where(i == cc, mydinv, 0.0))
    return out


def solve_pixel(G, u, g, s, Tp, p):
    """G: p x p Gram of the real ring rows (identity rows for missing neighbours already in place), u, g: p vectors"""
    NT = (p + 15) // 16
    N = 16 * NT
    Gp = np.eye(N); Gp[:p, :p] = G
    up = np.zeros(N); up[:p] = u
    gp = np.zeros(N); gp[:p] = g
    # gather: tile (I, J), lane (c, rq), reg r = G[16 I + c][16 J + rq + 4 r]
    T = {}
    for I in range(NT):
        for J in range(I + 1):
            T[tix(I, J)] = np.stack([Gp[16 * I + C, 16 * J + RQ + 4 * r] for r in range(4)])
    # factorisation
    for k in range(NT):
        sblk = np.zeros(16 * DS)
        for r in range(4):
            sblk[(RQ + 4 * r) * DS + C] = T[tix(k, k)][r]
        Linv = diag_block(sblk)
        sblk2 = np.zeros(16 * DS)
        for ii in range(16):
            sblk2[ii * DS:ii * DS + 16] = Linv[ii]
        X1 = np.stack([sblk2[C * DS + RQ + 4 * r] for r in range(4)])
        for i in range(k + 1, NT):
            T[tix(i, k)] = mfma(X1, T[tix(i, k)], np.zeros((4, 64)))
        for j in range(k + 1, NT):
            nP = -T[tix(j, k)]
            for i in range(j, NT):
                T[tix(i, j)] = mfma(nP, T[tix(i, k)], T[tix(i, j)])
        T[tix(k, k)] = X1
    # forward substitution of the two border vectors
    vec = [up.copy(), gp.copy()]
    pb = [[np.zeros(64) for _ in range(NT)] for _ in range(2)]
    for k in range(NT):
        zq = []
        for v in range(2):
            part = np.where(RQ == 0, vec[v][16 * k + C], 0.0) - pb[v][k]
            spart = np.zeros(64); spart[C * 4 + RQ] = part
            bq = [sum(spart[(RQ + 4 * r) * 4 + q] for q in range(4)) for r in range(4)]
            p2 = sum(T[tix(k, k)][r] * bq[r] for r in range(4))
            spart2 = np.zeros(64); spart2[C * 4 + RQ] = p2
            zq.append([sum(spart2[(RQ + 4 * r) * 4 + q] for q in range(4)) for r in range(4)])
            zc = sum(spart2[C * 4 + q] for q in range(4))
            vec[v][16 * k + C] = zc
        for i in range(k + 1, NT):
            for v in range(2):
                pb[v][i] = pb[v][i] + sum(T[tix(i, k)][r] * zq[v][r] for r in range(4))
    zu, zg = vec
    return T, zu, zg, NT, N


def finish(T, zu, zg, NT, N, s, Tp, lam):
    sigma = Tp + lam - zu @ zu
    w0 = (s - zu @ zg) / sigma
    y = zg - w0 * zu
    wc = [None] * NT
    for k in range(NT - 1, -1, -1):
        acc = [np.zeros(64) for _ in range(4)]
        for i in range(k + 1, NT):
            for r in range(4):
                acc[r] = acc[r] + T[tix(i, k)][r] * wc[i]
        sblk = np.zeros(16 * DS)
        for r in range(4):
            sblk[(RQ + 4 * r) * DS + C] = acc[r]
        yk = y[16 * k + C] - sum(sblk[C * DS + cc] for cc in range(16))
        for r in range(4):
            sblk[(RQ + 4 * r) * DS + C] = T[tix(k, k)][r] * yk
        wc[k] = sum(sblk[C * DS + cc] for cc in range(16))
    w = np.zeros(N)
    for k in range(NT):
        w[16 * k + C] = wc[k]
    return w, w0


def main():
    rng = np.random.default_rng(0)
    for p in (96, 40, 116, 16, 7):
        Tn = 300
        X = rng.standard_normal((p, Tn)) + 0.3 * rng.standard_normal((1, Tn))
        yv = rng.standard_normal(Tn)
        missing = rng.random(p) < 0.1
        G = X @ X.T
        u = X.sum(1); g = X @ yv; s = yv.sum()
        G[missing, :] = 0; G[:, missing] = 0; G[missing, missing] = 1.0
        u[missing] = 0; g[missing] = 0
        tr = np.trace(G[~missing][:, ~missing]) + Tn
        lam = 1e-5 * tr
        Gr = G.copy(); Gr[~missing, ~missing] += lam
        # reference: the full bordered system
        n = p + 1
        S = np.zeros((n, n)); S[:p, :p] = Gr; S[:p, p] = u; S[p, :p] = u; S[p, p] = Tn + lam
        rhs = np.concatenate([g, [s]])
        ref = np.linalg.solve(S, rhs)
        T, zu, zg, NT, N = solve_pixel(Gr, u, g, s, Tn, p)
        w, w0 = finish(T, zu, zg, NT, N, s, Tn, lam)
        err = np.abs(w[:p] - ref[:p]).max() / np.abs(ref[:p]).max()
        print("p=%3d NT=%d  max rel err %.2e  w0 err %.2e  pad %.1e" % (p, NT, err, abs(w0 - ref[p]), np.abs(w[p:]).max() if N > p else 0))
        assert err < 1e-9


if __name__ == "__main__":
    main()
