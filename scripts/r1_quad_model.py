"""Lane-level NumPy model of the R1 kernel with the four arc roles inside ONE wave (resid_quad.hpp): checks the symmetric arc program
(one canonical arc, the other three by 90-degree rotations with per-lane strides), the lane -> (role, strip) map built on the ds_read_b128
lane groups, the half-mirror pair combine and the wave-local exchange against a direct ring sum.     python scripts/r1_quad_model.py"""
import numpy as np

R = 15
TR = TC = 16
HR, HC = TR + 2 * R, TC + 2 * R
HRp = ((HR + 14) // 16) * 16 + 1
P = 4

ring = [(r, c) for c in range(-R, R + 1) for r in range(-R, R + 1) if R * R <= c * c + r * r < (R + 1) * (R + 1)]   # (dr, dc), find() order
ring_index = {o: i for i, o in enumerate(ring)}
# canonical arc: the left one, (mov, fix) = (dr, dc)
S0 = sorted([(dc, dr) for (dr, dc) in ring if dc < 0 and (abs(dc) > abs(dr) or (abs(dc) == abs(dr) and dr < 0))])   # sorted by (fix, mov)
NA = len(S0)
runs = []                                  # (fix, first mov, length, arc-local index of the first)
for a, (fix, mov) in enumerate(S0):
    if runs and runs[-1][0] == fix and runs[-1][1] + runs[-1][2] == mov:
        runs[-1][2] += 1
    else:
        runs.append([fix, mov, 1, a])
prog = []                                  # reads: (fix, movpos, [(j, a), ...])
for fix, rs, rl, a0 in runs:
    for x in range(rl + P - 1):
        prog.append((fix, rs + x, [(j, a0 + x - j) for j in range(P) if 0 <= x - j < rl]))
T_ROLE = [lambda m, f: (m, f), lambda m, f: (-m, -f), lambda m, f: (f, -m), lambda m, f: (-f, m)]    # canonical (mov, fix) -> (dr, dc)
assert sorted(T_ROLE[r](m, f) for r in range(4) for (f, m) in S0) == sorted(ring), "the four rotated arcs must tile the ring"
print("ring %d offsets, arc %d offsets in %d runs, %d reads per role and chunk" % (len(ring), NA, len(runs), len(prog)))

G0 = set(list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)))       # ds_read_b128 lane group 0 (MI355X_MICROARCH.md, LDS)


def lane_role_strip(l):
    half, q = l >> 5, l & 31
    par = 0 if q in G0 else 1
    return 2 * half + par, (l & 15) if par == 0 else ((l ^ 7) & 15)


def A(row, col):
    return (col + R) * HRp + (row + R)


def run_wave(w, halo, Wt):
    """halo: dict slot -> value (one frame); Wt[(row, col)][ring index] tile-local centre weights.  Returns {(row, col): ring sum} for the wave's block"""
    acc = np.zeros((64, P))
    cent = {}
    groups = {}
    for l in range(64):
        role, s = lane_role_strip(l)
        if role < 2:
            g, c = s >> 2, s & 3
            rb, cb = (4 * g, 4 * w + c) if role == 0 else (4 * g + 3, 4 * w + c)
        else:
            rb, cb = (s, 4 * w + 3) if role == 2 else (s, 4 * w)
        sM, sF = [(1, HRp), (-1, -HRp), (-HRp, 1), (HRp, -1)][role]
        base = A(rb, cb)
        cj = [[(rb + j, cb), (rb - j, cb), (rb, cb - j), (rb, cb + j)][role] for j in range(P)]
        cent[l] = cj
        for li, (fix, mp, feeds) in enumerate(prog):
            addr = base + fix * sF + mp * sM
            groups.setdefault((li, l >> 4 if False else None), None)
            v = halo[addr]
            for (j, a) in feeds:
                f, m = S0[a]
                dr, dc = T_ROLE[role](m, f)
                # the value read must be the neighbour of centre j at this offset
                assert addr == A(cj[j][0] + dr, cj[j][1] + dc), (role, j, a)
                acc[l, j] += Wt[cj[j]][ring_index[(dr, dc)]] * v
    # bank check: every read, every hardware lane group: 16 distinct slots mod 16
    HW = [sorted(G0), sorted(set(range(32)) - G0)]
    HW = HW + [[x + 32 for x in g] for g in HW]
    for li, (fix, mp, _) in enumerate(prog):
        for grp in HW:
            slots = []
            for l in grp:
                role, s = lane_role_strip(l)
                if role < 2:
                    g, c = s >> 2, s & 3
                    rb, cb = (4 * g, 4 * w + c) if role == 0 else (4 * g + 3, 4 * w + c)
                else:
                    rb, cb = (s, 4 * w + 3) if role == 2 else (s, 4 * w)
                sM, sF = [(1, HRp), (-1, -HRp), (-HRp, 1), (HRp, -1)][role]
                slots.append((A(rb, cb) + fix * sF + mp * sM) % 16)
            assert len(set(slots)) == 16, ("bank conflict", li, grp)
    # pair combine: parity-0 lanes add the partner's reversed accumulators (DPP row_half_mirror: l ^ 7)
    tot = np.zeros((64, P))
    for l in range(64):
        for j in range(P):
            tot[l, j] = acc[l, j] + acc[l ^ 7, P - 1 - j]
    V = np.zeros(64); H = np.zeros(64)
    for l in range(64):
        role, s = lane_role_strip(l)
        if role == 0:
            g, c = s >> 2, s & 3
            for j in range(P):
                V[c * 16 + 4 * g + j] = tot[l, j]
        elif role == 2:
            for j in range(P):
                H[(3 - j) * 16 + s] = tot[l, j]
    return {(l & 15, 4 * w + (l >> 4)): V[l] + H[l] for l in range(64)}


def main():
    rng = np.random.default_rng(0)
    halo = rng.standard_normal(HRp * HC + 64)
    Wt = {(r, c): rng.standard_normal(len(ring)) for r in range(TR) for c in range(TC)}
    for w in range(4):
        out = run_wave(w, halo, Wt)
        for (r, c), v in out.items():
            ref = sum(Wt[(r, c)][i] * halo[A(r + dr, c + dc)] for i, (dr, dc) in enumerate(ring))
            assert abs(v - ref) < 1e-9, (w, r, c, v, ref)
    print("4 waves x 64 centres: ring sums match; all ds_read_b128 lane groups conflict-free")


if __name__ == "__main__":
    main()
