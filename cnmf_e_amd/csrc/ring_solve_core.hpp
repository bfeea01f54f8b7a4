// B2b (core): the per-pixel ridge solve of fit_ring_model.m:92-108 -- ONE 64-lane wave per pixel, the normal equations in registers.
//
//   [G + lam I   u     ] [w ]   [g]        G(a,b) = sum_t Bf(ring a, t) Bf(ring b, t)   (block-pair covariance table)
//   [u'       Tp + lam ] [w0] = [s]        u(a) = sum_t Bf(ring a, t),  g(a) = Cov(ring a, centre),  s = sum_t Bf(centre, t)
//
// The p x p core (p <= 16 NT) lives as NT(NT+1)/2 transposed 16x16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64
// (lane (c, rq) = (l & 15, l >> 4), register r holds M[rq + 4r][c]).  Feeding such registers of X1 as the A operand and of X2 as the B
// operand of the four calls r = 0..3 computes X1' X2 in the same layout, so a block right-looking Cholesky closes on itself:
//   tile (i, j) holds C_ij' (C = trailing matrix);  X1_k = inv(L_kk)' comes out of a 16x16 diagonal step (lane = row, readlane
//   broadcasts, through 2.3 KB of LDS);  P_ik = L_ik' = X1_k' C_ik' (4 MFMAs);  C_ij' -= P_jk' P_ik (4 MFMAs) -- no operand ever
//   changes lanes.  The border (ones row, fit_ring_model.m:101, and the right-hand side :104) is eliminated by two forward
//   substitutions on the vector pipe, w0 by the Schur complement, w by a back substitution; both substitutions contract over a lane
//   index and exchange 16-value partial sums through LDS.  No barriers (a workgroup is one wave), two waves per SIMD for NT <= 6.
// Index algebra checked lane by lane in scripts/ring_solve5_model.py.
#pragma once
#include <hip/hip_runtime.h>

namespace cnmfe {

typedef double double4_t __attribute__((ext_vector_type(4)));
// 1/sqrt(x) in fp64: hardware seed (v_rsq_f64, ~2^-26) + two Newton steps
__device__ __forceinline__ double rs_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// A workgroup of ONE wave exchanges through LDS behind __syncthreads() (no barrier instruction is emitted for it); the four waves of a shared-diagonal workgroup
// (round 5, rs_factor4_step) exchange per wave: LDS operations of one wave execute in order, so only the COMPILER must be kept from moving them across the point.
template <int WG> __device__ __forceinline__ void rs_sync() {
    if constexpr (WG == 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
}

constexpr int RS_DS = 18;                 // row stride (doubles) of the 16x16 LDS exchange tile: conflict-free b128 row reads

__host__ __device__ constexpr int rs_tix(int i, int j) { return (i * (i + 1)) / 2 + j; }   // i >= j

__device__ __forceinline__ double4_t rs_mfma4(const double4_t &xa, const double4_t &xb, double4_t acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[r], xb[r], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ double rs_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// fp64 row broadcast inside each 16-lane DPP row: lane N of the row -> all 16 lanes (the only DPP control 64-bit operands take)
// (inline asm like the FMAs below, with its own wait states: the source may have been written by one of THEIR asm statements one or two
// instructions earlier, which the compiler's DPP hazard check does not see)
template <int N> __device__ __forceinline__ double rs_bc(double v) {
    double o;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(o) : "v"(v), "n"(N));
    return o;
}
// acc += row_newbcast<N>(src) * (-mul) as ONE instruction (v_fmac_f64 with a DPP source; hipcc only forms mov_dpp + fma).  The hazard
// "VALU writes a VGPR, a DPP operand reads it within 2 wait states" is not tracked through inline asm: NOP = true puts the wait states
// in front (first instruction of a run whose DPP source may just have been written by compiler-scheduled code).
template <int N, bool NOP> __device__ __forceinline__ void rs_fmac_bc(double &acc, const double &src, const double &mul) {
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
    else asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
}
template <int N, bool NOP> __device__ __forceinline__ void rs_fmac_bc_self(double &acc, const double &mul) {     // the DPP source is the accumulator itself
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mul), "n"(N));
    else asm volatile("v_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mul), "n"(N));
}
template <int J, int C> __device__ __forceinline__ void rs_chol_col(double (&a)[16]) {       // a[c] -= L(i, J) L(c, J) for c = C .. 15
    if constexpr (C < 16) { rs_fmac_bc<C, C == J + 1>(a[C], a[J], a[J]); rs_chol_col<J, C + 1>(a); }
}
template <int J, int CC> __device__ __forceinline__ void rs_inv_row(double (&a)[16], const double &lij) {   // E(i, cc) -= lij E(J, cc) for cc = CC .. J-1
    if constexpr (CC < J) { rs_fmac_bc_self<J, CC == 0>(a[CC], lij); rs_inv_row<J, CC + 1>(a, lij); }
}
// `inv` = 1 / sqrt(pivot J), handed in by the previous column: the NEXT pivot is final as soon as column J's first update (C = J + 1) has landed, so its
// broadcast and reciprocal square root (v_rsq_f64 + two Newton steps: a ~70-clock dependent chain) are started there and run under the remaining 14 - J updates
// of column J instead of in front of column J + 1.  Same operations on the same values: bit-identical factors.
template <int J> __device__ __forceinline__ void rs_chol(double (&a)[16], double &mydinv, int i, double inv) {
    if constexpr (J < 16) {
        a[J] *= inv;                                        // L(i, J), i >= J
        mydinv = i == J ? inv : mydinv;
        double inv_next = 0.0;
        if constexpr (J + 1 < 16) {
            rs_fmac_bc<J + 1, true>(a[J + 1], a[J], a[J]);
            inv_next = rs_rsqrt(rs_bc<J + 1>(a[J + 1]));
            rs_chol_col<J, J + 2>(a);
        }
        rs_chol<J + 1>(a, mydinv, i, inv_next);
    }
}
// in place: E(i, c) = -sum_{j = c .. i-1} L(i, j) Y(j, c),  Y(j, c) = E(j, c) / L(j, j),  Y(j, j) = 1 / L(j, j)
template <int J> __device__ __forceinline__ void rs_inv(double (&a)[16], const double &mydinv, int i) {
    if constexpr (J < 16) {
        const double dj = rs_bc<J>(mydinv);                 // (its own statement: inside the conditional below it would run under an exec mask that switches lane J off)
        const double lij = i > J ? a[J] * dj : 0.0;
        rs_inv_row<J, 0>(a, lij);
        a[J] = i > J ? -lij : a[J];
        asm volatile("s_nop 0" : "+v"(a[J]));                // (round 5, scripts/isa_stats.py: the scheduler may sink this select to right in front of the DPP read of a[J] in step J + 1)
        rs_inv<J + 1>(a, mydinv, i);
    }
}
// 16x16 diagonal step: sb holds the (symmetric) block, row i at sb[i * RS_DS]; on return it holds inv(L), L = chol(block), lower
// triangular with zeros above the diagonal.  Lane = row, the four 16-lane DPP rows of the wave work in replica (lane l: row l & 15);
// a column / row broadcast is the DPP source of the consuming fp64 FMA: 120 + 120 FMAs carry the whole step.
__device__ __forceinline__ void rs_diag_block(double *sb, int lane) {
    const int i = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) { const double2 v = *reinterpret_cast<const double2 *>(sb + i * RS_DS + c); a[c] = v.x; a[c + 1] = v.y; }
    double mydinv = 0.0;
    rs_chol<0>(a, mydinv, i, rs_rsqrt(rs_bc<0>(a[0])));
    rs_inv<0>(a, mydinv, i);
    __syncthreads();                                        // every lane has read its row
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            double2 v;
            v.x = i > c ? a[c] * mydinv : (i == c ? mydinv : 0.0);
            v.y = i > c + 1 ? a[c + 1] * mydinv : (i == c + 1 ? mydinv : 0.0);
            *reinterpret_cast<double2 *>(sb + i * RS_DS + c) = v;
        }
    }
    __syncthreads();
}

// ---- round 5: the diagonal step with the inversion UNDER the pivot chain ----
// rs_chol is a chain of 16 pivots (update -> broadcast -> v_rsq_f64 + two Newton steps -> scale: ~100 clocks each, of which the 15 - J updates of column J fill
// less and less), and rs_inv behind it a second chain of 16 steps whose J-th needs only column J of L and the rows of the inverse above J.  Fused, step J of the
// inversion (J updates on a[0 .. J-1]) runs in the shadow of pivot J + 1's reciprocal square root, where column J's own updates (14 - J of them, on a[J+2 ..])
// run out: every column issues ~17 updates whatever J is.  Same operations on the same values in the same per-register order: bit-identical to rs_chol + rs_inv.
// `hk.at<H>()`, H = 0 .. 31, is called twice per column: the look-ahead factorisation below issues the previous block column's trailing MFMAs there.
struct RsNoHook { template <int H> __device__ __forceinline__ void at() {} };
template <int J, class Hook> __device__ __forceinline__ void rs_cholinv(double (&a)[16], double &mydinv, int i, double inv, Hook &hk) {
    if constexpr (J < 16) {
        a[J] *= inv;                                        // L(i, J), i >= J
        mydinv = i == J ? inv : mydinv;
        double inv_next = 0.0;
        if constexpr (J + 1 < 16) {
            rs_fmac_bc<J + 1, true>(a[J + 1], a[J], a[J]);
            inv_next = rs_rsqrt(rs_bc<J + 1>(a[J + 1]));
        }
        hk.template at<2 * J>();
        if constexpr (J + 1 < 16) rs_chol_col<J, J + 2>(a);
        const double lij = i > J ? a[J] * inv : 0.0;        // (inv == the broadcast of lane J's mydinv that rs_inv takes)
        hk.template at<2 * J + 1>();
        if constexpr (J < 15) rs_inv_row<J, 0>(a, lij);     // (J = 15: lij = 0 in every lane)
        a[J] = i > J ? -lij : a[J];
        // a[J] is next read THROUGH DPP (rs_inv_row<J + 1, J>, no wait states of its own) a whole column later: without this pin the scheduler sank the select to
        // right in front of that read (found in the ISA of the NT = 6 instantiation by scripts/isa_stats.py: lane J + 1's read returned the old a[J])
        asm volatile("s_nop 0" : "+v"(a[J]));
        rs_cholinv<J + 1>(a, mydinv, i, inv_next, hk);
    }
}
template <class Hook> __device__ __forceinline__ void rs_diag_block2(double *sb, int lane, Hook &hk) {
    const int i = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) { const double2 v = *reinterpret_cast<const double2 *>(sb + i * RS_DS + c); a[c] = v.x; a[c + 1] = v.y; }
    double mydinv = 0.0;
    rs_cholinv<0>(a, mydinv, i, rs_rsqrt(rs_bc<0>(a[0])), hk);
    __syncthreads();                                        // every lane has read its row
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            double2 v;
            v.x = i > c ? a[c] * mydinv : (i == c ? mydinv : 0.0);
            v.y = i > c + 1 ? a[c + 1] * mydinv : (i == c + 1 ? mydinv : 0.0);
            *reinterpret_cast<double2 *>(sb + i * RS_DS + c) = v;
        }
    }
    __syncthreads();
}

// one block column of the factorisation: TRSM of the panel, rank-16 update of the trailing tiles, inv(L_kk)' parked in the diagonal slot
template <int NT, int K>
__device__ __forceinline__ void rs_step(double4_t (&T)[(NT * (NT + 1)) / 2], const double4_t &X1) {
#pragma unroll
    for (int i = K + 1; i < NT; ++i) T[rs_tix(i, K)] = rs_mfma4(X1, T[rs_tix(i, K)], (double4_t){0.0, 0.0, 0.0, 0.0});
#pragma unroll
    for (int j = K + 1; j < NT; ++j) {
        const double4_t nP = -T[rs_tix(j, K)];
#pragma unroll
        for (int i = j; i < NT; ++i) T[rs_tix(i, j)] = rs_mfma4(nP, T[rs_tix(i, K)], T[rs_tix(i, j)]);
    }
    T[rs_tix(K, K)] = X1;
}

__device__ __forceinline__ void rs_put_diag(const double4_t &D, double *s_blk, int c, int rq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_blk[(rq + 4 * r) * RS_DS + c] = D[r];
}
// the block columns are unrolled, every tile index a compile-time constant (the NT = 6 kernel is 87 KB of code; a looped variant with the tile
// indices in an if-chain spilled ~200 tile registers: 10.6 ms against 8.5 ms, removed in round 3)
template <int NT, int K>
__device__ __forceinline__ void rs_factor_unrolled(double4_t (&T)[(NT * (NT + 1)) / 2], double *s_blk, int lane, int c, int rq) {
    if constexpr (K < NT) {
        __syncthreads();
        rs_diag_block(s_blk, lane);
        double4_t X1;
#pragma unroll
        for (int r = 0; r < 4; ++r) X1[r] = s_blk[c * RS_DS + rq + 4 * r];
        __syncthreads();
        rs_step<NT, K>(T, X1);
        if constexpr (K + 1 < NT) rs_put_diag(T[rs_tix(K + 1, K + 1)], s_blk, c, rq);
        rs_factor_unrolled<NT, K + 1>(T, s_blk, lane, c, rq);
    }
}
template <int NT>
__device__ __forceinline__ void rs_factor(double4_t (&T)[(NT * (NT + 1)) / 2], double *s_blk, int lane, int c, int rq) {
    rs_put_diag(T[0], s_blk, c, rq);
    rs_factor_unrolled<NT, 0>(T, s_blk, lane, c, rq);
}


// ---- round 5: look-ahead.  Diagonal step K + 1 needs ONE tile of block column K's trailing update -- (K+1, K+1) -- and is a ~2200-clock chain on the vector
// pipe; the other TRSMs and rank-16 updates of column K (72 / 48 / 28 / 12 MFMAs of 64 clocks at NT = 6) do not depend on it.  An in-order wave overlaps the two
// only if they alternate in program order, so the diagonal step calls a hook twice per column (rs_cholinv) and the hook issues the next MFMAs of the column's
// remainder there.  Every MFMA is pinned between two `s_nop 0` asm statements on its accumulator (volatile asm keeps its order against the DPP statements of the
// diagonal step; a NON-empty pin because the hazard recogniser counts an asm statement as one wait state whatever it holds).  Same products, same accumulation
// order per tile: bit-identical to rs_factor.
struct RsIJ { int i, j; bool first; };
template <int NT, int K> __host__ __device__ constexpr RsIJ rs_upd_ij(int u) {          // u-th trailing tile of block column K in rs_step's order, (K+1, K+1) left out
    int idx = 0;
    for (int j = K + 1; j < NT; ++j) {
        bool first = true;
        for (int i = j; i < NT; ++i) {
            if (i == K + 1 && j == K + 1) continue;
            if (idx == u) return RsIJ{i, j, first};
            first = false; ++idx;
        }
    }
    return RsIJ{-1, -1, false};
}
__device__ __forceinline__ void rs_pin(double4_t &x) { asm volatile("s_nop 0" : "+v"(x)); }
template <int NT, int K> struct RsLook {
    static constexpr int n = NT - K - 1;                                    // tiles of the panel below the diagonal
    static constexpr int NTRSM = n - 1, NUPD = (n * (n + 1)) / 2 - 1, NOPS = 4 * (NTRSM + NUPD), NHOOK = 32;
    double4_t (&T)[(NT * (NT + 1)) / 2];
    const double4_t &X1;
    double4_t tmp, nP;
    __device__ __forceinline__ RsLook(double4_t (&T_)[(NT * (NT + 1)) / 2], const double4_t &X1_) : T(T_), X1(X1_) {}
    template <int Q> __device__ __forceinline__ void op() {
        constexpr int t = Q / 4, r = Q % 4;
        if constexpr (t < NTRSM) {                                          // P_iK = X1' C_iK', rows K + 2 ..
            constexpr int i = K + 2 + t;
            if constexpr (r == 0) { tmp = (double4_t){0.0, 0.0, 0.0, 0.0}; rs_pin(tmp); }     // (the pin behind call r is the pin in front of call r + 1)
            tmp = __builtin_amdgcn_mfma_f64_16x16x4f64(X1[r], T[rs_tix(i, K)][r], tmp, 0, 0, 0);
            rs_pin(tmp);
            if constexpr (r == 3) T[rs_tix(i, K)] = tmp;
        } else {
            constexpr RsIJ e = rs_upd_ij<NT, K>(t - NTRSM);
            if constexpr (r == 0 && e.first) nP = -T[rs_tix(e.j, K)];
            if constexpr (r == 0) rs_pin(T[rs_tix(e.i, e.j)]);
            T[rs_tix(e.i, e.j)] = __builtin_amdgcn_mfma_f64_16x16x4f64(nP[r], T[rs_tix(e.i, K)][r], T[rs_tix(e.i, e.j)], 0, 0, 0);
            rs_pin(T[rs_tix(e.i, e.j)]);
        }
    }
    template <int Q, int QE> __device__ __forceinline__ void run() {
        if constexpr (Q < QE) { op<Q>(); run<Q + 1, QE>(); }
    }
    template <int H> __device__ __forceinline__ void at() { run<(H * NOPS) / NHOOK, ((H + 1) * NOPS) / NHOOK>(); }
    __device__ __forceinline__ void run_all() { run<0, NOPS>(); }
};
template <int NT, int K>
__device__ __forceinline__ void rs_factor_la_step(double4_t (&T)[(NT * (NT + 1)) / 2], double *s_blk, int lane, int c, int rq, const double4_t &X1) {
    if constexpr (K + 1 < NT) {
        T[rs_tix(K + 1, K)] = rs_mfma4(X1, T[rs_tix(K + 1, K)], (double4_t){0.0, 0.0, 0.0, 0.0});
        {
            const double4_t nP = -T[rs_tix(K + 1, K)];
            T[rs_tix(K + 1, K + 1)] = rs_mfma4(nP, T[rs_tix(K + 1, K)], T[rs_tix(K + 1, K + 1)]);
        }
        rs_put_diag(T[rs_tix(K + 1, K + 1)], s_blk, c, rq);
        __syncthreads();
        RsLook<NT, K> hk(T, X1);
        rs_diag_block2(s_blk, lane, hk);
        double4_t X1n;
#pragma unroll
        for (int r = 0; r < 4; ++r) X1n[r] = s_blk[c * RS_DS + rq + 4 * r];
        __syncthreads();
        T[rs_tix(K, K)] = X1;
        rs_factor_la_step<NT, K + 1>(T, s_blk, lane, c, rq, X1n);
    } else {
        T[rs_tix(K, K)] = X1;
    }
}
template <int NT, int K>
__device__ __forceinline__ void rs_factor_fused(double4_t (&T)[(NT * (NT + 1)) / 2], double *s_blk, int lane, int c, int rq, const double4_t &X1) {
    rs_step<NT, K>(T, X1);
    if constexpr (K + 1 < NT) {
        rs_put_diag(T[rs_tix(K + 1, K + 1)], s_blk, c, rq);
        __syncthreads();
        RsNoHook nh;
        rs_diag_block2(s_blk, lane, nh);
        double4_t X1n;
#pragma unroll
        for (int r = 0; r < 4; ++r) X1n[r] = s_blk[c * RS_DS + rq + 4 * r];
        __syncthreads();
        rs_factor_fused<NT, K + 1>(T, s_blk, lane, c, rq, X1n);
    }
}
// ---- round 5: SHARED diagonal steps.  The diagonal step works in 16-lane DPP rows and a one-wave workgroup runs it four times in replica -- 3300 of the solve's
// 6082 vector instructions per pixel, on the pipe that bounds it (profiles/r04/pmc_pipes_v4.txt: vector pipe 50 % busy, matrix pipe 24 %).  A workgroup of FOUR
// waves = four pixels hands the four diagonal tiles of a block column to ONE wave, DPP row pp = pixel pp: the same instruction stream factors and inverts four
// different tiles.  The duty rotates (wave K mod 4 takes column K: the waves sit on different SIMDs), two workgroup barriers per block column; with the look-ahead
// the other three waves issue their column's remaining MFMAs meanwhile.  sb4 + w * stride = the exchange tile of wave w.
template <class Hook> __device__ __forceinline__ void rs_diag_block4(double *sb4, int stride, int lane, Hook &hk) {
    const int i = lane & 15;
    double *sb = sb4 + (lane >> 4) * stride;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) { const double2 v = *reinterpret_cast<const double2 *>(sb + i * RS_DS + c); a[c] = v.x; a[c + 1] = v.y; }
    double mydinv = 0.0;
    rs_cholinv<0>(a, mydinv, i, rs_rsqrt(rs_bc<0>(a[0])), hk);
    rs_sync<4>();                                           // every lane of this wave has read its row
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        double2 v;
        v.x = i > c ? a[c] * mydinv : (i == c ? mydinv : 0.0);
        v.y = i > c + 1 ? a[c + 1] * mydinv : (i == c + 1 ? mydinv : 0.0);
        *reinterpret_cast<double2 *>(sb + i * RS_DS + c) = v;
    }
}
// block column K is factored (X1 = inv(L_KK)' of this wave's pixel): its panel and trailing update, then the shared diagonal step of column K + 1
template <int NT, int K, bool LA>
__device__ __forceinline__ void rs_factor4_step(double4_t (&T)[(NT * (NT + 1)) / 2], double *sb4, int stride, int w, int lane, int c, int rq, const double4_t &X1) {
    if constexpr (K + 1 < NT) {
        double *s_blk = sb4 + w * stride;
        if constexpr (LA) {
            T[rs_tix(K + 1, K)] = rs_mfma4(X1, T[rs_tix(K + 1, K)], (double4_t){0.0, 0.0, 0.0, 0.0});
            const double4_t nP = -T[rs_tix(K + 1, K)];
            T[rs_tix(K + 1, K + 1)] = rs_mfma4(nP, T[rs_tix(K + 1, K)], T[rs_tix(K + 1, K + 1)]);
        } else {
            rs_step<NT, K>(T, X1);
        }
        rs_put_diag(T[rs_tix(K + 1, K + 1)], s_blk, c, rq);
        __syncthreads();                                    // the four tiles of column K + 1 are in LDS
        if constexpr (LA) {
            RsLook<NT, K> hk(T, X1);
            if (w == ((K + 1) & 3)) rs_diag_block4(sb4, stride, lane, hk);
            else hk.run_all();
        } else {
            RsNoHook nh;
            if (w == ((K + 1) & 3)) rs_diag_block4(sb4, stride, lane, nh);
        }
        __syncthreads();                                    // ... and hold inv(L)
        double4_t X1n;
#pragma unroll
        for (int r = 0; r < 4; ++r) X1n[r] = s_blk[c * RS_DS + rq + 4 * r];
        rs_sync<4>();
        if constexpr (LA) T[rs_tix(K, K)] = X1;
        rs_factor4_step<NT, K + 1, LA>(T, sb4, stride, w, lane, c, rq, X1n);
    } else {
        if constexpr (LA) T[rs_tix(K, K)] = X1;
        else rs_step<NT, K>(T, X1);
    }
}
template <int NT, bool LA>
__device__ __forceinline__ void rs_factor4(double4_t (&T)[(NT * (NT + 1)) / 2], double *sb4, int stride, int w, int lane, int c, int rq) {
    double *s_blk = sb4 + w * stride;
    rs_put_diag(T[0], s_blk, c, rq);
    __syncthreads();
    RsNoHook nh;
    if (w == 0) rs_diag_block4(sb4, stride, lane, nh);
    __syncthreads();
    double4_t X1;
#pragma unroll
    for (int r = 0; r < 4; ++r) X1[r] = s_blk[c * RS_DS + rq + 4 * r];
    rs_sync<4>();
    rs_factor4_step<NT, 0, LA>(T, sb4, stride, w, lane, c, rq, X1);
}
// VAR 0: rs_factor as it was; 1: the fused diagonal step; 2: fused + look-ahead
template <int NT, int VAR>
__device__ __forceinline__ void rs_factor_var(double4_t (&T)[(NT * (NT + 1)) / 2], double *s_blk, int lane, int c, int rq) {
    if constexpr (VAR == 0) rs_factor<NT>(T, s_blk, lane, c, rq);
    else {
        rs_put_diag(T[0], s_blk, c, rq);
        __syncthreads();
        RsNoHook nh;
        rs_diag_block2(s_blk, lane, nh);
        double4_t X1;
#pragma unroll
        for (int r = 0; r < 4; ++r) X1[r] = s_blk[c * RS_DS + rq + 4 * r];
        __syncthreads();
        if constexpr (VAR == 2) rs_factor_la_step<NT, 0>(T, s_blk, lane, c, rq, X1);
        else rs_factor_fused<NT, 0>(T, s_blk, lane, c, rq, X1);
    }
}

// Everything behind the gather: factorisation, the two forward substitutions, Schur complement of the ones row, back substitution.
//   T       : the tiles of G + lam I (see the header comment), consumed
//   s_vec   : [3][16 NT] LDS; on entry s_vec[0] = u, s_vec[1] = g (zeros on missing / padding rows)
//   s_blk   : [16 * RS_DS] LDS, s_part : [4][64] LDS
//   sc, lam, Tp : s, the ridge, the number of frames (the ones row's own Gram entry)
//   wc[k]   : on return, w(16 k + c) in every lane with l & 15 == c
//   WG = 4 (shared diagonal steps): s_blk = this wave's exchange tile, the tile of wave v at s_blk + (v - w) * stride
template <int NT, int VAR = 0, int WG = 1>
__device__ __forceinline__ void rs_solve_core(double4_t (&T)[(NT * (NT + 1)) / 2], double (*s_vec)[16 * NT], double *s_blk, double (*s_part)[64],
                                              double sc, double lam, double Tp, int lane, int probe, double (&wc)[NT], int w = 0, int stride = 0) {
    constexpr int N = 16 * NT;
    const int c = lane & 15, rq = lane >> 4;
    // ---- block Cholesky ----
    if (!(probe & 2)) {
        if constexpr (WG == 4) rs_factor4<NT, VAR == 2>(T, s_blk - w * stride, stride, w, lane, c, rq);
        else rs_factor_var<NT, VAR>(T, s_blk, lane, c, rq);
    }
    if (probe & 4) {
#pragma unroll
        for (int k = 0; k < NT; ++k) wc[k] = T[rs_tix(k, k)][0];
        return;
    }
    // ---- forward substitution of u and g: z_k = inv(L_kk) (b_k - sum_{j<k} L_kj z_j), right-looking ----
    // registers now hold, for every tile (i, k): lane (c, rq), r -> M_ik[c][rq + 4r] with M_kk = inv(L_kk), M_ik = L_ik
    double pb[2][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { pb[0][i] = 0.0; pb[1][i] = 0.0; }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double4_t Mkk = T[rs_tix(k, k)];
#pragma unroll
        for (int v = 0; v < 2; ++v) s_part[v][c * 4 + rq] = (rq == 0 ? s_vec[v][16 * k + c] : 0.0) - pb[v][k];
        rs_sync<WG>();
        double p2[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 lo = *reinterpret_cast<const double2 *>(&s_part[v][(rq + 4 * r) * 4]);
                const double2 hi = *reinterpret_cast<const double2 *>(&s_part[v][(rq + 4 * r) * 4 + 2]);
                acc = fma(Mkk[r], (lo.x + lo.y) + (hi.x + hi.y), acc);
            }
            p2[v] = acc;
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) s_part[2 + v][c * 4 + rq] = p2[v];
        rs_sync<WG>();
        double zq[2][4];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 lo = *reinterpret_cast<const double2 *>(&s_part[2 + v][(rq + 4 * r) * 4]);
                const double2 hi = *reinterpret_cast<const double2 *>(&s_part[2 + v][(rq + 4 * r) * 4 + 2]);
                zq[v][r] = (lo.x + lo.y) + (hi.x + hi.y);
            }
            if (rq == 0) {
                const double2 lo = *reinterpret_cast<const double2 *>(&s_part[2 + v][c * 4]);
                const double2 hi = *reinterpret_cast<const double2 *>(&s_part[2 + v][c * 4 + 2]);
                s_vec[v][16 * k + c] = (lo.x + lo.y) + (hi.x + hi.y);
            }
        }
#pragma unroll
        for (int i = k + 1; i < NT; ++i)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                double acc = pb[v][i];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = fma(T[rs_tix(i, k)][r], zq[v][r], acc);
                pb[v][i] = acc;
            }
        rs_sync<WG>();
    }
    // ---- Schur complement of the ones row: w0, then y = z_g - w0 z_u ----
    double zuu = 0.0, zug = 0.0;
    for (int a = lane; a < N; a += 64) { const double zu = s_vec[0][a], zg = s_vec[1][a]; zuu = fma(zu, zu, zuu); zug = fma(zu, zg, zug); }
    zuu = rs_wave_sum(zuu); zug = rs_wave_sum(zug);
    const double w0 = (sc - zug) / (Tp + lam - zuu);
    for (int a = lane; a < N; a += 64) s_vec[2][a] = s_vec[1][a] - w0 * s_vec[0][a];
    rs_sync<WG>();
    // ---- back substitution L' w = y: w_k = inv(L_kk)' (y_k - sum_{i>k} L_ik' w_i); both products contract over the lane index c ----
#pragma unroll
    for (int k = NT - 1; k >= 0; --k) {
        double yk = s_vec[2][16 * k + c];
        if (k < NT - 1) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = k + 1; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fma(T[rs_tix(i, k)][r], wc[i], acc[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) s_blk[(rq + 4 * r) * RS_DS + c] = acc[r];
            rs_sync<WG>();
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int cc = 0; cc < 16; cc += 2) { const double2 v = *reinterpret_cast<const double2 *>(s_blk + c * RS_DS + cc); s0 += v.x; s1 += v.y; }
            yk -= s0 + s1;
            rs_sync<WG>();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_blk[(rq + 4 * r) * RS_DS + c] = T[rs_tix(k, k)][r] * yk;
        rs_sync<WG>();
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; cc += 2) { const double2 v = *reinterpret_cast<const double2 *>(s_blk + c * RS_DS + cc); s0 += v.x; s1 += v.y; }
        wc[k] = s0 + s1;
        rs_sync<WG>();
    }
}

}  // namespace cnmfe
