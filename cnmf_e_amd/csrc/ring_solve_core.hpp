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

// a wave-uniform double into scalar registers (values loaded through the vector memory path stay in VGPRs otherwise -- and this kernel has none to spare)
__device__ __forceinline__ double ri_uni(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int SRC> __device__ __forceinline__ double ri_readlane(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC), hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}
// reductions without the LDS crossbar (a __shfl_xor is a ds_bpermute: six dependent ~100-clock steps per wave sum, and this kernel takes a dozen of them):
// a butterfly inside each 16-lane DPP row -- quad_perm [1 0 3 2], [2 3 0 1], row_half_mirror, row_mirror on the two halves of the double -- leaves the row's
// total in all its lanes; the four row totals meet through readlanes
template <int CTRL> __device__ __forceinline__ double ri_dpp(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ri_row_sum(double v) {                // over the 16 lanes of a DPP row, in all of them
    v += ri_dpp<0xB1>(v); v += ri_dpp<0x4E>(v); v += ri_dpp<0x141>(v); v += ri_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ double ri_wave_sum(double v) {               // over the wave, uniform
    v = ri_row_sum(v);
    return (ri_readlane<0>(v) + ri_readlane<16>(v)) + (ri_readlane<32>(v) + ri_readlane<48>(v));
}
__device__ __forceinline__ double ri_wave_max(double v) {
    v = fmax(v, ri_dpp<0xB1>(v)); v = fmax(v, ri_dpp<0x4E>(v)); v = fmax(v, ri_dpp<0x141>(v)); v = fmax(v, ri_dpp<0x140>(v));
    return fmax(fmax(ri_readlane<0>(v), ri_readlane<16>(v)), fmax(ri_readlane<32>(v), ri_readlane<48>(v)));
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
        // Round 5: a[J] is next read THROUGH DPP -- rs_inv_row<J + 1, J>, an asm statement without wait states of its own -- and the scheduler is free to sink
        // this select to right in front of that read (VALU write -> DPP read needs two wait states; the compiler's hazard recogniser does not look into asm).
        // scripts/isa_stats.py found exactly that in the NT = 4, 5, 7, 8 instantiations (at J = 15, where the stale value is multiplied by lij = 0: harmless by
        // luck) and, in this round's fused variant of the step, at J = 14 (wrong weights).  The pin keeps the select in front of the next step's asm sequence.
        asm volatile("s_nop 0" : "+v"(a[J]));
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


// Everything behind the gather: factorisation, the two forward substitutions, Schur complement of the ones row, back substitution.
//   T       : the tiles of G + lam I (see the header comment), consumed
//   s_vec   : [3][16 NT] LDS; on entry s_vec[0] = u, s_vec[1] = g (zeros on missing / padding rows)
//   s_blk   : [16 * RS_DS] LDS, s_part : [4][64] LDS
//   sc, lam, Tp : s, the ridge, the number of frames (the ones row's own Gram entry)
//   wc[k]   : on return, w(16 k + c) in every lane with l & 15 == c
template <int NT>
__device__ __forceinline__ void rs_solve_core(double4_t (&T)[(NT * (NT + 1)) / 2], double (*s_vec)[16 * NT], double *s_blk, double (*s_part)[64],
                                              double sc, double lam, double Tp, int lane, int probe, double (&wc)[NT]) {
    constexpr int N = 16 * NT;
    const int c = lane & 15, rq = lane >> 4;
    // ---- block Cholesky ----
    if (!(probe & 2)) rs_factor<NT>(T, s_blk, lane, c, rq);
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
        __syncthreads();
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
        __syncthreads();
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
        __syncthreads();
    }
    // ---- Schur complement of the ones row: w0, then y = z_g - w0 z_u ----
    double zuu = 0.0, zug = 0.0;
    for (int a = lane; a < N; a += 64) { const double zu = s_vec[0][a], zg = s_vec[1][a]; zuu = fma(zu, zu, zuu); zug = fma(zu, zg, zug); }
    zuu = rs_wave_sum(zuu); zug = rs_wave_sum(zug);
    const double w0 = (sc - zug) / (Tp + lam - zuu);
    for (int a = lane; a < N; a += 64) s_vec[2][a] = s_vec[1][a] - w0 * s_vec[0][a];
    __syncthreads();
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
            __syncthreads();
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int cc = 0; cc < 16; cc += 2) { const double2 v = *reinterpret_cast<const double2 *>(s_blk + c * RS_DS + cc); s0 += v.x; s1 += v.y; }
            yk -= s0 + s1;
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_blk[(rq + 4 * r) * RS_DS + c] = T[rs_tix(k, k)][r] * yk;
        __syncthreads();
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; cc += 2) { const double2 v = *reinterpret_cast<const double2 *>(s_blk + c * RS_DS + cc); s0 += v.x; s1 += v.y; }
        wc[k] = s0 + s1;
        __syncthreads();
    }
}


}  // namespace cnmfe
