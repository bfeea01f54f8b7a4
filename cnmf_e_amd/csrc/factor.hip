// S1-S4 / T1-T3 / S6: spatial HALS / NNLS, temporal HALS and the spatial post-processing on the
// resident Ysig (d x T fp32, frame-major) of one patch.
//
// Sparsity is the structure here: A is only ever non-zero on the search mask IND, so U = Ysig*C'
// (HALS_spatial.m:31) is evaluated on nnz(IND) entries instead of d*K, V = C*C' / A'*A only on the
// pairs of neurons that share a pixel, and the strictly sequential k = 1..K Gauss-Seidel order of
// HALS_spatial.m:37 / HALS_temporal.m:60 is reproduced exactly by a level schedule: neuron k waits
// for every smaller-index neuron it shares a pixel with; neurons of one level are independent.
#include "common.hpp"
#include <math.h>

namespace cnmfe {

// ---- S1: U(e) = sum_t Ysig(m_e, t) * Cc(k_e, t)   (Cc centred => equals Y*C' - T*Ymean*Cmean') ----
__global__ void __launch_bounds__(256) k_proj_spatial(const float4 *__restrict__ ysig4, int64_t d, int64_t T, const int *__restrict__ erow,
                                                      const int *__restrict__ ecol, int64_t nnz, const float *__restrict__ Cc, int64_t ldc,
                                                      int64_t tchunk, float *__restrict__ part) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int64_t t0 = (int64_t)blockIdx.y * tchunk, t1 = t0 + tchunk < T ? t0 + tchunk : T;      // tchunk is a multiple of 4
    const float4 *y = ysig4 + erow[e];
    const float *c = Cc + (int64_t)ecol[e] * ldc;            // centred traces are 0 beyond T, so padded frames drop out
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t t = t0; t < t1; t += 4) {
        const float4 yv = y[(t >> 2) * d];
        const float4 cv = *reinterpret_cast<const float4 *>(c + t);
        a0 = fmaf(yv.x, cv.x, a0); a1 = fmaf(yv.y, cv.y, a1); a2 = fmaf(yv.z, cv.z, a2); a3 = fmaf(yv.w, cv.w, a3);
    }
    part[(int64_t)blockIdx.y * nnz + e] = (a0 + a1) + (a2 + a3);
}
__global__ void k_reduce_parts(const float *__restrict__ part, int64_t n, int nparts, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double s = 0;
    for (int j = 0; j < nparts; ++j) s += part[(int64_t)j * n + e];
    out[e] = (float)s;
}

// ---- S2: V(k1,k2) = <Cc(k1,:), Cc(k2,:)> for the listed pairs, written symmetrically into dense K x K ----
__global__ void __launch_bounds__(256) k_pair_gram(const float *__restrict__ Cc, int64_t ldc, int64_t T, const int2 *__restrict__ pairs,
                                                   int K, float *__restrict__ V) {
    const int2 pr = pairs[blockIdx.x];
    const float *a = Cc + (int64_t)pr.x * ldc, *b = Cc + (int64_t)pr.y * ldc;
    double s = 0;
    for (int64_t t = threadIdx.x; t < T; t += 256) s += (double)a[t] * (double)b[t];
    __shared__ double red[256];
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { V[(int64_t)pr.x * K + pr.y] = (float)red[0]; V[(int64_t)pr.y * K + pr.x] = (float)red[0]; }
}

// ---- S3: one Gauss-Seidel level of HALS_spatial / HALS_spatial_thresh ---------------------------------
// ak = A(ind,k) + (U(ind,k) - A(ind,:)*V(:,k)) / cc(k);  plain: max(0,.) (HALS_spatial.m:42);
// thresh: ak(ak < sn*3/sqrt(cc)) = 0 (HALS_spatial_thresh.m:50-51).
__global__ void __launch_bounds__(128) k_hals_spatial(const int *__restrict__ lvl, const int64_t *__restrict__ colptr, const int *__restrict__ erow,
                                                      const int *__restrict__ rptr, const int *__restrict__ rcol, const int *__restrict__ rsrc,
                                                      const float *__restrict__ U, const float *__restrict__ V, int K,
                                                      const float *__restrict__ sn, int thresh, float *__restrict__ Aval) {
    const int k = lvl[blockIdx.x];
    const float cc = V[(int64_t)k * K + k];
    if (cc == 0.f) return;                                         // :38-40
    const float cthr = 3.0f / sqrtf(cc);
    for (int64_t e = colptr[k] + threadIdx.x; e < colptr[k + 1]; e += blockDim.x) {
        const int m = erow[e];
        float s = 0.f;
        for (int r = rptr[m]; r < rptr[m + 1]; ++r) s = fmaf(Aval[rsrc[r]], V[(int64_t)rcol[r] * K + k], s);
        float ak = Aval[e] + (U[e] - s) / cc;
        if (thresh) { if (ak < sn[m] * cthr) ak = 0.f; }
        else ak = fmaxf(0.f, ak);
        Aval[e] = ak;
    }
}

// ---- S4: nnls_spatial.m:26-38 + nnls() :41-109, one thread per pixel, fp64 --------------------------
constexpr int NN_MAX = 32;   // most passive variables (maxN) the kernel is instantiated for; the number of masks over a pixel is unbounded
__device__ inline bool solve_small(int n, double *M /* n x n row-major, destroyed */, double *b /* in: rhs, out: x */) {
    for (int c = 0; c < n; ++c) {
        int piv = c; double best = fabs(M[c * n + c]);
        for (int r = c + 1; r < n; ++r) { double v = fabs(M[r * n + c]); if (v > best) { best = v; piv = r; } }
        if (best == 0.0) return false;
        if (piv != c) { for (int j = 0; j < n; ++j) { double t = M[c * n + j]; M[c * n + j] = M[piv * n + j]; M[piv * n + j] = t; }
                        double t = b[c]; b[c] = b[piv]; b[piv] = t; }
        const double inv = 1.0 / M[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const double f = M[r * n + c] * inv;
            if (f != 0.0) { for (int j = c; j < n; ++j) M[r * n + j] -= f * M[c * n + j]; b[r] -= f * b[c]; }
        }
    }
    for (int r = n - 1; r >= 0; --r) { double s = b[r]; for (int j = r + 1; j < n; ++j) s -= M[r * n + j] * b[j]; b[r] = s / M[r * n + r]; }
    return true;
}

// One outer pass adds at most one variable to the passive set (:84-85) and there are at most maxN passes (:76), so s has at most maxN
// entries that were ever non-zero: the thread keeps s, b and the passive flag for those "touched" entries only (sorted by position in the
// pixel's row, which is the order of A(P,P) at :93) and reads everything else -- b from U, the Gram entries from V -- where it lies.
// `rows`: the pixels this launch solves.  The per-thread arrays (CAP^2 doubles) live in scratch memory, which the runtime sizes per DISPATCH as
// bytes per lane x the waves the grid can have in flight: one launch of the CAP = 20 instantiation over all 262144 pixels asks for 2 GB of it (and
// the first such launch of a process stalls for ~0.9 s while it is mapped).  So the pixels are split by the number of masks over them: n <= 4 (the
// bulk; 4 x 4 systems stay in registers), n <= 8, and the few crowded ones, each list through the smallest instantiation that holds min(n, maxN).
template <int CAP>
__global__ void __launch_bounds__(64) k_nnls_spatial(const int *__restrict__ rows, int nrows, const int *__restrict__ rptr, const int *__restrict__ rcol,
                                                     const int *__restrict__ rsrc, const float *__restrict__ U, const float *__restrict__ V, int K, int maxN,
                                                     double tol, float *__restrict__ Aval) {
    const int64_t ridx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ridx >= nrows) return;
    const int64_t m = rows[ridx];
    const int r0 = rptr[m], n = rptr[m + 1] - r0;
    if (n <= 0) return;
    int tpos[CAP], tcol[CAP], idx[CAP];
    double ts[CAP], tb[CAP], mu[CAP], Ms[CAP * CAP];
    bool P[CAP];
    int nt = 0;
    for (int it = 0; it < maxN; ++it) {                                      // :76
        double lmax = -1e300; int imax = 0, icol = 0; double ib = 0.0;
        for (int i = 0; i < n; ++i) {
            const int ci = rcol[r0 + i];
            const double bi = (double)U[rsrc[r0 + i]];
            double l = bi;
            for (int a = 0; a < nt; ++a) l -= (double)V[(int64_t)ci * K + tcol[a]] * ts[a];   // :77 (the untouched s are exact zeros)
            if (l > lmax) { lmax = l; imax = i; icol = ci; ib = bi; }        // first maximum
        }
        if (lmax < tol) break;                                               // :80
        int np = 0, at = -1;
        for (int a = 0; a < nt; ++a) { P[a] = ts[a] > 0.0; if (tpos[a] == imax) at = a; }   // :78
        if (at < 0) {                                                        // keep the touched list in row order
            at = nt;
            while (at > 0 && tpos[at - 1] > imax) { tpos[at] = tpos[at - 1]; tcol[at] = tcol[at - 1]; ts[at] = ts[at - 1]; tb[at] = tb[at - 1]; P[at] = P[at - 1]; --at; }
            tpos[at] = imax; tcol[at] = icol; ts[at] = 0.0; tb[at] = ib; ++nt;
        }
        P[at] = true;                                                        // :85
        for (int a = 0; a < nt; ++a) np += P[a];
        if (np > maxN) break;                                                // :86
        bool have_mu = false;
        int q = 0;
        while (np > 0) {                                                     // :90
            q = 0;
            for (int a = 0; a < nt; ++a) if (P[a]) idx[q++] = a;
            for (int a = 0; a < q; ++a) { mu[a] = tb[idx[a]]; for (int c = 0; c < q; ++c) Ms[a * q + c] = (double)V[(int64_t)tcol[idx[a]] * K + tcol[idx[c]]]; }
            if (!solve_small(q, Ms, mu)) {                                   // catch branch :94-96
                for (int a = 0; a < q; ++a) { mu[a] = tb[idx[a]]; for (int c = 0; c < q; ++c) Ms[a * q + c] = (double)V[(int64_t)tcol[idx[a]] * K + tcol[idx[c]]] + (a == c ? tol : 0.0); }
                solve_small(q, Ms, mu);
            }
            have_mu = true;
            bool all_pos = true;
            for (int a = 0; a < q; ++a) all_pos = all_pos && (mu[a] > tol);
            if (all_pos) break;                                              // :98
            double amin = 1e300;
            for (int a = 0; a < q; ++a) if (!(mu[a] > tol)) { double v = ts[idx[a]] / (ts[idx[a]] - mu[a]); if (v < amin) amin = v; }   // :102-104
            for (int a = 0; a < q; ++a) ts[idx[a]] += amin * (mu[a] - ts[idx[a]]);   // :105
            np = 0;
            for (int a = 0; a < nt; ++a) { if (ts[a] < tol) P[a] = false; np += P[a]; }   // :106
            have_mu = false;
        }
        if (have_mu) for (int a = 0; a < q; ++a) ts[idx[a]] = mu[a];         // :109
    }
    for (int i = 0; i < n; ++i) Aval[rsrc[r0 + i]] = 0.f;
    for (int a = 0; a < nt; ++a) Aval[rsrc[r0 + tpos[a]]] = (float)ts[a];
}

// ---- T1: U(k,t) = sum_e A(e) * Ysig(m_e, t)  (HALS_temporal.m:48) --------------------------------------
// one workgroup per (neuron, frame chunk); a wave owns frames t = t0 + wave, +4, ...; lanes stride the
// neuron's pixels and a 6-step DPP butterfly finishes the dot product.
constexpr int PT_NE = 6;      // entries of A per lane kept in registers over the chunk loop (64 * 6 = 384 pixels per footprint; more: the tail loop)
__global__ void __launch_bounds__(256) k_proj_temporal(const float4 *__restrict__ ysig4, int64_t d, int64_t T, const int64_t *__restrict__ colptr,
                                                       const int *__restrict__ erow, const float *__restrict__ aval, int64_t cchunk,
                                                       float *__restrict__ U, int64_t ldc) {
    const int k = blockIdx.x;
    const int64_t e0 = colptr[k], e1 = colptr[k + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t Tc = (T + 3) >> 2;
    const int64_t c0 = (int64_t)blockIdx.y * cchunk, c1 = c0 + cchunk < Tc ? c0 + cchunk : Tc;     // 4-frame groups
    // a lane's (row, value) entries do not change over the chunk loop: read once, so that the loop issues nothing but independent video loads
    // (it used to reload value and row per product, the video load waiting for the row load)
    const int nit = (int)(e1 - e0 + 63 < (int64_t)64 * PT_NE ? (e1 - e0 + 63) >> 6 : PT_NE);       // workgroup-uniform
    int rw[PT_NE]; float av[PT_NE];
#pragma unroll
    for (int i = 0; i < PT_NE; ++i) {
        const int64_t e = e0 + lane + 64 * i;
        const bool in = i < nit && e < e1;
        rw[i] = in ? erow[e] : 0; av[i] = in ? aval[e] : 0.f;
    }
    const int64_t et = e0 + (int64_t)64 * PT_NE;               // entries beyond the registers (footprints of more than 384 pixels)
    for (int64_t c = c0 + wave; c < c1; c += 4) {
        const float4 *y = ysig4 + c * d;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < PT_NE; ++i)
            if (i < nit) {
                const float4 yv = y[rw[i]];
                s0 = fmaf(av[i], yv.x, s0); s1 = fmaf(av[i], yv.y, s1); s2 = fmaf(av[i], yv.z, s2); s3 = fmaf(av[i], yv.w, s3);
            }
        for (int64_t e = et + lane; e < e1; e += 64) {
            const float a = aval[e]; const float4 yv = y[erow[e]];
            s0 = fmaf(a, yv.x, s0); s1 = fmaf(a, yv.y, s1); s2 = fmaf(a, yv.z, s2); s3 = fmaf(a, yv.w, s3);
        }
        for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o); }
        if (lane == 0) {
            float *u = U + (int64_t)k * ldc + 4 * c;        // ldc is a multiple of 4 >= T: the padded tail is never read
            u[0] = s0; u[1] = s1; u[2] = s2; u[3] = s3;
        }
    }
}

// ---- T2: V(k,k') = <A(:,k), A(:,k')> on the overlap pairs: one WAVE per pair -- lanes stride the shorter column and look their row up in the
// other by bisection (both sorted); a two-pointer merge per thread was a chain of ~300 dependent loads, 75 us whatever the size of the problem ----
__global__ void __launch_bounds__(256) k_ata_pairs(const int64_t *__restrict__ colptr, const int *__restrict__ erow, const float *__restrict__ aval,
                                                   const int *__restrict__ nk, const int *__restrict__ nidx, int nn, float *__restrict__ nval) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nn) return;
    int k = nk[i], k2 = nidx[i];
    if (colptr[k + 1] - colptr[k] > colptr[k2 + 1] - colptr[k2]) { const int t = k; k = k2; k2 = t; }
    const int64_t a0 = colptr[k], a1 = colptr[k + 1], b0 = colptr[k2], b1 = colptr[k2 + 1];
    double s = 0;
    for (int64_t a = a0 + lane; a < a1; a += 64) {
        const int r = erow[a];
        int64_t lo = b0, hi = b1;                            // first entry of column k2 with row >= r
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (erow[mid] < r) lo = mid + 1; else hi = mid; }
        if (lo < b1 && erow[lo] == r) s += (double)aval[a] * (double)aval[lo];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) nval[i] = (float)s;
}

// ---- T3: one Gauss-Seidel level of HALS_temporal (no-deconvolution branch :62-68) --------------------
constexpr int HT_NB = 64;        // neighbours of a neuron staged in LDS (more: read from the lists in global memory)
constexpr int HT_NR = 12;        // frames per thread kept in registers
__device__ __forceinline__ void hals_temporal_one(int k, const int *__restrict__ nptr, const int *__restrict__ nidx, const float *__restrict__ nval,
                                                  const float *__restrict__ aa, const float *__restrict__ U, float *C, float *Craw, int64_t ldc, int64_t T, float *red) {
    __shared__ int s_idx[HT_NB]; __shared__ float s_val[HT_NB];
    const float a = aa[k];
    float *ck = C + (int64_t)k * ldc;
    const float *uk = U + (int64_t)k * ldc;
    const int n0 = nptr[k], nn = nptr[k + 1] - n0;
    const int nt = (int)blockDim.x;
    // the neighbour list goes to LDS once: the trace loads below then depend on nothing but their own address, and a thread's frames are taken two
    // at a time -- this kernel is a handful of workgroups whose duration is the length of one thread's chain of loads
    const bool staged = nn <= HT_NB;
    if (staged && (int)threadIdx.x < nn) { s_idx[threadIdx.x] = nidx[n0 + threadIdx.x]; s_val[threadIdx.x] = nval[n0 + threadIdx.x]; }
    __syncthreads();
    float mn = INFINITY;
    float *rk = Craw + (int64_t)k * ldc;
    if (staged && T <= (int64_t)HT_NR * nt) {
        // every frame of the thread at once (T <= 12 blockDim): all trace loads of a neighbour are independent and the new values wait in registers
        // for the row minimum -- no second read of C_raw
        float vc[HT_NR], val[HT_NR];
#pragma unroll
        for (int i = 0; i < HT_NR; ++i) vc[i] = 0.f;
        const int64_t tl = T - 1;
        for (int j = 0; j < nn; ++j) {
            const float *cj = C + (int64_t)s_idx[j] * ldc; const float v = s_val[j];
#pragma unroll
            for (int i = 0; i < HT_NR; ++i) { const int64_t t = threadIdx.x + (int64_t)i * nt; vc[i] = fmaf(v, cj[t < T ? t : tl], vc[i]); }      // V(k,:)*C
        }
#pragma unroll
        for (int i = 0; i < HT_NR; ++i) {
            const int64_t t = threadIdx.x + (int64_t)i * nt, tc = t < T ? t : tl;
            val[i] = ck[tc] + (uk[tc] - vc[i]) / a;                                                    // :62
            if (t < T) mn = fminf(mn, val[i]);
        }
        for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
        __syncthreads();
        mn = red[0];
        for (int w = 1; w < (nt >> 6); ++w) mn = fminf(mn, red[w]);
#pragma unroll
        for (int i = 0; i < HT_NR; ++i) {
            const int64_t t = threadIdx.x + (int64_t)i * nt;
            if (t < T) { const float v = val[i] - mn; rk[t] = v; ck[t] = v; }                         // :66-68
        }
        return;
    }
    for (int64_t t0 = threadIdx.x; t0 < T; t0 += 2 * nt) {
        const int64_t t1 = t0 + nt;
        const bool two = t1 < T;
        const int64_t t1c = two ? t1 : t0;
        float vc0 = 0.f, vc1 = 0.f;
        if (staged) {
#pragma unroll 4
            for (int j = 0; j < nn; ++j) {
                const float *cj = C + (int64_t)s_idx[j] * ldc; const float v = s_val[j];
                vc0 = fmaf(v, cj[t0], vc0); vc1 = fmaf(v, cj[t1c], vc1);                        // V(k,:)*C
            }
        } else {
            for (int j = 0; j < nn; ++j) {
                const float *cj = C + (int64_t)nidx[n0 + j] * ldc; const float v = nval[n0 + j];
                vc0 = fmaf(v, cj[t0], vc0); vc1 = fmaf(v, cj[t1c], vc1);
            }
        }
        const float v0 = ck[t0] + (uk[t0] - vc0) / a;                                              // :62
        rk[t0] = v0; mn = fminf(mn, v0);
        if (two) { const float v1 = ck[t1] + (uk[t1] - vc1) / a; rk[t1] = v1; mn = fminf(mn, v1); }
    }
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    mn = red[0];
    for (int w = 1; w < (nt >> 6); ++w) mn = fminf(mn, red[w]);
    for (int64_t t = threadIdx.x; t < T; t += nt) {
        const float v = rk[t] - mn;                                                                // :66
        rk[t] = v; ck[t] = v;                                                                      // :67-68
    }
}
// (1024 threads per neuron: a level is a handful of workgroups whose duration is their serial depth over T, not their work)
__global__ void __launch_bounds__(1024) k_hals_temporal(const int *__restrict__ lvl, const int *__restrict__ nptr, const int *__restrict__ nidx,
                                                        const float *__restrict__ nval, const float *__restrict__ aa, const float *__restrict__ U,
                                                        float *C, float *Craw, int64_t ldc, int64_t T) {
    __shared__ float red[1024];
    hals_temporal_one(lvl[blockIdx.x], nptr, nidx, nval, aa, U, C, Craw, ldc, T, red);
}
// the same level over the jobs of a context (TemporalJob, common.hpp): workgroup -> (job, neuron)
// (the operand pointers come out of a device table: declared in the GLOBAL address space, or every access through them is a flat load / store that the
//  LDS waits of the staged neighbour list serialise -- scripts/isa_scan.py)
#define CNMFE_GPTR(T) T __attribute__((address_space(1))) *
struct HJobDev { CNMFE_GPTR(const int) nptr; CNMFE_GPTR(const int) nidx; CNMFE_GPTR(const float) nval; CNMFE_GPTR(const float) aa; CNMFE_GPTR(const float) U;
                 CNMFE_GPTR(float) C; CNMFE_GPTR(float) Craw; int64_t ldc; };
__global__ void __launch_bounds__(1024) k_hals_temporal_jobs(const HJobDev *__restrict__ jobs, const int2 *__restrict__ lvl, int64_t T) {
    __shared__ float red[1024];
    const int2 e = lvl[blockIdx.x];
    const HJobDev j = jobs[e.x];
    hals_temporal_one(e.y, (const int *)j.nptr, (const int *)j.nidx, (const float *)j.nval, (const float *)j.aa, (const float *)j.U, (float *)j.C, (float *)j.Craw, j.ldc, T, red);
}
// ---- S6: connectivity_constraint.m:1-18 on a per-neuron box -------------------------------------------
constexpr int PP_MAX = 64;    // max box side (bbox + 2 px margin each side)
__global__ void __launch_bounds__(256) k_connectivity(const int64_t *__restrict__ colptr, const int *__restrict__ erow, const float *__restrict__ aval,
                                                      const int4 *__restrict__ box /* r0,c0,h,w (0-based, box incl. margin) */, int d1, int d2,
                                                      unsigned char *__restrict__ keep) {
    __shared__ float img[PP_MAX * PP_MAX];
    __shared__ float er[PP_MAX * PP_MAX];
    __shared__ int lab[PP_MAX * PP_MAX];
    __shared__ float smax; __shared__ int sarg; __shared__ int changed;
    __shared__ float redv[256]; __shared__ int redi[256];
    const int k = blockIdx.x;
    const int4 bx = box[k];
    const int r0 = bx.x, c0 = bx.y, h = bx.z, w = bx.w, n = h * w;
    const int64_t e0 = colptr[k], e1 = colptr[k + 1];
    if (e1 == e0) return;
    for (int i = threadIdx.x; i < n; i += 256) img[i] = 0.f;
    __syncthreads();
    // image + first maximum in column-major order (connectivity_constraint.m:10)
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int pix = erow[e], r = pix % d1, c = pix / d1;
        const float v = aval[e];
        img[(c - c0) * h + (r - r0)] = v;
        if (v > bv || (v == bv && pix < bi)) { bv = v; bi = pix; }
    }
    redv[threadIdx.x] = bv; redi[threadIdx.x] = bi; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v2 = redv[threadIdx.x + o]; const int i2 = redi[threadIdx.x + o];
            if (v2 > redv[threadIdx.x] || (v2 == redv[threadIdx.x] && i2 < redi[threadIdx.x])) { redv[threadIdx.x] = v2; redi[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { smax = redv[0]; sarg = redi[0]; }
    __syncthreads();
    // the image maximum also sees the zeros outside the support
    const float vmax = fmaxf(smax, 0.f);
    // grey erosion with a 5x5 square; outside the FOV is ignored (+inf padding), outside the box is 0
    for (int i = threadIdx.x; i < n; i += 256) {
        const int lr = i % h, lc = i / h;
        float mnv = INFINITY;
        for (int dc = -2; dc <= 2; ++dc) for (int dr = -2; dr <= 2; ++dr) {
            const int rr = lr + dr, cc = lc + dc, ar = r0 + rr, ac = c0 + cc;
            if (ar < 0 || ar >= d1 || ac < 0 || ac >= d2) continue;
            const float v = (rr >= 0 && rr < h && cc >= 0 && cc < w) ? img[cc * h + rr] : 0.f;
            mnv = fminf(mnv, v);
        }
        er[i] = mnv;
    }
    __syncthreads();
    // dilation -> opening; temp = ai_open > max*thr (:15); initial labels
    const float thr = vmax * 0.01f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int lr = i % h, lc = i / h;
        if (r0 + lr < 0 || r0 + lr >= d1 || c0 + lc < 0 || c0 + lc >= d2) { lab[i] = 0; continue; }   // box cell outside the FOV
        float mxv = -INFINITY;
        for (int dc = -2; dc <= 2; ++dc) for (int dr = -2; dr <= 2; ++dr) {
            const int rr = lr + dr, cc = lc + dc, ar = r0 + rr, ac = c0 + cc;
            if (ar < 0 || ar >= d1 || ac < 0 || ac >= d2) continue;
            const float v = (rr >= 0 && rr < h && cc >= 0 && cc < w) ? er[cc * h + rr] : 0.f;
            mxv = fmaxf(mxv, v);
        }
        lab[i] = mxv > thr ? i + 1 : 0;
    }
    __syncthreads();
    // 4-connected components by min-label propagation (bwlabel(.,4), :16; only "same component" matters)
    for (int iter = 0; iter < PP_MAX * PP_MAX; ++iter) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            int l = lab[i];
            if (!l) continue;
            const int lr = i % h, lc = i / h;
            int best = l;
            if (lr > 0) { int v = lab[i - 1]; if (v && v < best) best = v; }
            if (lr < h - 1) { int v = lab[i + 1]; if (v && v < best) best = v; }
            if (lc > 0) { int v = lab[i - h]; if (v && v < best) best = v; }
            if (lc < w - 1) { int v = lab[i + h]; if (v && v < best) best = v; }
            if (best < l) { lab[i] = best; changed = 1; }
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    const int am = sarg;
    const int lmax = lab[((am / d1) - c0) * h + ((am % d1) - r0)];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int pix = erow[e], r = pix % d1, c = pix / d1;
        keep[e] = lab[(c - c0) * h + (r - r0)] == lmax ? 1 : 0;        // img(l ~= l(ind_max)) = 0  (:18)
    }
}

// =============================================================================================
// host orchestration
// =============================================================================================
struct PairGraph {
    std::vector<int2> pairs;                 // k1 <= k2, unique, diagonal included for every k
    std::vector<std::vector<int>> lower;     // lower[k] = neighbours with smaller index
    std::vector<std::vector<int>> levels;    // neurons per Gauss-Seidel level, ascending k inside a level
};

// co-occurrence graph of columns from a CSR (row -> columns), plus the order-preserving level schedule
static void build_graph(int K, const HostCSR &csr, int64_t nrow, const std::vector<char> &include, PairGraph &g) {
    g.pairs.clear(); g.lower.assign(K, {});
    for (int k = 0; k < K; ++k) g.pairs.push_back(make_int2(k, k));
    if ((int64_t)K * K <= (int64_t(1) << 28)) {
        // co-occurrence bitmap (bit k1*K + k2, k1 < k2), read back row by row: the pairs come out sorted and unique without the sort of a
        // (pixels x columns^2) pair list -- that sort was a millisecond of host time under a projection kernel that lasts 0.9 ms
        static thread_local std::vector<uint64_t> bm;
        const size_t nw = ((size_t)K * K + 63) / 64;
        bm.assign(nw, 0);
        for (int64_t r = 0; r < nrow; ++r) {
            const int64_t a = csr.rowptr[r], b = csr.rowptr[r + 1];
            for (int64_t i = a; i < b; ++i) for (int64_t j = i + 1; j < b; ++j) {
                int k1 = csr.col[i], k2 = csr.col[j];
                if (k1 > k2) std::swap(k1, k2);
                const size_t bit = (size_t)k1 * K + k2;
                bm[bit >> 6] |= uint64_t(1) << (bit & 63);
            }
        }
        for (int k1 = 0; k1 < K; ++k1) {
            const size_t b0 = (size_t)k1 * K + k1 + 1, b1 = (size_t)(k1 + 1) * K;     // bits of row k1, columns k1+1 .. K-1
            for (size_t w = b0 >> 6; w <= (b1 - 1) >> 6 && b0 < b1; ++w) {
                uint64_t x = bm[w];
                while (x) {
                    const size_t bit = (w << 6) + (size_t)__builtin_ctzll(x);
                    x &= x - 1;
                    if (bit < b0 || bit >= b1) continue;
                    const int k2 = (int)(bit - (size_t)k1 * K);
                    g.pairs.push_back(make_int2(k1, k2)); g.lower[k2].push_back(k1);
                }
            }
        }
    } else {
        std::vector<std::pair<int, int>> pr;
        for (int64_t r = 0; r < nrow; ++r) {
            const int64_t a = csr.rowptr[r], b = csr.rowptr[r + 1];
            for (int64_t i = a; i < b; ++i) for (int64_t j = i + 1; j < b; ++j) {
                int k1 = csr.col[i], k2 = csr.col[j];
                if (k1 > k2) std::swap(k1, k2);
                pr.push_back({k1, k2});
            }
        }
        std::sort(pr.begin(), pr.end());
        pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
        for (auto &p : pr) { g.pairs.push_back(make_int2(p.first, p.second)); g.lower[p.second].push_back(p.first); }
    }
    std::vector<int> lvl(K, 0);
    int nl = 0;
    for (int k = 0; k < K; ++k) {
        int l = 0;
        for (int a : g.lower[k]) if (include[a]) l = std::max(l, lvl[a] + 1);
        lvl[k] = l;
        if (include[k]) nl = std::max(nl, l + 1);
    }
    g.levels.assign(nl, {});
    for (int k = 0; k < K; ++k) if (include[k]) g.levels[lvl[k]].push_back(k);
}

// The maxIter Gauss-Seidel sweeps of ONE update as one dependency graph over (sweep, neuron) items (option sweep_dag, default 1).  HALS_spatial.m:36-44 /
// HALS_temporal.m:59-68 visit k = 1..K maxIter times; item (s, k) reads the rows of k's neighbours -- sweep s values of the lower-indexed ones, sweep s - 1 values of the
// higher-indexed ones -- and writes row k.  Its predecessors are therefore (s, j) for neighbours j < k, (s - 1, j) for neighbours j > k and (s - 1, k); the
// anti-dependencies are edges of the same kind ((s, k) -> (s, j) for j > k, (s, k) -> (s + 1, j) for j < k), so ANY order that respects the graph computes what the
// sequential loops compute, bit for bit.  A level-synchronous schedule launches maxIter x (levels of a sweep) times; the graph's depth is less whenever the chains of
// different sweeps do not line up (512 x 512, K = 500: 22 instead of 30) -- and a launch of the in-sweep deconvolution lasts as long as one trace's serial OASIS
// whatever the number of traces in it.  flag_last: bit 30 marks the items of the last sweep (the deconvolution keeps S and C_raw of that one, k_deconv).
static void dag_schedule(int K, const PairGraph &g, const std::vector<char> &include, int maxIter, bool flag_last, std::vector<std::vector<int>> &dag) {
    std::vector<std::vector<int>> upper(K);
    for (int k = 0; k < K; ++k) for (int a : g.lower[k]) upper[a].push_back(k);
    std::vector<int> prev(K, 0), cur(K, 0);                  // depth (1-based) of (s - 1, .) and (s, .); 0: no such item
    dag.clear();
    for (int s = 0; s < maxIter; ++s) {
        for (int k = 0; k < K; ++k) {
            if (!include[k]) { cur[k] = 0; continue; }
            int m = prev[k];
            for (int a : g.lower[k]) if (include[a]) m = std::max(m, cur[a]);
            for (int b : upper[k]) if (include[b]) m = std::max(m, prev[b]);
            cur[k] = m + 1;
            if ((int)dag.size() < cur[k]) dag.resize(cur[k]);
            dag[cur[k] - 1].push_back(k | ((flag_last && s == maxIter - 1) ? (1 << 30) : 0));
        }
        prev.swap(cur);
    }
}

int spatial_run(cnmfe_ctx *ctx, Patch *P, int algorithm, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                const float *A_val, const float *C, int c_order, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                const float *sn, int32_t param, float *A_out) {
    const int64_t T = P->T, d = P->d, nnz = IND_colptr[K];
    ctx->spatial_nnz = nnz == 0 ? 0 : -1;
    if (nnz == 0) return 0;
    if (nnz >= (int64_t(1) << 31)) return fail(CNMFE_EUNSUPPORTED, "nnz(IND) too large");
    DevBuf &dC = ctx->tmp[0], &dCc = ctx->tmp[1], &dCm = ctx->tmp[2];
    DevBuf *S_ = ctx->scr;
    DevBuf &dColptr = S_[0], &dErow = S_[1], &dEcol = S_[2], &dRptr = S_[3], &dRcol = S_[4], &dRsrc = S_[5], &dAval = S_[6], &dU = S_[7], &dPart = S_[8], &dV = S_[9], &dPairs = S_[10], &dSn = S_[11], &dLvl = S_[12];
    HostTrace ht(ctx, "spatial");
    int64_t ldc;
    RET(upload_centered(ctx, dC, C, K, T, c_order, dCc, dCm, &ldc));
    if (P->pend && !residual_term_foldable_spatial(P, K, ldc)) RET(residual_materialize(ctx, P));    // (a term that cannot enter through the projection below)
    // A restricted to the IND pattern (A(~active_pixel) = 0, HALS_spatial.m:26); NNLS starts from 0 (:32)
    std::vector<float> aval(nnz, 0.f);
    std::vector<int32_t> ecol(nnz);
    for (int32_t k = 0; k < K; ++k) {
        int64_t a = A_colptr ? A_colptr[k] : 0, ae = A_colptr ? A_colptr[k + 1] : 0;
        for (int64_t e = IND_colptr[k]; e < IND_colptr[k + 1]; ++e) {
            ecol[e] = k;
            while (a < ae && A_rowidx[a] < IND_rowidx[e]) ++a;
            if (a < ae && A_rowidx[a] == IND_rowidx[e] && algorithm != CNMFE_SPATIAL_NNLS) aval[e] = A_val[a];
        }
    }
    ht.mark("traces + A on the mask");
    HostCSR csr; csc_to_csr(d, K, IND_colptr, IND_rowidx, nullptr, csr);
    ht.mark("csr of the mask");
    if (algorithm == CNMFE_SPATIAL_NNLS && ((int)param > NN_MAX || (int)param < 1))
        return fail(CNMFE_EUNSUPPORTED, "maxN = %d; the NNLS kernel holds 1..%d passive variables", (int)param, NN_MAX);
    const std::vector<int32_t> &rptr = csr.rowptr;
    RET(to_dev(ctx, dColptr, IND_colptr, (size_t)K + 1));
    ht.mark("u colptr");
    RET(to_dev(ctx, dErow, IND_rowidx, (size_t)nnz));
    ht.mark("u erow");
    RET(to_dev(ctx, dEcol, ecol.data(), (size_t)nnz));
    ht.mark("u ecol");
    RET(to_dev(ctx, dRptr, rptr.data(), rptr.size()));
    ht.mark("u rptr");
    RET(to_dev(ctx, dRcol, csr.col.data(), csr.col.size()));
    RET(to_dev(ctx, dRsrc, csr.src.data(), csr.src.size()));
    ht.mark("u rcol rsrc");
    RET(to_dev(ctx, dAval, aval.data(), aval.size()));
    ht.mark("u aval");
    if (sn) RET(to_dev(ctx, dSn, sn, (size_t)d));
    ht.mark("uploads");
    // S1
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(32, T / 256));
    const int64_t tchunk = ((T + nparts - 1) / nparts + 3) & ~int64_t(3);
    RET(dPart.ensure((size_t)nparts * nnz * sizeof(float)));
    RET(dU.ensure((size_t)nnz * sizeof(float)));
    ht.mark("buffers");
    // a virtual residual (no sweep has run, resid.hip): U = P - W P out of the table P = Yc Cc' (vproj.hip); if that path cannot serve this update the sweep
    // runs now and Ysig is projected as before
    bool virt = P->ysig_virtual;
    if (virt) {
        const int rcv = P->res_kind == 2 ? vproj_spatial_ssub(ctx, P, K, C, c_order, IND_colptr, IND_rowidx, dErow.as<int>(), dEcol.as<int>(), dCc.as<float>(), ldc, dU.as<float>())
                                         : vproj_spatial(ctx, P, K, C, c_order, IND_colptr, IND_rowidx, dErow.as<int>(), dEcol.as<int>(), dCc.as<float>(), ldc, dU.as<float>());
        if (rcv < 0) return rcv;
        if (rcv > 0) { RET(residual_realize(ctx, P)); virt = false; }
    }
    if (!virt) {
    LAUNCH(ctx, "spatial_proj_U", k_proj_spatial, dim3((unsigned)((nnz + 255) / 256), nparts), dim3(256), 0, P->ysig.as<float4>(), d, T,
           dErow.as<int>(), dEcol.as<int>(), nnz, dCc.as<float>(), ldc, tchunk, dPart.as<float>());
    LAUNCH(ctx, "reduce_parts", k_reduce_parts, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, dPart.as<float>(), nnz, nparts, dU.as<float>());
    }
    ht.mark("projection");
    // a footprint term pending beside Ysig (patches with halo neurons: the sweep ran without it) enters here: U += (W A_prev)(Cc_prev Cc')
    RET(residual_term_fold_spatial(ctx, P, K, nnz, dErow.as<int>(), dEcol.as<int>(), dCc.as<float>(), ldc, dU.as<float>(), S_[20]));
    ht.mark("footprint term");
    // S2 on the co-occurrence pairs
    std::vector<char> include(K, 1);
    PairGraph g; build_graph(K, csr, d, include, g);
    ht.mark("graph");
    RET(to_dev(ctx, dPairs, g.pairs.data(), g.pairs.size()));
    RET(dV.ensure((size_t)K * K * sizeof(float)));
    CK(hipMemsetAsync(dV.p, 0, (size_t)K * K * sizeof(float), ctx->st()));
    LAUNCH(ctx, "spatial_pair_gram", k_pair_gram, dim3((unsigned)g.pairs.size()), dim3(256), 0, dCc.as<float>(), ldc, T, dPairs.as<int2>(), K, dV.as<float>());
    if (algorithm == CNMFE_SPATIAL_NNLS) {
        const int maxN = (int)param;
        std::vector<int> rows[3];                               // by min(n, maxN): <= 4, <= 8, more
        for (int64_t m = 0; m < d; ++m) {
            const int n = std::min<int>(csr.rowptr[m + 1] - csr.rowptr[m], maxN);
            if (n > 0) rows[n <= 4 ? 0 : n <= 8 ? 1 : 2].push_back((int)m);
        }
        DevBuf &dRows = S_[19];
        std::vector<int> all; int off3[4] = {0, 0, 0, 0};
        for (int g_ = 0; g_ < 3; ++g_) { all.insert(all.end(), rows[g_].begin(), rows[g_].end()); off3[g_ + 1] = (int)all.size(); }
        RET(to_dev(ctx, dRows, all.data(), all.size()));
#define NNLS_GO(CAP, G) do { if (off3[G + 1] > off3[G]) { LAUNCH(ctx, "spatial_nnls", k_nnls_spatial<CAP>, dim3((unsigned)((off3[G + 1] - off3[G] + 63) / 64)), dim3(64), 0, \
                            dRows.as<int>() + off3[G], off3[G + 1] - off3[G], dRptr.as<int>(), dRcol.as<int>(), dRsrc.as<int>(), dU.as<float>(), dV.as<float>(), K, \
                            maxN, 1e-4, dAval.as<float>()); } } while (0)
        NNLS_GO(4, 0); NNLS_GO(8, 1);
        if (maxN <= 20) { NNLS_GO(20, 2); } else { NNLS_GO(32, 2); }
#undef NNLS_GO
    } else {
        std::vector<int> flat; std::vector<int> off;
        const bool dag_on = ctx->opt("sweep_dag", 1) != 0 && param > 1;
        if (dag_on) { std::vector<std::vector<int>> dag; dag_schedule(K, g, include, (int)param, false, dag); g.levels.swap(dag); }
        for (auto &l : g.levels) { off.push_back((int)flat.size()); flat.insert(flat.end(), l.begin(), l.end()); }
        RET(to_dev(ctx, dLvl, flat.data(), flat.size()));
        for (int it = 0; it < (dag_on ? 1 : param); ++it)
            for (size_t l = 0; l < g.levels.size(); ++l)
                LAUNCH(ctx, "spatial_hals_level", k_hals_spatial, dim3((unsigned)g.levels[l].size()), dim3(128), 0, dLvl.as<int>() + off[l],
                       dColptr.as<int64_t>(), dErow.as<int>(), dRptr.as<int>(), dRcol.as<int>(), dRsrc.as<int>(), dU.as<float>(), dV.as<float>(), K,
                       dSn.as<float>(), algorithm == CNMFE_SPATIAL_HALS_THRESH ? 1 : 0, dAval.as<float>());
    }
    ht.mark("sweep launches");
    ctx->spatial_nnz = nnz;                                  // the result stays in scr[6] until the next spatial update (cnmfe_update_spatial_fetch)
    if (!A_out) return 0;                                    // deferred: the caller does other host work under the sweeps and fetches afterwards
    CK(hipMemcpyAsync(A_out, dAval.p, (size_t)nnz * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    return ctx_check_errflag(ctx);                           // the wait, and what the kernels queued since the last one had to report
}

int spatial_fetch(cnmfe_ctx *ctx, float *A_out, int64_t nnz) {
    if (ctx->spatial_nnz < 0 || nnz != ctx->spatial_nnz) return fail(CNMFE_ESTATE, "no deferred spatial update of %lld values (last one: %lld)", (long long)nnz, (long long)ctx->spatial_nnz);
    if (nnz) CK(hipMemcpyAsync(A_out, ctx->scr[6].p, (size_t)nnz * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    return ctx_check_errflag(ctx);
}

// implemented in deconv.hip
struct DeconvCfg; struct DeconvIO; struct DeconvScratch;
int temporal_deconv_sweeps(cnmfe_ctx *ctx, const cnmfe_deconv_opts *dopts, int64_t T, int K, int maxIter, const std::vector<std::vector<int>> &levels,
                           const int *dLvl, const std::vector<int> &off, float *dC, float *dCraw, float *dS, int64_t ldc, const float *dU,
                           const int *dNptr, const int *dNidx, const float *dNval, const float *dAa, float *dPars, float *dSn);

// ---- fast_temporal (use_c_hat = false): C_raw = (A .* [A >= max(A)/2])' * Y ./ aa   (update_temporal_parallel.m:314-337)
__global__ void k_scale_rows(float *__restrict__ U, int64_t ldc, int64_t T, const float *__restrict__ inv) {
    const int k = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) U[(int64_t)k * ldc + t] *= inv[k];
}

int fast_temporal_run(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                      int c_order, float *C_raw_out, float *aa_out) {
    const int64_t T = P->T, d = P->d, nnz = A_colptr[K];
    if (nnz >= (int64_t(1) << 31)) return fail(CNMFE_EUNSUPPORTED, "nnz(A) too large");
    // tmp_A = A .* (A ./ max(A,[],1) >= 0.5), aa = sum(tmp_A.^2, 1); aa == 0 -> 1/inf = 0 (:327-333)
    std::vector<float> av((size_t)std::max<int64_t>(1, nnz)), inv(K), aa(K);
    for (int k = 0; k < K; ++k) {
        double mx = 0.0;                                                   // max over the (sparse) column: implicit zeros count
        bool any = false;
        for (int64_t e = A_colptr[k]; e < A_colptr[k + 1]; ++e) { if (!any || A_val[e] > mx) mx = A_val[e]; any = true; }
        if (any && A_colptr[k + 1] - A_colptr[k] < d && mx < 0.0) mx = 0.0;
        double s = 0.0;
        for (int64_t e = A_colptr[k]; e < A_colptr[k + 1]; ++e) {
            const double a = A_val[e];
            const bool keep = (a * (1.0 / mx)) >= 0.5;                     // NaN (mx == 0) compares false, like MATLAB's ge
            av[e] = keep ? (float)a : 0.f;
            if (keep) s += a * a;
        }
        aa[k] = (float)s; inv[k] = s > 0.0 ? (float)(1.0 / s) : 0.f;
    }
    DevBuf *S_ = ctx->scr;
    DevBuf &dColptr = S_[0], &dErow = S_[1], &dAval = S_[6], &dU = ctx->last_craw, &dInv = S_[13];
    const int64_t ldc = (T + 3) & ~int64_t(3);
    RET(to_dev(ctx, dColptr, A_colptr, (size_t)K + 1));
    RET(to_dev(ctx, dErow, A_rowidx, (size_t)nnz));
    RET(to_dev(ctx, dAval, av.data(), (size_t)nnz));
    RET(to_dev(ctx, dInv, inv.data(), (size_t)K));
    RET(dU.ensure((size_t)K * ldc * sizeof(float)));
    CK(hipMemsetAsync(dU.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
    const int64_t Tc = (T + 3) / 4;
    const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(64, Tc / 32));
    const int64_t tchunk = (Tc + nchunk - 1) / nchunk;
    if (nnz > 0)
        LAUNCH(ctx, "temporal_proj_U", k_proj_temporal, dim3(K, nchunk), dim3(256), 0, P->ysig.as<float4>(), d, T, dColptr.as<int64_t>(),
               dErow.as<int>(), dAval.as<float>(), tchunk, dU.as<float>(), ldc);
    LAUNCH(ctx, "temporal_scale_rows", k_scale_rows, dim3((unsigned)((T + 255) / 256), (unsigned)K), dim3(256), 0, dU.as<float>(), ldc, T, dInv.as<float>());
    RET(to_dev(ctx, ctx->last_aa, aa.data(), (size_t)K));
    ctx->last_t_K = K; ctx->last_t_ldc = ldc; ctx->last_t_T = T; ctx->last_t_valid = true;      // for cnmfe_stitch_add
    RET(download_traces(ctx, dU.as<float>(), ldc, C_raw_out, K, T, c_order));
    if (aa_out) memcpy(aa_out, aa.data(), (size_t)K * sizeof(float));
    return 0;
}

int temporal_run(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                 const float *C_in, int c_order, int32_t maxIter, float *C_out, float *C_raw_out, float *aa_out,
                 const cnmfe_deconv_opts *dopts, float *kernel_pars, float *S_out, float *sn_out, TemporalJob *job) {
    const int64_t T = P->T, d = P->d, nnz = A_colptr[K];
    if (nnz >= (int64_t(1) << 31)) return fail(CNMFE_EUNSUPPORTED, "nnz(A) too large");
    // a job (cnmfe_hals_temporal_job) works in buffers of its own and stops in front of the sweeps: temporal_sweep_jobs runs them for all jobs together
    DevBuf *S_ = ctx->scr;
    DevBuf &dC = job ? job->dC : ctx->tmp[0];
    DevBuf &dColptr = job ? job->dColptr : S_[0], &dErow = job ? job->dErow : S_[1], &dAval = job ? job->dAval : S_[6], &dU = job ? job->dU : S_[7],
           &dCraw = job ? job->dCraw : ctx->last_craw, &dNk = job ? job->dNk : S_[14], &dNidx = job ? job->dNidx : S_[15], &dNval = job ? job->dNval : S_[16],
           &dNptr = job ? job->dNptr : S_[17], &dAa = job ? job->dAa : ctx->last_aa, &dLvl = S_[12], &dOvf = job ? job->dOvf : S_[18];
    if (!job) ctx->last_t_valid = false;
    HostTrace ht(ctx, "temporal");
    int64_t ldc;
    RET(upload_traces(ctx, dC, C_in, K, T, c_order, &ldc));
    RET(to_dev(ctx, dColptr, A_colptr, (size_t)K + 1));
    RET(to_dev(ctx, dErow, A_rowidx, (size_t)nnz));
    RET(to_dev(ctx, dAval, A_val, (size_t)nnz));
    RET(dU.ensure((size_t)K * ldc * sizeof(float)));
    RET(dCraw.ensure((size_t)K * ldc * sizeof(float)));
    CK(hipMemsetAsync(dU.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
    CK(hipMemsetAsync(dCraw.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));             // C_raw = zeros(K,T)  (:45)
    // T1
    const int64_t Tc = (T + 3) / 4;
    const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(64, Tc / 32));
    const int64_t tchunk = (Tc + nchunk - 1) / nchunk;                                              // in 4-frame groups
    bool term_applied = false;
    auto reproject = [&]() -> int {                             // fold the pending term into Ysig and project again
        RET(residual_materialize(ctx, P));
        CK(hipMemsetAsync(dU.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
        LAUNCH(ctx, "temporal_proj_U", k_proj_temporal, dim3(K, nchunk), dim3(256), 0, P->ysig.as<float4>(), d, T, dColptr.as<int64_t>(),
               dErow.as<int>(), dAval.as<float>(), tchunk, dU.as<float>(), ldc);
        return 0;
    };
    if (nnz > 0) {
        // a footprint term still pending on the residual enters through A' (W A)(C - mean C); if it cannot, it is folded into Ysig first
        if (P->pend && (P->pend_ldc != ldc || (P->res_ac && P->res_ldc != ldc))) RET(residual_materialize(ctx, P));
        // a virtual residual: A' Ysig = B' Yc + A' (Ymean - b0), B = A - W'A, one block-tiled pass over the centred video (vproj.hip)
        bool virt = P->ysig_virtual;
        if (virt) {
            const int rcv = P->res_kind == 2 ? vproj_temporal_ssub(ctx, P, K, A_colptr, A_rowidx, A_val, dColptr.as<int64_t>(), dErow.as<int>(), dAval.as<float>(), dU.as<float>(), ldc)
                                             : vproj_temporal(ctx, P, K, A_colptr, A_rowidx, A_val, dColptr.as<int64_t>(), dErow.as<int>(), dAval.as<float>(), dU.as<float>(), ldc);
            if (rcv < 0) return rcv;
            if (rcv > 0) { RET(residual_realize(ctx, P)); virt = false; }
        }
        if (!virt)
        LAUNCH(ctx, "temporal_proj_U", k_proj_temporal, dim3(K, nchunk), dim3(256), 0, P->ysig.as<float4>(), d, T, dColptr.as<int64_t>(),
               dErow.as<int>(), dAval.as<float>(), tchunk, dU.as<float>(), ldc);
        RET(dOvf.ensure(64));
        CK(hipMemsetAsync(dOvf.p, 0, 64, ctx->st()));
        const int rc_ = residual_term_project(ctx, P, K, dColptr.as<int64_t>(), dErow.as<int>(), dAval.as<float>(), dU.as<float>(), ldc, dOvf.as<int>());
        if (rc_ < 0) return rc_;
        if (rc_ > 0) RET(reproject());
        else term_applied = P->pend;
    }
    ht.mark("uploads + projection launch");
    // T2: overlap graph + V values (neighbour lists include k itself: V(k,k) = aa(k))
    HostCSR csr; csc_to_csr(d, K, A_colptr, A_rowidx, A_val, csr);
    std::vector<char> nonempty(K, 0);
    for (int k = 0; k < K; ++k) nonempty[k] = A_colptr[k + 1] > A_colptr[k];
    PairGraph g; build_graph(K, csr, d, nonempty, g);
    std::vector<std::vector<int>> nb(K);
    for (auto &p : g.pairs) { nb[p.x].push_back(p.y); if (p.x != p.y) nb[p.y].push_back(p.x); }
    std::vector<int> nptr(K + 1, 0), nk, nidx, diag(K, 0);
    for (int k = 0; k < K; ++k) {
        std::sort(nb[k].begin(), nb[k].end());
        nptr[k] = (int)nidx.size();
        for (int j : nb[k]) { if (j == k) diag[k] = (int)nidx.size(); nk.push_back(k); nidx.push_back(j); }
    }
    nptr[K] = (int)nidx.size();
    const int nn = (int)nidx.size();
    RET(to_dev(ctx, dNk, nk.data(), nk.size())); RET(to_dev(ctx, dNidx, nidx.data(), nidx.size())); RET(to_dev(ctx, dNptr, nptr.data(), nptr.size()));
    RET(dNval.ensure((size_t)std::max(1, nn) * sizeof(float)));
    LAUNCH(ctx, "temporal_ata_pairs", k_ata_pairs, dim3((nn + 3) / 4), dim3(256), 0, dColptr.as<int64_t>(), dErow.as<int>(), dAval.as<float>(),
           dNk.as<int>(), dNidx.as<int>(), nn, dNval.as<float>());
    // T3's level schedule only needs WHICH neurons are updated (ind_update = find(aa > 0), :51): aa(k) = sum of squares of column k is positive
    // exactly when some stored value squares to a non-zero float, so the schedule is built here, under the projection kernel, instead of
    // after the wait for the device's A'A
    std::vector<char> upd(K, 0);
    for (int k = 0; k < K; ++k)
        for (int64_t e = A_colptr[k]; e < A_colptr[k + 1] && !upd[k]; ++e) upd[k] = A_val[e] * A_val[e] > 0.f;
    if (upd != nonempty) build_graph(K, csr, d, upd, g);     // (almost always the same set: a stored column without a non-zero value is rare -- reuse the graph above)
    std::vector<int> flat, off;
    const bool dag_on = ctx->opt("sweep_dag", 1) != 0 && maxIter > 1;
    if (dag_on) { std::vector<std::vector<int>> dag; dag_schedule(K, g, upd, maxIter, dopts != nullptr, dag); g.levels.swap(dag); }
    for (auto &l : g.levels) { off.push_back((int)flat.size()); flat.insert(flat.end(), l.begin(), l.end()); }
    if (!job) RET(to_dev(ctx, dLvl, flat.data(), flat.size()));
    ht.mark("csr + graph + lists + levels");
    if (job) {
        // no wait here: A'A (and the overflow word of the term projection) go to the job's pinned buffer behind the kernels; temporal_sweep_jobs finishes the job
        const size_t need = ((size_t)nn + 1) * sizeof(float);
        if (need > job->pin_cap) {
            if (job->pin) (void)hipHostFree(job->pin);
            job->pin = nullptr; job->pin_cap = 0;
            CK(hipHostMalloc((void **)&job->pin, need * 2, hipHostMallocDefault));
            job->pin_cap = need * 2;
        }
        job->nn = nn; job->pin[nn] = 0.f;
        CK(hipMemcpyAsync(job->pin, dNval.p, (size_t)nn * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
        if (term_applied) CK(hipMemcpyAsync(job->pin + nn, dOvf.p, sizeof(int), hipMemcpyDeviceToHost, ctx->st()));
        job->P = P; job->K = K; job->ldc = ldc; job->T = T; job->maxIter = maxIter; job->deconv = dopts != nullptr; job->swept = false; job->finished = false;
        job->diag = std::move(diag); job->upd = std::move(upd); job->term_applied = term_applied;
        job->levels = std::move(g.levels); job->dag = dag_on;
        if (dopts) {
            job->dopts = *dopts;
            RET(job->dS.ensure((size_t)K * ldc * sizeof(float)));
            CK(hipMemsetAsync(job->dS.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));         // S = zeros(K,T)  (:55)
            RET(to_dev(ctx, job->dPars, kernel_pars, (size_t)K));
            RET(job->dSn.ensure((size_t)K * sizeof(float)));
            CK(hipMemsetAsync(job->dSn.p, 0, (size_t)K * sizeof(float), ctx->st()));
            RET(job->dB.ensure((size_t)K * sizeof(float)));
        }
        ht.mark("job queued");
        return 0;
    }
    std::vector<float> nval(nn);
    int ovf = 0;
    CK(hipMemcpyAsync(nval.data(), dNval.p, (size_t)nn * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    if (term_applied) CK(hipMemcpyAsync(&ovf, dOvf.p, sizeof(int), hipMemcpyDeviceToHost, ctx->st()));
    RET(ctx_check_errflag(ctx));                                // the wait for A'A; also what the residual's kernels had to report
    ht.mark("wait for A'A (sync)");
    if (ovf) RET(reproject());                                  // a footprint near more than 512 traces: the list kernel gave up
    std::vector<float> aa(K);
    for (int k = 0; k < K; ++k) {
        aa[k] = nval[diag[k]];
        if ((aa[k] > 0.f) != (upd[k] != 0)) return fail(CNMFE_EHIP, "temporal update: aa(%d) = %g disagrees with the host-side test of its column", k, (double)aa[k]);
    }
    RET(to_dev(ctx, dAa, aa.data(), aa.size()));
    ht.mark("level schedule");
    if (!dopts) {
        for (int it = 0; it < (dag_on ? 1 : maxIter); ++it)
            for (size_t l = 0; l < g.levels.size(); ++l)
                LAUNCH(ctx, "temporal_hals_level", k_hals_temporal, dim3((unsigned)g.levels[l].size()), dim3(T >= 2048 ? 1024 : 256), 0, dLvl.as<int>() + off[l], dNptr.as<int>(),
                       dNidx.as<int>(), dNval.as<float>(), dAa.as<float>(), dU.as<float>(), dC.as<float>(), dCraw.as<float>(), ldc, T);
    } else {
        DevBuf &dS = S_[19], &dPars = S_[21], &dSn = S_[22];
        RET(dS.ensure((size_t)K * ldc * sizeof(float)));
        CK(hipMemsetAsync(dS.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));                 // S = zeros(K,T)  (:55)
        RET(to_dev(ctx, dPars, kernel_pars, (size_t)K));
        RET(dSn.ensure((size_t)K * sizeof(float)));
        CK(hipMemsetAsync(dSn.p, 0, (size_t)K * sizeof(float), ctx->st()));
        RET(temporal_deconv_sweeps(ctx, dopts, T, K, dag_on ? -1 : maxIter, g.levels, dLvl.as<int>(), off, dC.as<float>(), dCraw.as<float>(), dS.as<float>(), ldc,
                                   dU.as<float>(), dNptr.as<int>(), dNidx.as<int>(), dNval.as<float>(), dAa.as<float>(), dPars.as<float>(), dSn.as<float>()));
        // nothing asked back (the sharded-free update_temporal_parallel keeps C_raw and aa on the device for the stitch and re-estimates the time constants
        // in deconvTemporal): kernel_pars is then input only and the call returns with the sweeps in flight
        const bool want_back = C_out || C_raw_out || S_out || sn_out;
        RET(download_traces(ctx, dS.as<float>(), ldc, S_out, K, T, c_order));
        if (want_back) {
            CK(hipMemcpyAsync(kernel_pars, dPars.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
            if (sn_out) CK(hipMemcpyAsync(sn_out, dSn.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
            CK(hipStreamSynchronize(ctx->st()));
        }
    }
    ctx->last_t_K = K; ctx->last_t_ldc = ldc; ctx->last_t_T = T; ctx->last_t_valid = true;      // C_raw rows + aa stay on the device for cnmfe_stitch_add
    RET(download_traces(ctx, dC.as<float>(), ldc, C_out, K, T, c_order));
    RET(download_traces(ctx, dCraw.as<float>(), ldc, C_raw_out, K, T, c_order));
    if (aa_out) memcpy(aa_out, aa.data(), (size_t)K * sizeof(float));
    if (C_out || C_raw_out) CK(hipStreamSynchronize(ctx->st()));
    return 0;
}

struct DeconvJobDev;     // deconv.hip
int temporal_deconv_sweeps_jobs(cnmfe_ctx *ctx, const cnmfe_deconv_opts *dopts, int64_t T, int maxIter, const std::vector<TemporalJob *> &jobs,
                                const int2 *dList, const std::vector<int> &off, DevBuf &dTab);

// Level l of every job in one launch (jobs are independent; inside a job the levels keep the k = 1..K order of HALS_temporal.m:60)
int temporal_sweep_jobs(cnmfe_ctx *ctx) {
    std::vector<TemporalJob *> jobs;
    for (int i = 0; i < ctx->tjobs_used; ++i) if (!ctx->tjobs[i]->swept && ctx->tjobs[i]->K > 0) jobs.push_back(ctx->tjobs[i]);
    if (jobs.empty()) return 0;
    // ONE wait for the A'A of all jobs (and for what the residuals' kernels had to report), then every job is finished as temporal_run finishes a single one
    bool waiting = false;
    for (auto *j : jobs) waiting = waiting || !j->finished;
    if (waiting) RET(ctx_check_errflag(ctx));
    for (auto *j : jobs) {
        if (j->finished) continue;
        int ovf = 0;
        memcpy(&ovf, j->pin + j->nn, sizeof(int));
        if (j->term_applied && ovf) {                            // a footprint near more than 512 traces: fold the term into Ysig and project again
            Patch *P = j->P;
            const int64_t Tc = (j->T + 3) / 4;
            const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(64, Tc / 32));
            const int64_t tchunk = (Tc + nchunk - 1) / nchunk;
            RET(residual_materialize(ctx, P));
            CK(hipMemsetAsync(j->dU.p, 0, (size_t)j->K * j->ldc * sizeof(float), ctx->st()));
            LAUNCH(ctx, "temporal_proj_U", k_proj_temporal, dim3(j->K, nchunk), dim3(256), 0, P->ysig.as<float4>(), P->d, j->T, j->dColptr.as<int64_t>(),
                   j->dErow.as<int>(), j->dAval.as<float>(), tchunk, j->dU.as<float>(), j->ldc);
        }
        std::vector<float> aa(j->K);
        for (int k = 0; k < j->K; ++k) {
            aa[k] = j->pin[j->diag[k]];
            if ((aa[k] > 0.f) != (j->upd[k] != 0)) return fail(CNMFE_EHIP, "temporal update: aa(%d) = %g disagrees with the host-side test of its column", k, (double)aa[k]);
        }
        RET(to_dev(ctx, j->dAa, aa.data(), aa.size()));
        j->finished = true;
    }
    const TemporalJob *j0 = jobs[0];
    size_t lmax = 0;
    for (auto *j : jobs) {
        if (j->T != j0->T || j->maxIter != j0->maxIter || j->deconv != j0->deconv || j->dag != j0->dag || (j->deconv && memcmp(&j->dopts, &j0->dopts, sizeof(j->dopts)) != 0))
            return fail(CNMFE_ESTATE, "the temporal jobs of one sweep must share T, maxIter and the deconvolution options");
        lmax = std::max(lmax, j->levels.size());
    }
    std::vector<int2> flat; std::vector<int> off;
    for (size_t l = 0; l < lmax; ++l) {
        off.push_back((int)flat.size());
        for (size_t ji = 0; ji < jobs.size(); ++ji)
            if (l < jobs[ji]->levels.size()) for (int k : jobs[ji]->levels[l]) flat.push_back(make_int2((int)ji, k));
    }
    off.push_back((int)flat.size());
    DevBuf &dList = ctx->scr[12], &dTab = ctx->scr[13];
    RET(to_dev(ctx, dList, flat.data(), flat.size()));
    HostTrace ht(ctx, "temporal sweep (jobs)");
    if (!j0->deconv) {
        std::vector<HJobDev> tab(jobs.size());
        for (size_t ji = 0; ji < jobs.size(); ++ji) {
            TemporalJob *j = jobs[ji];
            tab[ji] = HJobDev{(CNMFE_GPTR(const int))j->dNptr.as<int>(), (CNMFE_GPTR(const int))j->dNidx.as<int>(), (CNMFE_GPTR(const float))j->dNval.as<float>(),
                              (CNMFE_GPTR(const float))j->dAa.as<float>(), (CNMFE_GPTR(const float))j->dU.as<float>(), (CNMFE_GPTR(float))j->dC.as<float>(),
                              (CNMFE_GPTR(float))j->dCraw.as<float>(), j->ldc};
        }
        RET(to_dev(ctx, dTab, tab.data(), tab.size()));
        for (int it = 0; it < (j0->dag ? 1 : j0->maxIter); ++it)          // (dag: the levels ARE the items of all sweeps, dag_schedule)
            for (size_t l = 0; l < lmax; ++l) {
                const int n = off[l + 1] - off[l];
                if (n > 0)
                    LAUNCH(ctx, "temporal_hals_level", k_hals_temporal_jobs, dim3((unsigned)n), dim3(j0->T >= 2048 ? 1024 : 256), 0, dTab.as<HJobDev>(), dList.as<int2>() + off[l], j0->T);
            }
    } else {
        RET(temporal_deconv_sweeps_jobs(ctx, &j0->dopts, j0->T, j0->dag ? -1 : j0->maxIter, jobs, dList.as<int2>(), off, dTab));
    }
    for (auto *j : jobs) j->swept = true;
    ht.mark("launches");
    return 0;
}

static int postproc_boxes(int32_t d1, int32_t d2, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, std::vector<int4> &box);

// The connectivity constraint on the result of a deferred spatial update WHERE IT LIES (scr[0], scr[1], scr[6]: the mask's CSC pattern and the new
// values, explicit zeros included -- they are zero pixels of the footprint image either way): valid when the patch is the whole field of view, so
// that patch rows are FOV pixels.  One wait, one download of values + keep flags; no second upload of A.
int spatial_fetch_connected(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *IND_colptr, const int32_t *IND_rowidx, float *A_out, uint8_t *keep, bool wait) {
    const int64_t nnz = IND_colptr[K];
    if (ctx->spatial_nnz < 0 || nnz != ctx->spatial_nnz) return fail(CNMFE_ESTATE, "no deferred spatial update of %lld values (last one: %lld)", (long long)nnz, (long long)ctx->spatial_nnz);
    if (nnz == 0) return 0;
    std::vector<int4> box;
    RET(postproc_boxes(d1, d2, K, IND_colptr, IND_rowidx, box));
    DevBuf *S_ = ctx->scr;
    DevBuf &dColptr = S_[0], &dErow = S_[1], &dAval = S_[6], &dBox = S_[13], &dKeep = S_[14];
    RET(to_dev(ctx, dBox, box.data(), box.size()));
    RET(dKeep.ensure((size_t)nnz));
    CK(hipMemsetAsync(dKeep.p, 0, (size_t)nnz, ctx->st()));
    LAUNCH(ctx, "spatial_connectivity", k_connectivity, dim3(K), dim3(256), 0, dColptr.as<int64_t>(), dErow.as<int>(), dAval.as<float>(),
           dBox.as<int4>(), d1, d2, dKeep.as<unsigned char>());
    CK(hipMemcpyAsync(A_out, dAval.p, (size_t)nnz * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipMemcpyAsync(keep, dKeep.p, (size_t)nnz, hipMemcpyDeviceToHost, ctx->st()));
    return wait ? ctx_check_errflag(ctx) : 0;                // (no wait: the caller records a ticket behind the copies, cnmfe_update_spatial_fetch_connected_async)
}

static int postproc_boxes(int32_t d1, int32_t d2, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, std::vector<int4> &box) {
    box.assign(K, make_int4(0, 0, 0, 0));
    for (int k = 0; k < K; ++k) {
        int rmin = d1, rmax = -1, cmin = d2, cmax = -1;
        for (int64_t e = A_colptr[k]; e < A_colptr[k + 1]; ++e) {
            int r = A_rowidx[e] % d1, c = A_rowidx[e] / d1;
            rmin = std::min(rmin, r); rmax = std::max(rmax, r); cmin = std::min(cmin, c); cmax = std::max(cmax, c);
        }
        if (rmax < 0) continue;
        int h = rmax - rmin + 5, w = cmax - cmin + 5;
        if (h > PP_MAX || w > PP_MAX)
            return fail(CNMFE_EUNSUPPORTED, "footprint %d spans %dx%d pixels; post-processing supports %dx%d", k, h - 4, w - 4, PP_MAX - 4, PP_MAX - 4);
        box[k] = make_int4(rmin - 2, cmin - 2, h, w);
    }
    return 0;
}

int postproc_run(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                 const float *A_val, uint8_t *keep) {
    const int64_t nnz = A_colptr[K];
    if (nnz == 0) return 0;
    std::vector<int4> box(K);
    for (int k = 0; k < K; ++k) {
        int rmin = d1, rmax = -1, cmin = d2, cmax = -1;
        for (int64_t e = A_colptr[k]; e < A_colptr[k + 1]; ++e) {
            int r = A_rowidx[e] % d1, c = A_rowidx[e] / d1;
            rmin = std::min(rmin, r); rmax = std::max(rmax, r); cmin = std::min(cmin, c); cmax = std::max(cmax, c);
        }
        if (rmax < 0) { box[k] = make_int4(0, 0, 0, 0); continue; }
        int h = rmax - rmin + 5, w = cmax - cmin + 5;
        if (h > PP_MAX || w > PP_MAX)
            return fail(CNMFE_EUNSUPPORTED, "footprint %d spans %dx%d pixels; post-processing supports %dx%d", k, h - 4, w - 4, PP_MAX - 4, PP_MAX - 4);
        box[k] = make_int4(rmin - 2, cmin - 2, h, w);
    }
    DevBuf *S_ = ctx->scr;
    DevBuf &dColptr = S_[0], &dErow = S_[1], &dAval = S_[6], &dBox = S_[13], &dKeep = S_[14];
    RET(to_dev(ctx, dColptr, A_colptr, (size_t)K + 1));
    RET(to_dev(ctx, dErow, A_rowidx, (size_t)nnz));
    RET(to_dev(ctx, dAval, A_val, (size_t)nnz));
    RET(to_dev(ctx, dBox, box.data(), box.size()));
    RET(dKeep.ensure((size_t)nnz));
    CK(hipMemsetAsync(dKeep.p, 0, (size_t)nnz, ctx->st()));
    LAUNCH(ctx, "spatial_connectivity", k_connectivity, dim3(K), dim3(256), 0, dColptr.as<int64_t>(), dErow.as<int>(), dAval.as<float>(),
           dBox.as<int4>(), d1, d2, dKeep.as<unsigned char>());
    CK(hipMemcpyAsync(keep, dKeep.p, (size_t)nnz, hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    return 0;
}

// loads this translation unit's code object now (HIP loads it at the first launch of one of its kernels -- milliseconds each that would otherwise fall into the first iteration): cnmfe_create
int tu_warm_factor() { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)k_reduce_parts) == hipSuccess ? 0 : -1; }

}  // namespace cnmfe
