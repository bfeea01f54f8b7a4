// Stage-by-stage device check of cnmf_e_amd/csrc/ring_solve_core.hpp against host fp64 references:
//   1. rs_diag_block on random SPD 16x16 blocks  (inv(chol(B)))
//   2. the MFMA operand algebra: D = X1' X2 from accumulator-layout registers
//   3. rs_solve_core<NT> on random bordered systems handed over as dense matrices (no table gather)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cnmf_e_amd/csrc scripts/probes/rs_test.hip -o /tmp/rs_test && /tmp/rs_test
#include "ring_solve_core.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using namespace cnmfe;

#define HCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_diag(const double *in, double *out) {
    __shared__ __attribute__((aligned(16))) double sb[16 * RS_DS];
    const int lane = threadIdx.x;
    const double *B = in + (size_t)blockIdx.x * 256;
    for (int e = lane; e < 256; e += 64) sb[(e / 16) * RS_DS + (e % 16)] = B[e];
    __syncthreads();
    rs_diag_block(sb, lane);
    for (int e = lane; e < 256; e += 64) out[(size_t)blockIdx.x * 256 + e] = sb[(e / 16) * RS_DS + (e % 16)];
}

__global__ void __launch_bounds__(64) k_mfma(const double *X1, const double *X2, double *D) {
    const int lane = threadIdx.x, c = lane & 15, rq = lane >> 4;
    double4_t a, b, acc = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) { a[r] = X1[(rq + 4 * r) * 16 + c]; b[r] = X2[(rq + 4 * r) * 16 + c]; }
    acc = rs_mfma4(a, b, acc);
    for (int r = 0; r < 4; ++r) D[(rq + 4 * r) * 16 + c] = acc[r];
}

template <int NT, bool LOOP>
__global__ void __launch_bounds__(64, (NT <= 2 ? 4 : (NT <= 3 ? 3 : (NT <= 6 ? 2 : 1))))
k_solve(const double *G, const double *u, const double *g, const double *sc_lam_tp, double *w, int probe) {
    constexpr int N = 16 * NT, NTILE = (NT * (NT + 1)) / 2;
    __shared__ __attribute__((aligned(16))) double s_vec[3][N];
    __shared__ __attribute__((aligned(16))) double s_blk[16 * RS_DS];
    __shared__ __attribute__((aligned(16))) double s_part[4][64];
    const int lane = threadIdx.x, c = lane & 15, rq = lane >> 4;
    const size_t sys = blockIdx.x;
    const double *Gs = G + sys * N * N;
    for (int a = lane; a < N; a += 64) { s_vec[0][a] = u[sys * N + a]; s_vec[1][a] = g[sys * N + a]; }
    double4_t T[NTILE];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[rs_tix(I, J)][r] = Gs[(size_t)(16 * I + c) * N + 16 * J + rq + 4 * r];
    __syncthreads();
    double wc[NT];
    rs_solve_core<NT, LOOP>(T, s_vec, s_blk, s_part, sc_lam_tp[sys * 3], sc_lam_tp[sys * 3 + 1], sc_lam_tp[sys * 3 + 2], lane, probe, wc);
    if (rq == 0)
#pragma unroll
        for (int k = 0; k < NT; ++k) w[sys * N + 16 * k + c] = wc[k];
}

static void host_solve(int n, std::vector<double> S, std::vector<double> b, std::vector<double> &x) {   // Gaussian elimination, partial pivoting
    for (int k = 0; k < n; ++k) {
        int pv = k; for (int i = k + 1; i < n; ++i) if (fabs(S[i * n + k]) > fabs(S[pv * n + k])) pv = i;
        if (pv != k) { for (int j = 0; j < n; ++j) std::swap(S[k * n + j], S[pv * n + j]); std::swap(b[k], b[pv]); }
        for (int i = k + 1; i < n; ++i) { const double f = S[i * n + k] / S[k * n + k]; for (int j = k; j < n; ++j) S[i * n + j] -= f * S[k * n + j]; b[i] -= f * b[k]; }
    }
    x.assign(n, 0.0);
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int j = i + 1; j < n; ++j) s -= S[i * n + j] * x[j]; x[i] = s / S[i * n + i]; }
}

template <int NT, bool LOOP> static int run_solve(int p, int nsys, std::mt19937_64 &rng) {
    const int N = 16 * NT, Tn = 400;
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> G((size_t)nsys * N * N, 0.0), u((size_t)nsys * N, 0.0), g((size_t)nsys * N, 0.0), slt((size_t)nsys * 3), ref((size_t)nsys * N, 0.0);
    for (int s = 0; s < nsys; ++s) {
        std::vector<double> X((size_t)p * Tn), y(Tn);
        for (auto &v : X) v = nd(rng); for (auto &v : y) v = nd(rng);
        std::vector<char> miss(p, 0); for (int a = 0; a < p; ++a) miss[a] = (rng() % 10) == 0;
        double *Gs = &G[(size_t)s * N * N];
        for (int a = 0; a < N; ++a) Gs[a * N + a] = 1.0;
        double tr = Tn, sc = 0; for (int t = 0; t < Tn; ++t) sc += y[t];
        for (int a = 0; a < p; ++a) {
            if (miss[a]) continue;
            for (int b = 0; b < p; ++b) { if (miss[b]) continue; double v = 0; for (int t = 0; t < Tn; ++t) v += X[(size_t)a * Tn + t] * X[(size_t)b * Tn + t]; Gs[a * N + b] = v; }
            double su = 0, sg = 0; for (int t = 0; t < Tn; ++t) { su += X[(size_t)a * Tn + t]; sg += X[(size_t)a * Tn + t] * y[t]; }
            u[(size_t)s * N + a] = su; g[(size_t)s * N + a] = sg; tr += Gs[a * N + a];
        }
        const double lam = 1e-5 * tr;
        for (int a = 0; a < p; ++a) if (!miss[a]) Gs[a * N + a] += lam;
        slt[s * 3] = sc; slt[s * 3 + 1] = lam; slt[s * 3 + 2] = Tn;
        const int n = p + 1;
        std::vector<double> S((size_t)n * n, 0.0), b(n), x;
        for (int a = 0; a < p; ++a) { for (int bb = 0; bb < p; ++bb) S[a * n + bb] = Gs[a * N + bb]; S[a * n + p] = S[p * n + a] = u[(size_t)s * N + a]; b[a] = g[(size_t)s * N + a]; }
        S[p * n + p] = Tn + lam; b[p] = sc;
        host_solve(n, S, b, x);
        for (int a = 0; a < p; ++a) ref[(size_t)s * N + a] = x[a];
    }
    double *dG, *du, *dg, *ds, *dw;
    HCK(hipMalloc(&dG, G.size() * 8)); HCK(hipMalloc(&du, u.size() * 8)); HCK(hipMalloc(&dg, g.size() * 8)); HCK(hipMalloc(&ds, slt.size() * 8)); HCK(hipMalloc(&dw, u.size() * 8));
    HCK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice)); HCK(hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice));
    HCK(hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice)); HCK(hipMemcpy(ds, slt.data(), slt.size() * 8, hipMemcpyHostToDevice));
    HCK(hipMemset(dw, 0, u.size() * 8));
    hipLaunchKernelGGL((k_solve<NT, LOOP>), dim3(nsys), dim3(64), 0, 0, dG, du, dg, ds, dw, 0);
    HCK(hipDeviceSynchronize());
    std::vector<double> w(u.size());
    HCK(hipMemcpy(w.data(), dw, w.size() * 8, hipMemcpyDeviceToHost));
    double err = 0, mx = 0; int nan = 0;
    for (size_t i = 0; i < w.size(); ++i) { if (w[i] != w[i]) ++nan; else err = fmax(err, fabs(w[i] - ref[i])); mx = fmax(mx, fabs(ref[i])); }
    printf("solve NT=%d loop=%d p=%d: max |w - ref| / max|ref| = %.3e  (nan %d)  %s\n", NT, (int)LOOP, p, err / mx, nan, (nan == 0 && err / mx < 1e-9) ? "OK" : "FAIL");
    (void)hipFree(dG); (void)hipFree(du); (void)hipFree(dg); (void)hipFree(ds); (void)hipFree(dw);
    return 0;
}

int main() {
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    // ---- 1. diagonal step ----
    {
        const int nb = 64;
        std::vector<double> B((size_t)nb * 256), ref((size_t)nb * 256, 0.0), out((size_t)nb * 256);
        for (int s = 0; s < nb; ++s) {
            double X[16][40];
            for (auto &r : X) for (auto &v : r) v = nd(rng);
            double *b = &B[(size_t)s * 256];
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = 0; for (int t = 0; t < 40; ++t) v += X[i][t] * X[j][t]; b[i * 16 + j] = v + (i == j ? 0.5 : 0.0); }
            double Lm[16][16] = {};
            for (int j = 0; j < 16; ++j) {
                double d = b[j * 16 + j]; for (int k = 0; k < j; ++k) d -= Lm[j][k] * Lm[j][k];
                Lm[j][j] = sqrt(d);
                for (int i = j + 1; i < 16; ++i) { double v = b[i * 16 + j]; for (int k = 0; k < j; ++k) v -= Lm[i][k] * Lm[j][k]; Lm[i][j] = v / Lm[j][j]; }
            }
            double *Y = &ref[(size_t)s * 256];                       // Y = inv(L), forward substitution per column
            for (int cc = 0; cc < 16; ++cc)
                for (int i = 0; i < 16; ++i) { double v = (i == cc) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) v -= Lm[i][k] * Y[k * 16 + cc]; Y[i * 16 + cc] = v / Lm[i][i]; }
        }
        double *dB, *dO;
        HCK(hipMalloc(&dB, B.size() * 8)); HCK(hipMalloc(&dO, B.size() * 8));
        HCK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_diag, dim3(nb), dim3(64), 0, 0, dB, dO);
        HCK(hipDeviceSynchronize());
        HCK(hipMemcpy(out.data(), dO, out.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, mx = 0; int nan = 0;
        for (size_t i = 0; i < out.size(); ++i) { if (out[i] != out[i]) ++nan; else err = fmax(err, fabs(out[i] - ref[i])); mx = fmax(mx, fabs(ref[i])); }
        printf("diag block: max |inv(L) - ref| / max|ref| = %.3e  (nan %d)  %s\n", err / mx, nan, (nan == 0 && err / mx < 1e-11) ? "OK" : "FAIL");
        if (nan || err / mx >= 1e-11) {
            printf("  first block, device vs ref (rows 0..3):\n");
            for (int i = 0; i < 4; ++i) { for (int j = 0; j < 6; ++j) printf(" %9.4f/%9.4f", out[i * 16 + j], ref[i * 16 + j]); printf("\n"); }
        }
    }
    // ---- 2. MFMA operand algebra ----
    {
        std::vector<double> X1(256), X2(256), D(256);
        for (auto &v : X1) v = nd(rng); for (auto &v : X2) v = nd(rng);
        double *d1, *d2, *dD;
        HCK(hipMalloc(&d1, 2048)); HCK(hipMalloc(&d2, 2048)); HCK(hipMalloc(&dD, 2048));
        HCK(hipMemcpy(d1, X1.data(), 2048, hipMemcpyHostToDevice)); HCK(hipMemcpy(d2, X2.data(), 2048, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, d1, d2, dD);
        HCK(hipDeviceSynchronize());
        HCK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
        double err = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = 0; for (int k = 0; k < 16; ++k) v += X1[k * 16 + i] * X2[k * 16 + j]; err = fmax(err, fabs(v - D[i * 16 + j])); }
        printf("mfma X1' X2: max err %.3e  %s\n", err, err < 1e-12 ? "OK" : "FAIL");
    }
    // ---- 3. the whole core ----
    run_solve<1, false>(16, 8, rng); run_solve<1, true>(7, 8, rng);
    run_solve<3, false>(40, 8, rng); run_solve<3, true>(40, 8, rng);
    run_solve<6, false>(96, 16, rng); run_solve<6, true>(96, 16, rng);
    run_solve<8, false>(116, 8, rng); run_solve<8, true>(120, 8, rng);
    return 0;
}
