// bg_ssub > 1: the ring model on a spatially downsampled block.
//   @Sources2D/initComponents_parallel.m:237-251      W on the ceil(nr_b/s) x ceil(nc_b/s) grid, ring radius ceil(r/s)
//   @Sources2D/update_background_parallel.m:219-230   b0 = mean(Y - A*C); W fitted on imresize(., 1/s, 'nearest')
//   @Sources2D/update_spatial_parallel.m:167-178      Ysig = Y(patch) - up(W * down(R - mean R))(patch) - b0
//   (= update_temporal_parallel.m:153-165)            down = imresize(., 1/s) (bicubic, antialiased), up = imresize(., [nr_b nc_b])
//
// Design: both resizes are fixed linear maps, so the downsampled data are resident videos of their own.  Every real patch
// gets two DERIVED patches whose "FOV" is the low-resolution block (patch == block, so the ring is clipped at the block
// exactly like initComponents_parallel.m:246 does):
//   fit patch:  nearest-neighbour samples of the centred video  -> cnmfe_fit_ring_model's kernels run on it unchanged
//   res patch:  bicubic-antialiased downsample of the centred video, mean 0, b0 0 -> the R1 kernels run on it unchanged and
//               give  down(Y') - W * down(R - mean R);  A_prev enters as down(A_prev) (linearity), built on the host.
// What is new here is only the resampling: imresize's `contributions` restated from MathWorks' documentation (the source is
// not in the reference; PARITY UNPINNED like every toolbox function), one gather kernel, one separable
// downsample kernel (set-up time) and a two-pass separable upsample fused with the final combination (per call).
#include "common.hpp"
#include <cmath>
#include <climits>

namespace cnmfe {

// ---- imresize contributions -----------------------------------------------------------------------------
struct Taps {
    int n_in = 0, n_out = 0, P = 0;
    std::vector<int> idx;      // [n_out * P] source index, mirrored into 0..n_in-1
    std::vector<float> w;      // [n_out * P] weights, rows normalised to 1
};

static double cubic_k(double x) {
    const double ax = std::fabs(x), ax2 = ax * ax, ax3 = ax2 * ax;
    if (ax <= 1) return 1.5 * ax3 - 2.5 * ax2 + 1;
    if (ax <= 2) return -0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2;
    return 0.0;
}

static Taps make_taps(int n_in, int n_out, double scale, bool nearest) {
    Taps t; t.n_in = n_in; t.n_out = n_out;
    double kw = nearest ? 1.0 : 4.0;
    const bool stretch = !nearest && scale < 1.0;            // antialiasing: kernel stretched by 1/scale when shrinking
    if (stretch) kw /= scale;
    t.P = (int)std::ceil(kw) + 2;
    t.idx.resize((size_t)n_out * t.P); t.w.resize((size_t)n_out * t.P);
    std::vector<double> wd(t.P);
    for (int o = 0; o < n_out; ++o) {
        const double x = o + 1.0, u = x / scale + 0.5 * (1.0 - 1.0 / scale);
        const double left = std::floor(u - kw / 2);
        double sum = 0;
        for (int i = 0; i < t.P; ++i) {
            const double d = u - (left + i);
            double h;
            if (nearest) h = (d >= -0.5 && d < 0.5) ? 1.0 : 0.0;
            else h = stretch ? scale * cubic_k(scale * d) : cubic_k(d);
            wd[i] = h; sum += h;
        }
        for (int i = 0; i < t.P; ++i) {
            long v = (long)left + i - 1;                          // 0-based virtual index
            const long per = 2L * n_in;
            long mth = ((v % per) + per) % per;                   // aux = [1:n n:-1:1]
            const int src = (int)(mth < n_in ? mth : per - 1 - mth);
            t.idx[(size_t)o * t.P + i] = src; t.w[(size_t)o * t.P + i] = (float)(wd[i] / sum);
        }
    }
    return t;
}

// ---- kernels --------------------------------------------------------------------------------------------
// derived video by pixel selection (imresize 'nearest'): dst[c][q] = src[c][sel[q]]
__global__ void __launch_bounds__(256) k_derive_gather(const float4 *__restrict__ src, int64_t d_src, const int *__restrict__ sel,
                                                       float4 *__restrict__ dst, int64_t d_dst) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (q < d_dst) dst[c * d_dst + q] = src[c * d_src + sel[q]];
}
__global__ void __launch_bounds__(256) k_gather_mean(const double *__restrict__ src, const int *__restrict__ sel, double *__restrict__ dd,
                                                     float *__restrict__ df, int64_t n) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q < n) { const double v = src[sel[q]]; dd[q] = v; df[q] = (float)v; }
}
// separable resample of a column-major image per 4-frame group: dst(ro,co) = sum_i sum_j wr[ro][i] wc[co][j] src(ir[ro][i], ic[co][j])
__global__ void __launch_bounds__(256) k_resample2d(const float4 *__restrict__ src, int nr_in, int64_t d_in, float4 *__restrict__ dst, int nr_out,
                                                    int64_t d_out, const int *__restrict__ ir, const float *__restrict__ wr, int Pr,
                                                    const int *__restrict__ ic, const float *__restrict__ wc, int Pc) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (q >= d_out) return;
    const int ro = (int)(q % nr_out), co = (int)(q / nr_out);
    const float4 *s = src + c * d_in;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Pc; ++j) {
        const float wj = wc[co * Pc + j];
        if (wj == 0.f) continue;
        const float4 *col = s + (int64_t)ic[co * Pc + j] * nr_in;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < Pr; ++i) {
            const float wi = wr[ro * Pr + i];
            const float4 v = col[ir[ro * Pr + i]];
            t.x = fmaf(wi, v.x, t.x); t.y = fmaf(wi, v.y, t.y); t.z = fmaf(wi, v.z, t.z); t.w = fmaf(wi, v.w, t.w);
        }
        acc.x = fmaf(wj, t.x, acc.x); acc.y = fmaf(wj, t.y, acc.y); acc.z = fmaf(wj, t.z, acc.z); acc.w = fmaf(wj, t.w, acc.w);
    }
    dst[c * d_out + q] = acc;
}
// upsample pass A (columns): tmp(rl, cb) = sum_j wc[cb][j] * (Ylow - YsigLow)(rl, ic[cb][j])      [low rows x full columns]
__global__ void __launch_bounds__(256) k_up_cols(const float4 *__restrict__ ylow, const float4 *__restrict__ yslow, int d1s, int64_t d_low,
                                                 float4 *__restrict__ tmp, int nc_b, const int *__restrict__ ic, const float *__restrict__ wc, int Pc) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    const int64_t n = (int64_t)d1s * nc_b;
    if (q >= n) return;
    const int rl = (int)(q % d1s), cb = (int)(q / d1s);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (Pc == 6) {                                               // bicubic upsampling: 6 taps; all 12 loads in flight at once
        int id[6]; float w[6]; float4 a[6], b[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { id[j] = ic[cb * 6 + j]; w[j] = wc[cb * 6 + j]; }
#pragma unroll
        for (int j = 0; j < 6; ++j) { const int64_t o = c * d_low + (int64_t)id[j] * d1s + rl; a[j] = ylow[o]; b[j] = yslow[o]; }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            acc.x = fmaf(w[j], a[j].x - b[j].x, acc.x); acc.y = fmaf(w[j], a[j].y - b[j].y, acc.y);
            acc.z = fmaf(w[j], a[j].z - b[j].z, acc.z); acc.w = fmaf(w[j], a[j].w - b[j].w, acc.w);
        }
        tmp[c * n + q] = acc;
        return;
    }
    for (int j = 0; j < Pc; ++j) {
        const float wj = wc[cb * Pc + j];
        if (wj == 0.f) continue;
        const int64_t o = c * d_low + (int64_t)ic[cb * Pc + j] * d1s + rl;
        const float4 a = ylow[o], b = yslow[o];                  // W * (...) = Ylow' - YsigLow (the res patch has mean 0 and b0 0)
        acc.x = fmaf(wj, a.x - b.x, acc.x); acc.y = fmaf(wj, a.y - b.y, acc.y); acc.z = fmaf(wj, a.z - b.z, acc.z); acc.w = fmaf(wj, a.w - b.w, acc.w);
    }
    tmp[c * n + q] = acc;
}
// upsample pass B (rows) + combination: Ysig(m) = Y'(centre) + (Ymean - b0) - sum_i wr[rb][i] * tmp(ir[rb][i], cb)
__global__ void __launch_bounds__(256) k_up_rows_combine(const float4 *__restrict__ tmp, int d1s, int nc_b, const float4 *__restrict__ Y4, int64_t d_b,
                                                         int nr_b, int nr, int roff, int coff, int64_t d, const float *__restrict__ dlt,
                                                         const int *__restrict__ ir, const float *__restrict__ wr, int Pr, float4 *__restrict__ ysig) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (m >= d) return;
    const int rb = (int)(m % nr) + roff, cb = (int)(m / nr) + coff;
    const float4 *col = tmp + c * (int64_t)d1s * nc_b + (int64_t)cb * d1s;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (Pr == 6) {
        int id[6]; float w[6]; float4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { id[i] = ir[rb * 6 + i]; w[i] = wr[rb * 6 + i]; }
        const float4 y = Y4[c * d_b + (int64_t)cb * nr_b + rb];
        const float dl = dlt[m];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = col[id[i]];
#pragma unroll
        for (int i = 0; i < 6; ++i) { acc.x = fmaf(w[i], v[i].x, acc.x); acc.y = fmaf(w[i], v[i].y, acc.y); acc.z = fmaf(w[i], v[i].z, acc.z); acc.w = fmaf(w[i], v[i].w, acc.w); }
        ysig[c * d + m] = make_float4(y.x + dl - acc.x, y.y + dl - acc.y, y.z + dl - acc.z, y.w + dl - acc.w);
        return;
    }
    for (int i = 0; i < Pr; ++i) {
        const float wi = wr[rb * Pr + i];
        if (wi == 0.f) continue;
        const float4 v = col[ir[rb * Pr + i]];
        acc.x = fmaf(wi, v.x, acc.x); acc.y = fmaf(wi, v.y, acc.y); acc.z = fmaf(wi, v.z, acc.z); acc.w = fmaf(wi, v.w, acc.w);
    }
    const float4 y = Y4[c * d_b + (int64_t)cb * nr_b + rb];
    const float dl = dlt[m];
    ysig[c * d + m] = make_float4(y.x + dl - acc.x, y.y + dl - acc.y, y.z + dl - acc.z, y.w + dl - acc.w);
}
// upsample + combination in ONE pass (bicubic, 6 taps per dimension): a workgroup owns 64 x 16 patch pixels, stages the low-
// resolution window of W*(...) = Ylow' - YsigLow that feeds them in LDS (<= 48 x 16 low pixels), interpolates the 16 output
// columns for every low row into a second LDS tile, then each thread finishes 4 pixels of one output row.  The two-pass version
// above re-reads every low pixel 6 + 6 times through L2 (12 ms per call at 512x512x10000); this one reads it ~1.1 times.
constexpr int UF_TR = 64, UF_TC = 16, UF_NLR = 48, UF_NLC = 16;
__global__ void __launch_bounds__(256) k_up_fused(const float4 *__restrict__ ylow, const float4 *__restrict__ yslow, int d1s, int64_t d_low,
                                                  const float4 *__restrict__ Y4, int64_t d_b, int nr_b, int nr, int nc, int roff, int coff, int64_t d,
                                                  const float *__restrict__ dlt, const int *__restrict__ ir, const float *__restrict__ wr,
                                                  const int *__restrict__ ic, const float *__restrict__ wc, int ntile_r, int64_t Tc, int cseg,
                                                  float4 *__restrict__ ysig) {
    __shared__ __attribute__((aligned(16))) float4 low[UF_NLC][UF_NLR];
    __shared__ __attribute__((aligned(16))) float4 mid[UF_TC][UF_NLR];
    const int tid = threadIdx.x;
    const int tile_r = blockIdx.x % ntile_r, tile_c = blockIdx.x / ntile_r;
    const int pr0 = tile_r * UF_TR, pc0 = tile_c * UF_TC;
    const int nrow = min(UF_TR, nr - pr0), ncol = min(UF_TC, nc - pc0);
    // low-resolution window of this tile (taps are mirrored at the borders, so take min / max over all of them)
    int lo_r = 1 << 30, hi_r = -1, lo_c = 1 << 30, hi_c = -1;
    for (int q = 0; q < nrow * 6; ++q) { const int v = ir[(pr0 + roff) * 6 + q]; lo_r = min(lo_r, v); hi_r = max(hi_r, v); }
    for (int q = 0; q < ncol * 6; ++q) { const int v = ic[(pc0 + coff) * 6 + q]; lo_c = min(lo_c, v); hi_c = max(hi_c, v); }
    const int nlr = hi_r - lo_r + 1, nlc = hi_c - lo_c + 1;           // host guarantees <= UF_NLR, UF_NLC
    // this thread's output row and its six row taps
    const int orow = tid & 63, ocg = tid >> 6;                        // pixels (orow, ocg + 4k), k = 0..3
    const bool rvalid = orow < nrow;
    int rid[6]; float rw[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { rid[i] = rvalid ? ir[(pr0 + orow + roff) * 6 + i] - lo_r : 0; rw[i] = rvalid ? wr[(pr0 + orow + roff) * 6 + i] : 0.f; }
    const int64_t c0 = (int64_t)blockIdx.y * cseg, c1 = min(Tc, c0 + cseg);
    for (int64_t c = c0; c < c1; ++c) {
        for (int q = tid; q < nlr * nlc; q += 256) {
            const int lr = q % nlr, lc = q / nlr;
            const int64_t o = c * d_low + (int64_t)(lo_c + lc) * d1s + lo_r + lr;
            const float4 a = ylow[o], b = yslow[o];
            low[lc][lr] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        }
        __syncthreads();
        for (int q = tid; q < ncol * nlr; q += 256) {
            const int lr = q % nlr, oc = q / nlr;
            const int cb = pc0 + oc + coff;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float w = wc[cb * 6 + j];
                const float4 v = low[ic[cb * 6 + j] - lo_c][lr];
                acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
            }
            mid[oc][lr] = acc;
        }
        __syncthreads();
        if (rvalid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int oc = ocg + 4 * k;
                if (oc < ncol) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const float4 v = mid[oc][rid[i]];
                        acc.x = fmaf(rw[i], v.x, acc.x); acc.y = fmaf(rw[i], v.y, acc.y); acc.z = fmaf(rw[i], v.z, acc.z); acc.w = fmaf(rw[i], v.w, acc.w);
                    }
                    const int pr = pr0 + orow, pc = pc0 + oc;
                    const int64_t m = (int64_t)pc * nr + pr;
                    const float4 y = Y4[c * d_b + (int64_t)(pc + coff) * nr_b + pr + roff];
                    const float dl = dlt[m];
                    ysig[c * d + m] = make_float4(y.x + dl - acc.x, y.y + dl - acc.y, y.z + dl - acc.z, y.w + dl - acc.w);
                }
            }
        }
        __syncthreads();
    }
}

__global__ void k_dlt2(const float *__restrict__ ymean_f, const double *__restrict__ b0, float *__restrict__ dlt, int64_t d, int nr, int nr_b, int roff, int coff) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= d) return;
    const int64_t q = (int64_t)((int)(m / nr) + coff) * nr_b + (int)(m % nr) + roff;
    dlt[m] = (float)((double)ymean_f[q] - b0[m]);
}

static void low_dims(const Patch *P, int ssub, int &d1s, int &d2s) { d1s = (P->nr_b + ssub - 1) / ssub; d2s = (P->nc_b + ssub - 1) / ssub; }

int ssub_derive(cnmfe_ctx *ctx, Patch *S, int dst_id, int ssub, int mode) {
    RET(ensure_ymean(ctx, S));
    int d1s, d2s; low_dims(S, ssub, d1s, d2s);
    if (ctx->patches.count(dst_id)) { delete ctx->patches[dst_id]; ctx->patches.erase(dst_id); }
    Patch *P = new Patch();
    P->derived = true; P->lane = S->lane;                   // (the low-resolution patches run on their source's lane: the calls that name two of them see one scratch set)
    const int32_t rect[4] = {1, d1s, 1, d2s};
    memcpy(P->prect, rect, sizeof(rect)); memcpy(P->brect, rect, sizeof(rect));
    P->d1 = d1s; P->d2 = d2s; P->T = S->T; P->Tc = S->Tc;
    P->nr = P->nr_b = d1s; P->nc = P->nc_b = d2s; P->roff = P->coff = 0;
    P->d = P->d_b = (int64_t)d1s * d2s;
    ctx->patches[dst_id] = P;
    RET(P->Yc4.ensure((size_t)P->Tc * P->d_b * sizeof(float4)));
    RET(P->ymean_d.ensure(P->d_b * sizeof(double)));
    RET(P->ymean_f.ensure(P->d_b * sizeof(float)));
    const double scale = 1.0 / ssub;
    Taps tr = make_taps(S->nr_b, d1s, scale, mode == CNMFE_DERIVE_NEAREST), tc = make_taps(S->nc_b, d2s, scale, mode == CNMFE_DERIVE_NEAREST);
    dim3 grid((unsigned)((P->d_b + 255) / 256), (unsigned)P->Tc);
    if (mode == CNMFE_DERIVE_NEAREST) {
        std::vector<int> sel((size_t)P->d_b);
        for (int co = 0; co < d2s; ++co)
            for (int ro = 0; ro < d1s; ++ro) {
                int ri = 0, ci = 0;
                for (int i = 0; i < tr.P; ++i) if (tr.w[(size_t)ro * tr.P + i] != 0.f) ri = tr.idx[(size_t)ro * tr.P + i];
                for (int j = 0; j < tc.P; ++j) if (tc.w[(size_t)co * tc.P + j] != 0.f) ci = tc.idx[(size_t)co * tc.P + j];
                sel[(size_t)co * d1s + ro] = ci * S->nr_b + ri;
            }
        DevBuf dSel;
        RET(to_dev(ctx, dSel, sel.data(), sel.size()));
        LAUNCH(ctx, "ssub_derive_gather", k_derive_gather, grid, dim3(256), 0, S->Yc4.as<float4>(), S->d_b, dSel.as<int>(), P->Yc4.as<float4>(), P->d_b);
        LAUNCH(ctx, "ssub_gather_mean", k_gather_mean, dim3((unsigned)((P->d_b + 255) / 256)), dim3(256), 0, S->ymean_d.as<double>(), dSel.as<int>(),
               P->ymean_d.as<double>(), P->ymean_f.as<float>(), P->d_b);
        CK(hipStreamSynchronize(ctx->st()));
    } else {
        DevBuf dIr, dWr, dIc, dWc;
        RET(to_dev(ctx, dIr, tr.idx.data(), tr.idx.size())); RET(to_dev(ctx, dWr, tr.w.data(), tr.w.size()));
        RET(to_dev(ctx, dIc, tc.idx.data(), tc.idx.size())); RET(to_dev(ctx, dWc, tc.w.data(), tc.w.size()));
        LAUNCH(ctx, "ssub_derive_bicubic", k_resample2d, grid, dim3(256), 0, S->Yc4.as<float4>(), S->nr_b, S->d_b, P->Yc4.as<float4>(), d1s, P->d_b,
               dIr.as<int>(), dWr.as<float>(), tr.P, dIc.as<int>(), dWc.as<float>(), tc.P);
        CK(hipMemsetAsync(P->ymean_d.p, 0, P->d_b * sizeof(double), ctx->st()));
        CK(hipMemsetAsync(P->ymean_f.p, 0, P->d_b * sizeof(float), ctx->st()));
        CK(hipStreamSynchronize(ctx->st()));
    }
    P->ymean_valid = true; P->frames_uploaded = P->T;
    return 0;
}

// A restricted to the sampled pixels (nearest) or pushed through the bicubic downsample (linear map): d_low x K CSC.
// Round 5: on the host's critical path once the sweep left the iteration (1.9 ms per call at H) -- 'nearest' is a filter through two look-up tables, the bicubic map
// accumulates every column in the dense low-resolution box of its taps (flat transposed taps; the same sums in the same order as the first version's hash-and-sort).
static void a_low(const Patch *S, int ssub, bool nearest, int32_t K, const int64_t *cp, const int32_t *ri, const float *va,
                  std::vector<int64_t> &ocp, std::vector<int32_t> &ori, std::vector<float> &ova) {
    int d1s, d2s; low_dims(S, ssub, d1s, d2s);
    Taps tr = make_taps(S->nr_b, d1s, 1.0 / ssub, nearest), tc = make_taps(S->nc_b, d2s, 1.0 / ssub, nearest);
    ocp.assign(K + 1, 0); ori.clear(); ova.clear();
    ori.reserve((size_t)cp[K] / (nearest ? 2 : 1) + 16); ova.reserve((size_t)cp[K] / (nearest ? 2 : 1) + 16);
    if (nearest) {
        // (the weight of the one kept tap is 1: rows normalised; an input index sampled by no output is dropped)
        std::vector<int> rlow((size_t)S->nr_b, -1), clow((size_t)S->nc_b, -1);
        bool simple = true;
        auto fill = [&](const Taps &t, std::vector<int> &low) {
            for (int o = 0; o < t.n_out; ++o)
                for (int i = 0; i < t.P; ++i) if (t.w[(size_t)o * t.P + i] != 0.f) { int &l = low[t.idx[(size_t)o * t.P + i]]; if (l >= 0 || t.w[(size_t)o * t.P + i] != 1.f) simple = false; l = o; }
        };
        fill(tr, rlow); fill(tc, clow);
        if (simple) {
            for (int32_t k = 0; k < K; ++k) {
                for (int64_t e = cp[k]; e < cp[k + 1]; ++e) {
                    const int pr = rlow[ri[e] % S->nr_b], pc = clow[ri[e] / S->nr_b];
                    if (pr >= 0 && pc >= 0 && va[e] != 0.f) { ori.push_back(pc * d1s + pr); ova.push_back(va[e]); }
                }
                ocp[k + 1] = (int64_t)ori.size();
            }
            return;
        }
    }
    // transposed taps, flat: for every input index the (output, weight) pairs it feeds, outputs ascending
    auto transpose = [](const Taps &t, std::vector<int> &ptr, std::vector<int> &out, std::vector<float> &w) {
        ptr.assign((size_t)t.n_in + 1, 0);
        for (int o = 0; o < t.n_out; ++o) for (int i = 0; i < t.P; ++i) if (t.w[(size_t)o * t.P + i] != 0.f) ++ptr[t.idx[(size_t)o * t.P + i] + 1];
        for (int j = 0; j < t.n_in; ++j) ptr[j + 1] += ptr[j];
        out.resize((size_t)ptr[t.n_in]); w.resize((size_t)ptr[t.n_in]);
        std::vector<int> cur(ptr.begin(), ptr.end() - 1);
        for (int o = 0; o < t.n_out; ++o) for (int i = 0; i < t.P; ++i) { const float v = t.w[(size_t)o * t.P + i]; if (v != 0.f) { const int q = cur[t.idx[(size_t)o * t.P + i]]++; out[q] = o; w[q] = v; } }
    };
    std::vector<int> rp, ro, cq, co; std::vector<float> rw, cw;
    transpose(tr, rp, ro, rw); transpose(tc, cq, co, cw);
    std::vector<double> box;
    for (int32_t k = 0; k < K; ++k) {
        int r0 = INT_MAX, r1 = -1, c0 = INT_MAX, c1 = -1;              // the low-resolution box of this column's taps
        for (int64_t e = cp[k]; e < cp[k + 1]; ++e) {
            const int r = ri[e] % S->nr_b, c = ri[e] / S->nr_b;
            if (rp[r + 1] > rp[r]) { r0 = std::min(r0, ro[rp[r]]); r1 = std::max(r1, ro[rp[r + 1] - 1]); }
            if (cq[c + 1] > cq[c]) { c0 = std::min(c0, co[cq[c]]); c1 = std::max(c1, co[cq[c + 1] - 1]); }
        }
        if (r1 >= r0 && c1 >= c0) {
            const int h = r1 - r0 + 1, w = c1 - c0 + 1;
            box.assign((size_t)h * w, 0.0);
            for (int64_t e = cp[k]; e < cp[k + 1]; ++e) {
                const int r = ri[e] % S->nr_b, c = ri[e] / S->nr_b;
                const float v = va[e];
                for (int a = rp[r]; a < rp[r + 1]; ++a) {
                    const float wr_ = rw[a];
                    double *col0 = box.data() + (ro[a] - r0);
                    for (int b = cq[c]; b < cq[c + 1]; ++b) col0[(size_t)(co[b] - c0) * h] += (double)wr_ * cw[b] * v;
                }
            }
            for (int c = 0; c < w; ++c)
                for (int r = 0; r < h; ++r) { const double x = box[(size_t)c * h + r]; if (x != 0.0) { ori.push_back((c0 + c) * d1s + (r0 + r)); ova.push_back((float)x); } }
        }
        ocp[k + 1] = (int64_t)ori.size();
    }
}

int ssub_fit(cnmfe_ctx *ctx, Patch *M, Patch *F, Patch *R, int ssub, int32_t K, const int64_t *cp, const int32_t *ri, const float *va,
             const float *C, int c_order, int with_projection, int64_t info[4], double thresh_outlier) {
    std::vector<int64_t> ocp; std::vector<int32_t> ori; std::vector<float> ova;
    ctx->bgs_patch = -1;                                   // W changes: a kept W*R_low (cnmfe_background_ssub) is stale
    HostTrace ht(ctx, "fit_ssub");
    if (K > 0) a_low(M, ssub, true, K, cp, ri, va, ocp, ori, ova);
    ht.mark("nearest(A)");
    // b0 = mean(Y - A*C, 2) on the patch pixels (:222-223) = Ymean - A*mean(C): independent of W, so it goes first (a short, synchronous call)
    RET(bg_fit_ring(ctx, M, K, cp, ri, va, C, c_order, with_projection, nullptr, nullptr, /*b0_only=*/1));
    ht.mark("b0");
    // W: fit_ring_model(imresize(Y - A*C, 1/s, 'nearest'), [], [], W_old, ...)  (update_background_parallel.m:224-227).  The call returns with the
    // Gram / solve kernels in flight (its host staging has been consumed); the copy of W to the residual patch is ordered behind them.
    RET(bg_fit_ring(ctx, F, K, K > 0 ? ocp.data() : nullptr, ori.data(), ova.data(), C, c_order, with_projection, nullptr, info, /*A = [] for ind_active*/ 2,
                    thresh_outlier));          // sn of the low-resolution block: cnmfe_set_noise on the fit patch (update_background_parallel.m:137)
    if (R && R != F) {
        if (R->p != F->p || R->d != F->d) return fail(CNMFE_ESTATE, "fit / residual low-resolution patches have different rings");
        CK(hipMemcpyAsync(R->W.p, F->W.p, (size_t)F->p * F->d * sizeof(float), hipMemcpyDeviceToDevice, ctx->st()));
        R->stat_valid = false;
        R->ysig_valid = false;
    }
    M->ysig_valid = false;
    return 0;
}

// (W A)(C - mean C) of the low-resolution sweep, seen from the full-resolution patch: up[(W A_low)] as ELL rows per patch pixel.
// wa_up(i, l) = sum over the row / column taps of pixel i of w_r w_c wa_low(j(r,c), l).  One thread per patch pixel, its <= UP_CAP slots in LDS.
constexpr int UP_CAP = 64, UP_NT = 64;    // slots per pixel (twice what ONE low-resolution ring may touch, WA_CAP: the 6 x 6 taps of a pixel share most of theirs) / threads per workgroup
__global__ void __launch_bounds__(UP_NT) k_wa_upsample(int64_t d, int nr, int nr_b, int roff, int coff, int d1s, int64_t d_low, const int *__restrict__ ir,
                                                     const float *__restrict__ wr, int Pr, const int *__restrict__ ic, const float *__restrict__ wc, int Pc,
                                                     const int *__restrict__ cnt_l, const int *__restrict__ k_l, const float *__restrict__ v_l,
                                                     int *__restrict__ cnt, int *__restrict__ kk, float *__restrict__ vv, int *__restrict__ overflow) {
    __shared__ int tk[UP_CAP][UP_NT];
    __shared__ float tv[UP_CAP][UP_NT];
    const int64_t m = (int64_t)blockIdx.x * UP_NT + threadIdx.x;
    if (m >= d) return;
    const int t = threadIdx.x;
    const int rb = (int)(m % nr) + roff, cb = (int)(m / nr) + coff;
    int n = 0;
    for (int u = 0; u < Pc; ++u) {
        const float wcu = wc[cb * Pc + u];
        if (wcu == 0.f) continue;
        const int jc = ic[cb * Pc + u];
        for (int q = 0; q < Pr; ++q) {
            const float w = wcu * wr[rb * Pr + q];
            if (w == 0.f) continue;
            const int64_t j = (int64_t)jc * d1s + ir[rb * Pr + q];
            const int ne = cnt_l[j];
            for (int e = 0; e < ne; ++e) {
                const int k = k_l[(int64_t)e * d_low + j];
                int s_ = 0;
                while (s_ < n && tk[s_][t] != k) ++s_;
                if (s_ == n) {
                    if (n == UP_CAP) { atomicOr(overflow, 4); continue; }     // (bit 2 of the context's error flag: reported at the next wait)
                    tk[s_][t] = k; tv[s_][t] = 0.f; ++n;
                }
                tv[s_][t] = fmaf(w, v_l[(int64_t)e * d_low + j], tv[s_][t]);
            }
        }
    }
    cnt[m] = n;
    for (int s_ = 0; s_ < n; ++s_) { kk[(int64_t)s_ * d + m] = tk[s_][t]; vv[(int64_t)s_ * d + m] = tv[s_][t]; }
}

// imresize's bicubic upsampling taps of M's block region from R's grid: host copies, their (monotone) ranges, their transposes, and the device arrays -- once per patch
int ssub_taps(cnmfe_ctx *ctx, Patch *M, const Patch *R) {
    const int d1s = R->d1, d2s = R->d2;
    if (M->ss_taps && M->ss_d1s == d1s && M->ss_d2s == d2s) return 0;
    M->ss_taps = false;
    Taps tr = make_taps(d1s, M->nr_b, (double)M->nr_b / d1s, false), tc = make_taps(d2s, M->nc_b, (double)M->nc_b / d2s, false);
    auto ranges = [](const Taps &t, std::vector<int> &lo, std::vector<int> &hi) {
        lo.assign((size_t)t.n_out, INT_MAX); hi.assign((size_t)t.n_out, -1);
        for (int o = 0; o < t.n_out; ++o)
            for (int i = 0; i < t.P; ++i) if (t.w[(size_t)o * t.P + i] != 0.f) { lo[o] = std::min(lo[o], t.idx[(size_t)o * t.P + i]); hi[o] = std::max(hi[o], t.idx[(size_t)o * t.P + i]); }
        for (int o = t.n_out - 2; o >= 0; --o) lo[o] = std::min(lo[o], lo[o + 1]);        // monotone: a run of rows [a, b] reaches [lo[a], hi[b]]
        for (int o = 1; o < t.n_out; ++o) hi[o] = std::max(hi[o], hi[o - 1]);
    };
    auto transpose = [](const Taps &t, std::vector<int> &ptr, std::vector<int> &idx, std::vector<float> &w) {
        ptr.assign((size_t)t.n_in + 1, 0);
        for (int o = 0; o < t.n_out; ++o) for (int i = 0; i < t.P; ++i) if (t.w[(size_t)o * t.P + i] != 0.f) ++ptr[t.idx[(size_t)o * t.P + i] + 1];
        for (int j = 0; j < t.n_in; ++j) ptr[j + 1] += ptr[j];
        idx.resize((size_t)ptr[t.n_in]); w.resize((size_t)ptr[t.n_in]);
        std::vector<int> cur(ptr.begin(), ptr.end() - 1);
        for (int o = 0; o < t.n_out; ++o)                    // ascending output index, tap order: a fixed order of every transposed sum
            for (int i = 0; i < t.P; ++i) { const float v = t.w[(size_t)o * t.P + i]; if (v != 0.f) { const int q = cur[t.idx[(size_t)o * t.P + i]]++; idx[q] = o; w[q] = v; } }
    };
    ranges(tr, M->ss_rlo, M->ss_rhi); ranges(tc, M->ss_clo, M->ss_chi);
    std::vector<float> trw, tcw;
    transpose(tr, M->ss_trp_h, M->ss_tri_h, trw); transpose(tc, M->ss_tcp_h, M->ss_tci_h, tcw);
    RET(to_dev(ctx, M->ss_ir, tr.idx.data(), tr.idx.size())); RET(to_dev(ctx, M->ss_wr, tr.w.data(), tr.w.size()));
    RET(to_dev(ctx, M->ss_ic, tc.idx.data(), tc.idx.size())); RET(to_dev(ctx, M->ss_wc, tc.w.data(), tc.w.size()));
    RET(to_dev(ctx, M->ss_trp, M->ss_trp_h.data(), M->ss_trp_h.size())); RET(to_dev(ctx, M->ss_tri, M->ss_tri_h.data(), M->ss_tri_h.size())); RET(to_dev(ctx, M->ss_trw, trw.data(), trw.size()));
    RET(to_dev(ctx, M->ss_tcp, M->ss_tcp_h.data(), M->ss_tcp_h.size())); RET(to_dev(ctx, M->ss_tci, M->ss_tci_h.data(), M->ss_tci_h.size())); RET(to_dev(ctx, M->ss_tcw, tcw.data(), tcw.size()));
    M->ss_Pr = tr.P; M->ss_Pc = tc.P; M->ss_d1s = d1s; M->ss_d2s = d2s;
    M->ss_tr_idx.swap(tr.idx); M->ss_tr_w.swap(tr.w); M->ss_tc_idx.swap(tc.idx); M->ss_tc_w.swap(tc.w);
    M->ss_taps = true;
    return 0;
}

// Ysig = Y' + (Ymean - b0) - up(W down(Y')): the upsample of the low-resolution sweep in ctx->ysig_low, combined with the video
static int ssub_upsample(cnmfe_ctx *ctx, Patch *M, Patch *R) {
    const int d1s = M->ss_d1s;
    const int Pr = M->ss_Pr, Pc = M->ss_Pc;
    DevBuf &dIr = M->ss_ir, &dWr = M->ss_wr, &dIc = M->ss_ic, &dWc = M->ss_wc, &dDlt = M->ss_dlt;
    const std::vector<int> &tri = M->ss_tr_idx, &tci = M->ss_tc_idx;
    const int64_t ntmp = (int64_t)d1s * M->nc_b;
    RET(M->ysig.ensure((size_t)M->d * M->Tc * sizeof(float4)));
    RET(dDlt.ensure(M->d * sizeof(float)));
    LAUNCH(ctx, "r1_dlt", k_dlt2, dim3((unsigned)((M->d + 255) / 256)), dim3(256), 0, M->ymean_f.as<float>(), M->b0.as<double>(), dDlt.as<float>(),
           M->d, M->nr, M->nr_b, M->roff, M->coff);
    // the fused kernel needs 6-tap (upsampling) tables and a low-resolution window per 64 x 16 tile that fits its LDS tiles
    bool fused = Pr == 6 && Pc == 6;
    if (fused) {
        for (int r0 = M->roff; r0 < M->roff + M->nr && fused; r0 += UF_TR) {
            int lo = 1 << 30, hi = -1;
            for (int q = r0 * 6; q < std::min(r0 + UF_TR, M->roff + M->nr) * 6; ++q) { lo = std::min(lo, tri[q]); hi = std::max(hi, tri[q]); }
            if (hi - lo + 1 > UF_NLR) fused = false;
        }
        for (int c0 = M->coff; c0 < M->coff + M->nc && fused; c0 += UF_TC) {
            int lo = 1 << 30, hi = -1;
            for (int q = c0 * 6; q < std::min(c0 + UF_TC, M->coff + M->nc) * 6; ++q) { lo = std::min(lo, tci[q]); hi = std::max(hi, tci[q]); }
            if (hi - lo + 1 > UF_NLC) fused = false;
        }
    }
    if (fused) {
        const int ntr = (M->nr + UF_TR - 1) / UF_TR, ntc = (M->nc + UF_TC - 1) / UF_TC;
        const int64_t ntile = (int64_t)ntr * ntc;
        int64_t nseg = std::max<int64_t>(1, std::min<int64_t>(M->Tc, (4096 + ntile - 1) / ntile));
        const int cseg = (int)((M->Tc + nseg - 1) / nseg);
        nseg = (M->Tc + cseg - 1) / cseg;
        LAUNCH(ctx, "ssub_up_fused", k_up_fused, dim3((unsigned)ntile, (unsigned)nseg), dim3(256), 0, R->Yc4.as<float4>(), ctx->ysig_low.as<float4>(), d1s, R->d_b,
               M->Yc4.as<float4>(), M->d_b, M->nr_b, M->nr, M->nc, M->roff, M->coff, M->d, dDlt.as<float>(), dIr.as<int>(), dWr.as<float>(),
               dIc.as<int>(), dWc.as<float>(), ntr, M->Tc, cseg, M->ysig.as<float4>());
        return 0;
    }
    RET(ctx->up_tmp.ensure((size_t)ntmp * M->Tc * sizeof(float4)));
    LAUNCH(ctx, "ssub_up_cols", k_up_cols, dim3((unsigned)((ntmp + 255) / 256), (unsigned)M->Tc), dim3(256), 0, R->Yc4.as<float4>(), ctx->ysig_low.as<float4>(),
           d1s, R->d_b, ctx->up_tmp.as<float4>(), M->nc_b, dIc.as<int>(), dWc.as<float>(), Pc);
    LAUNCH(ctx, "ssub_up_rows_combine", k_up_rows_combine, dim3((unsigned)((M->d + 255) / 256), (unsigned)M->Tc), dim3(256), 0, ctx->up_tmp.as<float4>(), d1s,
           M->nc_b, M->Yc4.as<float4>(), M->d_b, M->nr_b, M->nr, M->roff, M->coff, M->d, dDlt.as<float>(), dIr.as<int>(), dWr.as<float>(), Pr,
           M->ysig.as<float4>());
    return 0;
}

// a virtual residual of cnmfe_residual_ssub becomes a resident one: the low-resolution sweep (no footprint term: the term, if any, pends beside Ysig at full
// resolution either way) and its upsample run now
int ssub_realize(cnmfe_ctx *ctx, Patch *M) {
    Patch *R = get_patch(ctx, M->ss_res);
    if (!R) return fail(CNMFE_ESTATE, "the low-resolution residual patch %d of a recorded residual is gone", M->ss_res);
    RET(ssub_taps(ctx, M, R));
    RET(residual_run(ctx, R, M->ss_res, 0, nullptr, nullptr, nullptr, nullptr, CNMFE_BOUND, nullptr, CNMFE_HOST, &ctx->ysig_low, 0));
    RET(ssub_upsample(ctx, M, R));
    M->ysig_virtual = false;
    return 0;
}

int ssub_residual(cnmfe_ctx *ctx, Patch *M, int pid, Patch *R, int res_id, int ssub, int32_t K, const int64_t *cp, const int32_t *ri,
                  const float *va, const float *C, int c_order, float *Ysig_out, int out_memspace) {
    int d1s, d2s; low_dims(M, ssub, d1s, d2s);
    if (R->d1 != d1s || R->d2 != d2s || R->T != M->T) return fail(CNMFE_ESTATE, "patch %d is not the low-resolution residual patch of patch %d", res_id, pid);
    std::vector<int64_t> ocp; std::vector<int32_t> ori; std::vector<float> ova;
    const bool has_a = K > 0 && cp[K] > 0;
    HostTrace ht(ctx, "residual_ssub");
    if (has_a) a_low(M, ssub, false, K, cp, ri, va, ocp, ori, ova);
    ht.mark("down(A_prev)");
    RET(ssub_taps(ctx, M, R));
    M->ss_res = res_id; M->ss_ssub = ssub;
    // Ysig = [Y' + (Ymean - b0) - up(W down(Y'))] + up(W down(A_prev)) (C - mean C): while the video, W (of the residual patch) and b0 are
    // unchanged a further call only changes the second bracket, exactly as in cnmfe_residual -- its full-resolution ELL form (k_wa_upsample)
    // goes through the same pending-term / streaming-delta machinery (resid.hip), so the iteration does at most ONE low-resolution sweep + upsample.
    // Round 5: and usually none -- a request nobody asks the output of is only RECORDED (ysig_virtual, as cnmfe_residual does since round 4): the spatial and
    // the temporal update take their projections through the resampling maps (vproj.hip, vproj_*_ssub), every other consumer realises it (ssub_realize).
    const bool lazy = ctx->opt("r1_delta", 1) != 0;
    const bool reuse = M->ysig_valid && M->res_kind == 2 && (M->ysig.p || M->ysig_virtual) && lazy;
    // (ssub_virtual 1, the default: only where it pays -- profiles/r05/ssub_virtual_check.txt: the sweep-free form wins at 512 x 512 x 10000 (8.2 against 13.9 ms per
    //  iteration) and loses on patches of 256 x 256 x 3000 and below (5.8 against 4.6 ms: its list building, k_vp_build_b_ssub and two extra passes over the
    //  low-resolution video outweigh a sweep that small); 2: always)
    const int64_t sv = ctx->opt("ssub_virtual", 1);
    const bool virt_new = !reuse && !Ysig_out && lazy && (sv >= 2 || (sv == 1 && (double)M->d_b * (double)M->T >= 5e8)) && ctx->opt("r1_virtual", 1) != 0 && ctx->opt("r1_lazy", 1) != 0;
    const bool tables = reuse || virt_new;
    RET(residual_run(ctx, R, res_id, has_a ? K : 0, has_a ? ocp.data() : nullptr, ori.data(), ova.data(), C, c_order, nullptr, CNMFE_HOST, &ctx->ysig_low, tables ? 1 : 0));
    const int64_t ldc_t = ctx->last_ldc;
    ht.mark("low-resolution W*A tables");
    // the centred traces of this call move to the patch (pendCc / resCc) and the buffer the patch held before comes back as tmp[1]: a swap
    // both ways, no hipMalloc / hipFree per call (the guard hands the leftover buffer back on every exit path)
    struct GiveBack { DevBuf b; DevBuf &home; ~GiveBack() { if (b.p && !home.p) b.swap(home); } } gb{DevBuf(), ctx->tmp[1]};
    DevBuf &tCc = gb.b;
    if (has_a) tCc.swap(ctx->tmp[1]);
    // the footprint term of this call at full resolution: into the pending (recorded / reused residual) or the applied slot of the main patch.  A pixel whose
    // interpolation window meets more than UP_CAP footprints raises the context's error flag (reported by the next call that waits, like the low-resolution ring's own
    // limit in k_ring_wa): reading a flag back here cost a drain of the stream in each of the iteration's two residual calls
    {
        DevBuf &tCnt = tables ? M->pendCnt : M->resCnt, &tK = tables ? M->pendK : M->resK, &tV = tables ? M->pendV : M->resV;
        if (has_a) {
            int *dErr = nullptr;
            RET(ctx_errflag(ctx, &dErr));
            RET(tCnt.ensure((size_t)M->d * sizeof(int))); RET(tK.ensure((size_t)UP_CAP * M->d * sizeof(int))); RET(tV.ensure((size_t)UP_CAP * M->d * sizeof(float)));
            LAUNCH(ctx, "ssub_wa_upsample", k_wa_upsample, dim3((unsigned)((M->d + UP_NT - 1) / UP_NT)), dim3(UP_NT), 0, M->d, M->nr, M->nr_b, M->roff, M->coff, d1s, R->d,
                   M->ss_ir.as<int>(), M->ss_wr.as<float>(), M->ss_Pr, M->ss_ic.as<int>(), M->ss_wc.as<float>(), M->ss_Pc, ctx->tmp[8].as<int>(), ctx->tmp[9].as<int>(),
                   ctx->tmp[10].as<float>(), tCnt.as<int>(), tK.as<int>(), tV.as<float>(), dErr);
            (tables ? M->pendCc : M->resCc).swap(tCc);
        }
        if (tables) {
            if (virt_new) {
                M->res_ac = false; M->res_ldc = ldc_t; M->res_K = 0; M->res_kind = 2;
                M->ysig_valid = true; M->ysig_virtual = true;
            }
            M->pend = reuse ? true : has_a; M->pend_ac = has_a; M->pend_ldc = ldc_t; M->pend_K = K;
            if (ctx->opt("r1_lazy", 1) == 0 || Ysig_out) RET(residual_materialize(ctx, M));
            if (Ysig_out) RET(ysig_export(ctx, M, M->ysig, Ysig_out, out_memspace));
            return 0;
        }
        M->res_ac = has_a; M->res_ldc = ldc_t; M->res_K = K; M->pend = false;
        M->res_kind = 2; M->ysig_virtual = false;
    }
    RET(ssub_upsample(ctx, M, R));
    M->ysig_valid = true;
    if (Ysig_out) RET(ysig_export(ctx, M, M->ysig, Ysig_out, out_memspace));
    return 0;
}


// ---- reconstruct_background / compute_RSS with bg_ssub > 1 (Sources2D.m:1325-1334, :1479-1486) -------------------------------------
// Unlike the update methods (bicubic), these two resize with 'nearest' both ways:
//   Bf = up( W * down(Y_block - b0_block - A_prev*C_prev) ),  down = pixel selection, up = pixel replication,
// so on the fit patch (whose resident video IS down(Y - Ymean)) it is three plain kernels: R_low = Yc_F + (Ymean - b0_block)[sel] -
// down(A_prev)*C_prev, B_low = W*R_low, and a sweep of the patch pixels that looks its low pixel up.  B_low stays with the context until
// the next call; neither function is on the path of the iteration.
__global__ void __launch_bounds__(256) k_bgs_rlow(const float4 *__restrict__ ycF, int64_t dF, const float *__restrict__ ymeanF, const float *__restrict__ b0low,
                                                  const int *__restrict__ arow, const int *__restrict__ acol, const float *__restrict__ aval,
                                                  const float *__restrict__ C, int64_t ldc, float4 *__restrict__ R) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (q >= dF) return;
    float4 v = ycF[c * dF + q];
    const float k = ymeanF[q] - b0low[q];
    v.x += k; v.y += k; v.z += k; v.w += k;
    if (arow)
        for (int e = arow[q]; e < arow[q + 1]; ++e) {
            const float a = aval[e];
            const float4 t = *reinterpret_cast<const float4 *>(C + (int64_t)acol[e] * ldc + 4 * c);
            v.x = fmaf(-a, t.x, v.x); v.y = fmaf(-a, t.y, v.y); v.z = fmaf(-a, t.z, v.z); v.w = fmaf(-a, t.w, v.w);
        }
    R[c * dF + q] = v;
}
__global__ void __launch_bounds__(256) k_bgs_wr(const float4 *__restrict__ R, int64_t dF, int d1s, int d2s, int p, const int *__restrict__ dr, const int *__restrict__ dc,
                                                const float *__restrict__ W, float4 *__restrict__ B) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (m >= dF) return;
    const int r0 = (int)(m % d1s), c0 = (int)(m / d1s);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < p; ++i) {
        const float w = W[(int64_t)i * dF + m];
        const int rr = r0 + dr[i], cc = c0 + dc[i];
        if (w == 0.f || rr < 0 || rr >= d1s || cc < 0 || cc >= d2s) continue;
        const float4 v = R[c * dF + (int64_t)cc * d1s + rr];
        a0 += (double)w * v.x; a1 += (double)w * v.y; a2 += (double)w * v.z; a3 += (double)w * v.w;
    }
    B[c * dF + m] = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
}
__global__ void __launch_bounds__(256) k_bgs_out(const float4 *__restrict__ B, int64_t dF, int d1s, const int *__restrict__ upr, const int *__restrict__ upc,
                                                 int64_t d, int nr, int roff, int coff, const float *__restrict__ b0new, int64_t frame0, float *__restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= d) return;
    const int64_t u = (int64_t)upc[(int)(m / nr) + coff] * d1s + upr[(int)(m % nr) + roff];
    const int64_t t = frame0 + blockIdx.y;
    out[(int64_t)blockIdx.y * d + m] = reinterpret_cast<const float *>(B + (t >> 2) * dF + u)[t & 3] + b0new[m];
}
// sum((Y(patch) - A*C - (Bf + b0_new)).^2): one thread per patch pixel and segment of frames
__global__ void __launch_bounds__(256) k_bgs_rss(const float4 *__restrict__ B, int64_t dF, int d1s, const int *__restrict__ upr, const int *__restrict__ upc,
                                                 const float4 *__restrict__ ycM, int64_t d_b, int nr_b, const float *__restrict__ ymeanM, int64_t d, int nr, int roff, int coff,
                                                 const float *__restrict__ b0new, const int *__restrict__ arow, const int *__restrict__ acol, const float *__restrict__ aval,
                                                 const float *__restrict__ C, int64_t ldc, int64_t T, int64_t Tc, int64_t cseg, double *__restrict__ partial) {
    __shared__ double red[4];
    const int64_t m0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = m0 < d;
    const int64_t m = valid ? m0 : d - 1;
    const int rb = (int)(m % nr) + roff, cb = (int)(m / nr) + coff;
    const int64_t q = (int64_t)cb * nr_b + rb, u = (int64_t)upc[cb] * d1s + upr[rb];
    const float k = ymeanM[q] - b0new[m];
    const int e0 = arow ? arow[m] : 0, e1 = arow ? arow[m + 1] : 0;
    const int64_t c0 = (int64_t)blockIdx.y * cseg, c1 = c0 + cseg < Tc ? c0 + cseg : Tc;
    double acc = 0.0;
    for (int64_t c = c0; c < c1; ++c) {
        const float4 y = ycM[c * d_b + q], b = B[c * dF + u];
        float4 r = make_float4(y.x + k - b.x, y.y + k - b.y, y.z + k - b.z, y.w + k - b.w);
        for (int e = e0; e < e1; ++e) {
            const float a = aval[e];
            const float4 t = *reinterpret_cast<const float4 *>(C + (int64_t)acol[e] * ldc + 4 * c);
            r.x = fmaf(-a, t.x, r.x); r.y = fmaf(-a, t.y, r.y); r.z = fmaf(-a, t.z, r.z); r.w = fmaf(-a, t.w, r.w);
        }
        if (valid) {
            const int64_t t0 = 4 * c;
            acc += (double)r.x * r.x;
            if (t0 + 1 < T) acc += (double)r.y * r.y;
            if (t0 + 2 < T) acc += (double)r.z * r.z;
            if (t0 + 3 < T) acc += (double)r.w * r.w;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

static std::vector<int> nearest_sel(const Taps &t) {          // the one source index a 'nearest' tap row keeps
    std::vector<int> s((size_t)t.n_out, 0);
    for (int o = 0; o < t.n_out; ++o) for (int i = 0; i < t.P; ++i) if (t.w[(size_t)o * t.P + i] != 0.f) s[o] = t.idx[(size_t)o * t.P + i];
    return s;
}

int ssub_background(cnmfe_ctx *ctx, Patch *M, int pid, Patch *F, int ssub, int32_t K, const int64_t *cp, const int32_t *ri, const float *va,
                    const float *C, int c_order, const float *b0_block) {
    int d1s, d2s; low_dims(M, ssub, d1s, d2s);
    if (F->d1 != d1s || F->d2 != d2s || F->T != M->T) return fail(CNMFE_ESTATE, "the fit patch is not the low-resolution patch of patch %d", pid);
    ctx->bgs_patch = -1;
    const int64_t dF = F->d;
    DevBuf &dC = ctx->tmp[0], &dArow = ctx->tmp[3], &dAcol = ctx->tmp[4], &dAval = ctx->tmp[5], &dB0 = ctx->tmp[6];
    const bool has_a = K > 0 && cp[K] > 0;
    int64_t ldc = 4;
    if (has_a) {
        std::vector<int64_t> ocp; std::vector<int32_t> ori; std::vector<float> ova;
        a_low(M, ssub, true, K, cp, ri, va, ocp, ori, ova);
        RET(upload_traces(ctx, dC, C, K, M->T, c_order, &ldc));
        HostCSR csr; csc_to_csr(dF, K, ocp.data(), ori.data(), ova.data(), csr);
        RET(to_dev(ctx, dArow, csr.rowptr.data(), csr.rowptr.size())); RET(to_dev(ctx, dAcol, csr.col.data(), csr.col.size())); RET(to_dev(ctx, dAval, csr.val.data(), csr.val.size()));
        CK(hipStreamSynchronize(ctx->st()));
    }
    const std::vector<int> sr = nearest_sel(make_taps(M->nr_b, d1s, 1.0 / ssub, true)), sc = nearest_sel(make_taps(M->nc_b, d2s, 1.0 / ssub, true));
    std::vector<float> b0low((size_t)dF);
    for (int co = 0; co < d2s; ++co) for (int ro = 0; ro < d1s; ++ro) b0low[(size_t)co * d1s + ro] = b0_block[(size_t)sc[co] * M->nr_b + sr[ro]];
    RET(to_dev(ctx, dB0, b0low.data(), b0low.size()));
    // up = imresize(., [nr_block nc_block], 'nearest'): the low pixel every block row / column replicates (:1329, :1483)
    const std::vector<int> ur = nearest_sel(make_taps(d1s, M->nr_b, (double)M->nr_b / d1s, true)), uc = nearest_sel(make_taps(d2s, M->nc_b, (double)M->nc_b / d2s, true));
    RET(to_dev(ctx, ctx->bgs_upr, ur.data(), ur.size())); RET(to_dev(ctx, ctx->bgs_upc, uc.data(), uc.size()));
    RET(ctx->bgs_r.ensure((size_t)dF * F->Tc * sizeof(float4))); RET(ctx->bgs_b.ensure((size_t)dF * F->Tc * sizeof(float4)));
    dim3 grid((unsigned)((dF + 255) / 256), (unsigned)F->Tc);
    LAUNCH(ctx, "bgs_rlow", k_bgs_rlow, grid, dim3(256), 0, F->Yc4.as<float4>(), dF, F->ymean_f.as<float>(), dB0.as<float>(), has_a ? dArow.as<int>() : nullptr,
           dAcol.as<int>(), dAval.as<float>(), dC.as<float>(), ldc, ctx->bgs_r.as<float4>());
    LAUNCH(ctx, "bgs_wr", k_bgs_wr, grid, dim3(256), 0, ctx->bgs_r.as<float4>(), dF, d1s, d2s, F->p, F->ring_dr.as<int>(), F->ring_dc.as<int>(), F->W.as<float>(),
           ctx->bgs_b.as<float4>());
    CK(hipStreamSynchronize(ctx->st()));
    ctx->bgs_patch = pid; ctx->bgs_d1s = d1s; ctx->bgs_dF = dF;
    return 0;
}

int ssub_bg_out(cnmfe_ctx *ctx, Patch *M, const float *b0_new, int64_t frame0, int64_t nframes, float *out, int out_memspace) {
    DevBuf &dB0n = ctx->tmp[7];
    RET(to_dev(ctx, dB0n, b0_new, (size_t)M->d));
    float *dst = out;
    if (out_memspace != CNMFE_DEVICE) { RET(ctx->stage.ensure((size_t)M->d * nframes * sizeof(float))); dst = ctx->stage.as<float>(); }
    LAUNCH(ctx, "bgs_out", k_bgs_out, dim3((unsigned)((M->d + 255) / 256), (unsigned)nframes), dim3(256), 0, ctx->bgs_b.as<float4>(), ctx->bgs_dF, ctx->bgs_d1s,
           ctx->bgs_upr.as<int>(), ctx->bgs_upc.as<int>(), M->d, M->nr, M->roff, M->coff, dB0n.as<float>(), frame0, dst);
    if (out_memspace != CNMFE_DEVICE) CK(hipMemcpyAsync(out, dst, (size_t)M->d * nframes * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    return 0;
}

int ssub_rss(cnmfe_ctx *ctx, Patch *M, int32_t K, const int64_t *cp, const int32_t *ri, const float *va, const float *C, int c_order, const float *b0_new,
             double *rss_out) {
    DevBuf &dC = ctx->tmp[0], &dArow = ctx->tmp[3], &dAcol = ctx->tmp[4], &dAval = ctx->tmp[5], &dB0n = ctx->tmp[7], &dPart = ctx->tmp[13];
    const bool has_a = K > 0 && cp[K] > 0;
    int64_t ldc = 4;
    if (has_a) {
        RET(upload_traces(ctx, dC, C, K, M->T, c_order, &ldc));
        HostCSR csr; csc_to_csr(M->d, K, cp, ri, va, csr);
        RET(to_dev(ctx, dArow, csr.rowptr.data(), csr.rowptr.size())); RET(to_dev(ctx, dAcol, csr.col.data(), csr.col.size())); RET(to_dev(ctx, dAval, csr.val.data(), csr.val.size()));
        CK(hipStreamSynchronize(ctx->st()));
    }
    RET(to_dev(ctx, dB0n, b0_new, (size_t)M->d));
    const int64_t nblk = (M->d + 255) / 256;
    int64_t nseg = std::max<int64_t>(1, std::min<int64_t>(M->Tc, (8192 + nblk - 1) / nblk));
    const int64_t cseg = (M->Tc + nseg - 1) / nseg;
    nseg = (M->Tc + cseg - 1) / cseg;
    RET(dPart.ensure((size_t)nblk * nseg * sizeof(double)));
    LAUNCH(ctx, "bgs_rss", k_bgs_rss, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, ctx->bgs_b.as<float4>(), ctx->bgs_dF, ctx->bgs_d1s, ctx->bgs_upr.as<int>(),
           ctx->bgs_upc.as<int>(), M->Yc4.as<float4>(), M->d_b, M->nr_b, M->ymean_f.as<float>(), M->d, M->nr, M->roff, M->coff, dB0n.as<float>(),
           has_a ? dArow.as<int>() : nullptr, dAcol.as<int>(), dAval.as<float>(), dC.as<float>(), ldc, M->T, M->Tc, cseg, dPart.as<double>());
    std::vector<double> part((size_t)nblk * nseg);
    CK(hipMemcpyAsync(part.data(), dPart.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    double s_ = 0.0;
    for (double v : part) s_ += v;
    *rss_out = s_;
    return 0;
}

// loads this translation unit's code object now (HIP loads it at the first launch of one of its kernels -- milliseconds each that would otherwise fall into the first iteration): cnmfe_create
int tu_warm_ssub() { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)k_wa_upsample) == hipSuccess ? 0 : -1; }

}  // namespace cnmfe

using namespace cnmfe;

int cnmfe_patch_derive(cnmfe_ctx *ctx, int src_patch, int new_patch, int32_t ssub, int mode) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *S = get_patch(ctx, src_patch);
    if (!S) return fail(CNMFE_ESTATE, "patch %d not created", src_patch);
    if (ssub < 2 || ssub > 8) return fail(CNMFE_EINVAL, "bg_ssub %d out of range (2..8)", ssub);
    if (mode != CNMFE_DERIVE_NEAREST && mode != CNMFE_DERIVE_BICUBIC) return fail(CNMFE_EINVAL, "unknown derive mode %d", mode);
    if (new_patch == src_patch) return fail(CNMFE_EINVAL, "a patch cannot be derived onto itself");
    CK(hipSetDevice(ctx->device));
    return ssub_derive(ctx, S, new_patch, ssub, mode);
}

int cnmfe_fit_ring_model_ssub(cnmfe_ctx *ctx, int patch_id, int fit_patch, int res_patch, int32_t ssub, int32_t K, const int64_t *A_colptr,
                              const int32_t *A_rowidx, const float *A_val, const float *C, int c_order, double thresh_outlier,
                              int with_projection, int64_t info[4]) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *M = get_patch(ctx, patch_id), *F = get_patch(ctx, fit_patch), *R = get_patch(ctx, res_patch);
    if (!M || !F || !R) return fail(CNMFE_ESTATE, "patch %d / %d / %d not created", patch_id, fit_patch, res_patch);
    if (!F->ring_ready || !R->ring_ready || !M->ring_ready) return fail(CNMFE_ESTATE, "rings not initialised");
    if (K < 0) return fail(CNMFE_EINVAL, "K=%d", K);
    if (K > 0) { RET(check_csc_pub("A", K, M->d_b, A_colptr, A_rowidx)); if ((!A_val && A_colptr[K] > 0) || (!C && c_order != CNMFE_BOUND)) return fail(CNMFE_EINVAL, "null A_val / C"); }
    CK(hipSetDevice(ctx->device));
    return ssub_fit(ctx, M, F, R, ssub, K, A_colptr, A_rowidx, A_val, C, c_order, with_projection, info, thresh_outlier);
}

int cnmfe_residual_ssub(cnmfe_ctx *ctx, int patch_id, int res_patch, int32_t ssub, int32_t Ksel, const int64_t *A_colptr,
                        const int32_t *A_rowidx, const float *A_val, const float *C, int c_order, float *Ysig_out, int out_memspace) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *M = get_patch(ctx, patch_id), *R = get_patch(ctx, res_patch);
    if (!M || !R) return fail(CNMFE_ESTATE, "patch %d / %d not created", patch_id, res_patch);
    if (!M->ring_ready || !R->ring_ready) return fail(CNMFE_ESTATE, "rings not initialised");
    if (Ksel < 0) return fail(CNMFE_EINVAL, "Ksel=%d", Ksel);
    if (Ksel > 0) { RET(check_csc_pub("A_prev", Ksel, M->d_b, A_colptr, A_rowidx)); if ((!A_val && A_colptr[Ksel] > 0) || (!C && c_order != CNMFE_BOUND)) return fail(CNMFE_EINVAL, "null A_val / C"); }
    CK(hipSetDevice(ctx->device));
    RET(ensure_ymean(ctx, M));
    return ssub_residual(ctx, M, patch_id, R, res_patch, ssub, Ksel, A_colptr, A_rowidx, A_val, C, c_order, Ysig_out, out_memspace);
}

int cnmfe_background_ssub(cnmfe_ctx *ctx, int patch_id, int fit_patch, int32_t ssub, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                          const float *A_val, const float *C, int c_order, const float *b0_block) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *M = get_patch(ctx, patch_id), *F = get_patch(ctx, fit_patch);
    if (!M || !F) return fail(CNMFE_ESTATE, "patch %d / %d not created", patch_id, fit_patch);
    if (!F->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", fit_patch);
    if (K < 0 || !b0_block) return fail(CNMFE_EINVAL, "K=%d / null b0_block", K);
    if (K > 0) { RET(check_csc_pub("A_prev", K, M->d_b, A_colptr, A_rowidx)); if ((!A_val && A_colptr[K] > 0) || (!C && c_order != CNMFE_BOUND)) return fail(CNMFE_EINVAL, "null A_val / C"); }
    CK(hipSetDevice(ctx->device));
    RET(ensure_ymean(ctx, M));
    return ssub_background(ctx, M, patch_id, F, ssub, K, A_colptr, A_rowidx, A_val, C, c_order, b0_block);
}

int cnmfe_reconstruct_background_ssub(cnmfe_ctx *ctx, int patch_id, const float *b0_new, int64_t frame0, int64_t nframes, float *Ybg_out, int out_memspace) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *M = get_patch(ctx, patch_id);
    if (!M) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (ctx->bgs_patch != patch_id) return fail(CNMFE_ESTATE, "cnmfe_background_ssub has not been run for patch %d", patch_id);
    if (!b0_new || !Ybg_out) return fail(CNMFE_EINVAL, "null b0_new / Ybg_out");
    if (frame0 < 0 || nframes <= 0 || frame0 + nframes > M->T || nframes > 65535) return fail(CNMFE_EINVAL, "frames [%lld, %lld) outside [0, %lld) or more than 65535 at once", (long long)frame0, (long long)(frame0 + nframes), (long long)M->T);
    CK(hipSetDevice(ctx->device));
    return ssub_bg_out(ctx, M, b0_new, frame0, nframes, Ybg_out, out_memspace);
}

int cnmfe_compute_rss_ssub(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val, const float *C,
                           int c_order, const float *b0_new, double *rss_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *M = get_patch(ctx, patch_id);
    if (!M) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (ctx->bgs_patch != patch_id) return fail(CNMFE_ESTATE, "cnmfe_background_ssub has not been run for patch %d", patch_id);
    if (K < 0 || !b0_new || !rss_out) return fail(CNMFE_EINVAL, "bad K / null b0_new / rss_out");
    if (K > 0) { RET(check_csc_pub("A", K, M->d, A_colptr, A_rowidx)); if ((!A_val && A_colptr[K] > 0) || (!C && c_order != CNMFE_BOUND)) return fail(CNMFE_EINVAL, "null A_val / C"); }
    CK(hipSetDevice(ctx->device));
    return ssub_rss(ctx, M, K, A_colptr, A_rowidx, A_val, C, c_order, b0_new, rss_out);
}
