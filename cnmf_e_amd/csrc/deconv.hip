// T4/T6: AR(1) FOOPSI deconvolution (OASIS) of calcium traces on the GPU, one workgroup per trace.
//   deconvolveCa.m:61-123,199-206 (type 'ar1', method 'foopsi')  ->  foopsi_oasisAR1.m:36-180  ->  oasisAR1.m:30-109
//   GetSn.m:19-46 (pwelch defaults)      estimate_time_constant.m:21-66 (p = 1)
// Callers: the deconvolution branch of HALS_temporal (utilities/HALS_temporal.m:70-104, fused here with the
// HALS row update :62) and @Sources2D/deconvTemporal.m:29-105.
//
// All scalar logic is fp64 like the reference; traces are fp32 rows.  Phases of one workgroup (256 threads):
//   raw trace -> LDS | bitonic sort (median, 15 % quantile) | Welch PSD via radix-2 FFT in LDS (sn) |
//   autocovariance (g) | OASIS pool stack, sequential on lane 0 | Brent fminbnd over g, each rss(g) evaluated in
//   parallel over <=64-sample tasks of the pools | solution c, s written in parallel.
// MathWorks semantics restated (parity unpinned, see oracle/oasis_oracle.py): pwelch, fminbnd (TolX 1e-4), quantile.
#include "common.hpp"
#include <math.h>

namespace cnmfe {

// the longest trace the Welch estimator takes: pwelch's segment length floor(T / 4.5) <= 32768 = the largest transform built (two 16384-point halves in LDS)
constexpr int WELCH_TMAX = 147456;
struct DeconvCfg {
    int T, P2, nfft, L, nov, nseg;     // trace length, pow2 >= T, Welch geometry
    int split;                         // round 6: nfft = 32768 (73728 < T <= 147456) as two 16384-point transforms (even / odd samples) + one butterfly, LDS = re | im of one of them
    int ylong;                         // 1: the trace lives in global memory (k_deconv<true>); 2: and so do the Welch tables (T > 36868: nfft = 16384)
    int maxIter;                       // foopsi iterations (20 inside HALS_temporal, 10 in deconvTemporal)
    int optimize_b, optimize_g;
    double smin_opt, lam, gmax;
    int hals;                          // 1: fuse the HALS row update and the median baseline (HALS_temporal.m:62,78)
    int last;                          // HALS: last sweep -> also write C_raw, S (:100-103)
    int trace;                         // option deconv_trace = k + 1: thread 0 of trace k prints the foopsi / fminbnd sequence (scripts/deconv_trace.py compares it with the oracle's)
};

// the per-job part of DeconvIO when one launch serves several temporal jobs (factor.hip, temporal_sweep_jobs): workgroup -> jlist[slot] = (job, neuron)
struct DeconvJobDev { float *C, *Craw, *S; int64_t ldc; const float *U; const int *nptr, *nidx; const float *nval, *aa; float *pars, *sn_out, *b_out; };
struct DeconvIO {
    const DeconvJobDev *jobs = nullptr; const int2 *jlist = nullptr;
    const int *list;                   // neuron ids handled by this launch
    float *C; float *Craw; float *S; int64_t ldc;
    const float *U; const int *nptr; const int *nidx; const float *nval; const float *aa;   // HALS inputs
    float *pars; float *sn_out; float *b_out;
    double *pv, *pw; int *pt, *pl;     // pool scratch, T entries per trace slot
    int *tk_pool, *tk_off, *tk_len;    // task scratch, 2*T per trace slot
    double *tk_val;
    double *pnum;                      // per-pool numerators, T per trace slot
    float *ybuf, *obuf;                // long traces only: the raw trace and the output staging, Tal floats per trace slot each
    float *tbuf = nullptr;             // ylong == 2 (nfft >= 16384): the Welch twiddle / window tables (+ the even samples' transform when the transform is split), 3 nfft floats per trace slot
};

// (red: one double per wave of the workgroup -- 4 for the 256-thread kernels, 8 for k_deconv's 512)
__device__ __forceinline__ double block_sum(double v, double *red) {
    const int tid = threadIdx.x, nw = (int)(blockDim.x >> 6);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = (red[0] + red[1]) + (red[2] + red[3]);
    for (int w = 4; w < nw; w += 4) r += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
    __syncthreads();
    return r;
}

// ---- order statistics without a sort -----------------------------------------------------------------------
// The trace needs two adjacent order statistics twice (median of HALS_temporal.m:78, interpolated 15 % quantile of
// foopsi_oasisAR1.m:93).  An MSD radix select on the order-preserving integer key of the floats finds the k-th smallest in four
// 256-bin histogram passes over the LDS copy of the trace (a full bitonic sort of 16384 slots cost 0.67 ms per trace, a fifth
// of the whole kernel); the (k+1)-th is the same value when the k-th is repeated often enough, else the smallest key above it.
__device__ __forceinline__ unsigned fkey(float x) { const unsigned u = __float_as_uint(x); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ void select_pair(const float *y, int T, int k, int *hist /* 260 ints of LDS */, float &vk, float &vk1) {
    const int tid = threadIdx.x, NTH = (int)blockDim.x;
    unsigned prefix = 0, known = 0;
    int kk = k, cnt_eq = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int t = tid; t < T; t += NTH) { const unsigned key = fkey(y[t]); if ((key & known) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1); }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, b = 0;
            for (; b < 255; ++b) { const int h = hist[b]; if (acc + h > kk) break; acc += h; }
            hist[256] = b; hist[257] = kk - acc; hist[258] = hist[b];
        }
        __syncthreads();
        prefix |= (unsigned)hist[256] << shift; known |= 255u << shift; kk = hist[257]; cnt_eq = hist[258];
        __syncthreads();
    }
    vk = fkey_inv(prefix);
    if (kk + 1 < cnt_eq) { vk1 = vk; return; }
    unsigned mn = 0xffffffffu;
    for (int t = tid; t < T; t += NTH) { const unsigned key = fkey(y[t]); if (key > prefix && key < mn) mn = key; }
    for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_xor(mn, o); mn = other < mn ? other : mn; }
    if ((tid & 63) == 0) hist[tid >> 6] = (int)mn;
    __syncthreads();
    unsigned r = (unsigned)hist[0];
    for (int w = 1; w < (NTH >> 6); ++w) r = (unsigned)hist[w] < r ? (unsigned)hist[w] : r;
    __syncthreads();
    vk1 = r == 0xffffffffu ? vk : fkey_inv(r);
}

// in-place radix-2 FFT of n complex points in LDS (re, im), 256 threads; the input is already in bit-reversed order and
// tw[k] = exp(-2 pi i k / n), k < n/4, is a table (a sincospif per butterfly was most of the transform's time); the second quarter
// of the circle is -i times the first
// In-place radix-2 decimation-in-time FFT on bit-reversed input, TWO stages per pass: a thread takes the four elements i0 + {0, h, 2h, 3h} that stages s and s + 1
// combine among themselves, runs both butterflies in registers and writes them back -- the same operations with the same twiddles as two single-stage passes (bit-
// identical results), at half the barriers and 0.6x the LDS traffic (8 reads + 8 writes + 3 twiddles per 4 points instead of 2 x (8 + 8 + 4)).  The per-pixel GetSn
// is four 4096-point transforms per pixel, 262144 pixels: 54 ms at the headline size with one stage per pass.
__device__ __forceinline__ float2 fft_tw(const float2 *tw, int tk, int n) {
    float2 w = tw[tk & (n / 4 - 1)];
    if (tk >= n / 4) w = make_float2(w.y, -w.x);
    return w;
}
__device__ void fft_lds(float *re, float *im, const float2 *tw, int n, int logn) {
    const int tid = threadIdx.x, NTH = (int)blockDim.x;
    int s = 1;
    if (logn & 1) {                                          // an odd number of stages: the first one (twiddle 1) on its own
        for (int b = tid; b < n / 2; b += NTH) {
            const int i0 = 2 * b, i1 = i0 + 1;
            const float ur = re[i0], ui = im[i0], xr = re[i1], xi = im[i1];
            re[i0] = ur + xr; im[i0] = ui + xi; re[i1] = ur - xr; im[i1] = ui - xi;
        }
        __syncthreads();
        s = 2;
    }
    for (; s < logn; s += 2) {
        const int h = 1 << (s - 1), t1 = n >> s, t2 = n >> (s + 1);
        for (int q = tid; q < n / 4; q += NTH) {
            const int pos = q & (h - 1);
            const int i0 = ((q >> (s - 1)) << (s + 1)) + pos, i1 = i0 + h, i2 = i0 + 2 * h, i3 = i0 + 3 * h;
            const float2 w1 = fft_tw(tw, pos * t1, n), wa = fft_tw(tw, pos * t2, n), wb = fft_tw(tw, (pos + h) * t2, n);
            // stage s: (i0, i1) and (i2, i3), both with w1
            float xr = re[i1] * w1.x - im[i1] * w1.y, xi = re[i1] * w1.y + im[i1] * w1.x;
            float ur = re[i0], ui = im[i0];
            const float a0r = ur + xr, a0i = ui + xi, a1r = ur - xr, a1i = ui - xi;
            xr = re[i3] * w1.x - im[i3] * w1.y; xi = re[i3] * w1.y + im[i3] * w1.x;
            ur = re[i2]; ui = im[i2];
            const float a2r = ur + xr, a2i = ui + xi, a3r = ur - xr, a3i = ui - xi;
            // stage s + 1: (i0, i2) with wa, (i1, i3) with wb
            xr = a2r * wa.x - a2i * wa.y; xi = a2r * wa.y + a2i * wa.x;
            re[i0] = a0r + xr; im[i0] = a0i + xi; re[i2] = a0r - xr; im[i2] = a0i - xi;
            xr = a3r * wb.x - a3i * wb.y; xi = a3r * wb.y + a3i * wb.x;
            re[i1] = a1r + xr; im[i1] = a1i + xi; re[i3] = a1r - xr; im[i3] = a1i - xi;
        }
        __syncthreads();
    }
}

// GetSn(y, [0.25 0.5], 'logmexp'): Welch PSD (pwelch defaults: Hamming window of floor(T/4.5) samples, 50 % overlap, nfft =
// max(256, nextpow2(L))), noise = sqrt(exp(mean(log(psd/2)))) over the upper half band.  scr = re | im | twiddles (nfft/2 floats) | window (nfft floats, only
// with `wintab`; k_sn_pixels does without to keep two workgroups per CU).
// Two real segments ride one complex transform (z = a + i b): the Welch sum only needs |A_k|^2 + |B_k|^2 = (|Z_k|^2 + |Z_{n-k}|^2) / 2,
// and the same expression is |A_k|^2 for an unpaired last segment (b = 0).
struct Ysig4Acc {                                    // frame t of one pixel of a [T/4][npix] float4 video (+ a constant): no LDS copy of the trace
    const float4 *base; int64_t npix; float add;
    __device__ __forceinline__ float operator[](int t) const { return reinterpret_cast<const float *>(base + (int64_t)(t >> 2) * npix)[t & 3] + add; }
};
// `y` is anything indexable by the frame (a pointer into LDS or global memory, or an accessor of the 4-frame-interleaved video: Ysig4Acc below).
// `tabs`: where the twiddle (nfft / 2 floats) and window (nfft floats) tables live -- nullptr: behind re | im in scr; a global buffer for recordings whose
// transform (nfft = 16384: T > 36868) leaves no room for them in the 160 KB of LDS.
// MAXB: band bins per thread -- nfft / 4 + 1 bins over the workgroup's threads: 17 covers nfft <= 16384 with 256 threads and nfft <= 32768 with 512 (k_deconv); the
// 256-thread kernel of long recordings asks for 33
// SPLIT: the instantiation that also knows the split transform (nfft = 32768) -- its own kernels: with the branch compiled into the common ones hipcc
// allocated them 250-500 more spilled registers (k_deconv 71 -> 317, k_sn_pixels from 151 registers to 256)
template <class YT, int MAXB = 17, bool SPLIT = false>
__device__ double get_sn(const YT &y, const DeconvCfg &c, float *scr, double *red, bool wintab, float *tabs = nullptr) {
    const int tid = threadIdx.x, NTH = (int)blockDim.x;
    const int nfft = c.nfft, L = c.L, step = c.L - c.nov;
    const int nh = SPLIT ? nfft / 2 : nfft;                    // the transform that runs in LDS
    float *tb = tabs ? tabs : scr + 2 * nfft;
    float *re = scr, *im = scr + nh, *win = tb + nfft / 2;
    float *ebuf = tb + nfft / 2 + nfft;                        // (split: the even samples' transform, 2 nh floats, in the tables' global slot)
    float2 *tw = reinterpret_cast<float2 *>(tb);
    int logn = 0; while ((1 << logn) < nh) ++logn;
    const int k0 = (nfft + 3) / 4, k1 = nfft / 2;              // bins with 0.25 <= k/nfft <= 0.5
    const int nb = k1 - k0 + 1;
    float acc[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) acc[i] = 0.f;
    double w2 = 0;
    for (int i = tid; i < L; i += NTH) { const double w = 0.54 - 0.46 * cospi(2.0 * i / (double)(L - 1)); w2 += w * w; if (wintab) win[i] = (float)w; }
    for (int k = tid; k < nh / 4; k += NTH) { float sn_, cs_; sincospif(-(float)k / (float)(nh / 2), &sn_, &cs_); tw[k] = make_float2(cs_, sn_); }
    if (tabs) __threadfence_block();                 // (tables in global memory: written and read by this workgroup alone; block_sum's barriers order them)
    w2 = block_sum(w2, red);
    if constexpr (SPLIT) {
        // X_k = E_k + w^k O_k (w = exp(-2 pi i / nfft)) with E, O the nh-point transforms of the even / odd samples of the windowed, zero-padded pair of segments:
        // E goes to the global slot, O stays in LDS, and the band's bins k, nfft - k (k = nfft/4 .. nfft/2: indices k mod nh and nh - k) are combined from there
        for (int sg = 0; sg < c.nseg; sg += 2) {
            const bool two = sg + 1 < c.nseg;
            for (int par = 0; par < 2; ++par) {
                for (int j = tid; j < nh; j += NTH) {
                    const int i = 2 * j + par;
                    float va = 0.f, vb = 0.f;
                    if (i < L) { const float w = wintab ? win[i] : (float)(0.54 - 0.46 * cospi(2.0 * i / (double)(L - 1))); va = y[sg * step + i] * w; if (two) vb = y[(sg + 1) * step + i] * w; }
                    const int jr = (int)(__brev((unsigned)j) >> (32 - logn));
                    re[jr] = va; im[jr] = vb;
                }
                __syncthreads();
                fft_lds(re, im, tw, nh, logn);
                if (par == 0) {
                    for (int j = tid; j < nh; j += NTH) { ebuf[j] = re[j]; ebuf[nh + j] = im[j]; }
                    __threadfence_block();
                    __syncthreads();
                }
            }
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int k = k0 + tid + i * NTH;
                if (k <= k1) {
                    float sn_, cs_;
                    sincospif(-2.0f * (float)k / (float)nfft, &sn_, &cs_);        // w^k; w^(nfft - k) is its conjugate
                    const int i1 = k & (nh - 1), j2 = (nfft - k) & (nh - 1);                  // k mod nh, (nfft - k) mod nh
                    const float o1r = re[i1], o1i = im[i1], o2r = re[j2], o2i = im[j2];
                    const float z1r = ebuf[i1] + (cs_ * o1r - sn_ * o1i), z1i = ebuf[nh + i1] + (cs_ * o1i + sn_ * o1r);
                    const float z2r = ebuf[j2] + (cs_ * o2r + sn_ * o2i), z2i = ebuf[nh + j2] + (cs_ * o2i - sn_ * o2r);
                    acc[i] += 0.5f * ((z1r * z1r + z1i * z1i) + (z2r * z2r + z2i * z2i));
                }
            }
            __syncthreads();
        }
    } else
    for (int sg = 0; sg < c.nseg; sg += 2) {
        const bool two = sg + 1 < c.nseg;
        for (int i = tid; i < nfft; i += NTH) {
            float va = 0.f, vb = 0.f;
            if (i < L) { const float w = wintab ? win[i] : (float)(0.54 - 0.46 * cospi(2.0 * i / (double)(L - 1))); va = y[sg * step + i] * w; if (two) vb = y[(sg + 1) * step + i] * w; }
            const int j = (int)(__brev((unsigned)i) >> (32 - logn));
            re[j] = va; im[j] = vb;
        }
        __syncthreads();
        fft_lds(re, im, tw, nfft, logn);
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            const int k = k0 + tid + i * NTH;
            if (k <= k1) acc[i] += 0.5f * ((re[k] * re[k] + im[k] * im[k]) + (re[nfft - k] * re[nfft - k] + im[nfft - k] * im[nfft - k]));
        }
        __syncthreads();
    }
    double ls = 0;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int k = k0 + tid + i * NTH;
        if (k <= k1) {
            double psd = (double)acc[i] / ((double)c.nseg * w2);
            if (k != nfft / 2) psd *= 2.0;                     // one-sided: Nyquist bin is not doubled
            ls += log(psd / 2.0);
        }
    }
    ls = block_sum(ls, red);
    return sqrt(exp(ls / (double)nb));
}

// estimate_time_constant(y, 1, sn): returns g, or a negative flag (-2) when |g| > 1
__device__ __forceinline__ double est_g(const float *y, double shift, int T, double sn, double *red) {
    const int tid = threadIdx.x, NTH = (int)blockDim.x;
    double m = 0;
    for (int t = tid; t < T; t += NTH) m += (double)y[t] - shift;
    m = block_sum(m, red) / T;
    double xc[7];
    for (int k = 0; k <= 6; ++k) {
        double s = 0;
        for (int t = tid; t + k < T; t += NTH) s += ((double)y[t + k] - shift - m) * ((double)y[t] - shift - m);
        xc[k] = block_sum(s, red) / T;
    }
    double num = 0, den = 0;
    for (int i = 0; i < 6; ++i) { const double a = xc[i] - (i == 0 ? sn * sn : 0.0); num += a * xc[i + 1]; den += a * a; }
    double g = num / den;
    if (fabs(g) > 1.0) return -2.0;
    if (g < 0) g = 0.15;
    return g;
}

struct Pools { double *v, *w; int *t, *l; int n; };

// oasisAR1 on lane 0: left-to-right pool stack.  Input pools: singletons of (y - b) when `warm` is 0, else the
// current pool list (v, w recomputed by the caller).  gl = g^l is carried multiplicatively.
template <bool warm>
__device__ void oasis_seq_t(const float *y, double bsub, int T, double g, double lam, double smin, Pools &P, float *lds_scr, int nc) {
    const int nin = warm ? P.n : T;
    // the lowest `nc` stack entries are mirrored in LDS (cold pass): the pool that comes to the top after a back-track merge is then
    // re-read from LDS, not from global memory -- in a decaying transient every other sample takes that path (1 us per global round trip)
    double *lv = reinterpret_cast<double *>(lds_scr), *lw = lv + nc;
    int *lt = reinterpret_cast<int *>(lw + nc), *ll = lt + nc;
    // the stack is built in place: output index <= input index, so reading input i never sees an overwritten slot.
    // This loop is one lane's dependent fp64 chain, so its length is what matters:
    //  * the pool under the current one is mirrored in registers (pv, pw, pt, pl, pgl): the back-track test that follows every merge costs
    //    no memory round trip (the stack lives in global memory; one dependent L2 read per sample made this loop 1 us per sample) and no
    //    pow(); memory is read only after a back-track merge succeeded;
    //  * the pool values v/w are compared cross-multiplied (weights are sums of g^2j > 0), which removes three fp64 divisions (~35
    //    dependent instructions each) per sample: v_n/w_n >= (v/w) g^l + smin  <=>  v_n w >= (v g^l + smin w) w_n, and
    //    v/w < max(v_p/w_p g^lp, 0) + smin  <=>  v w_p < (max(v_p g^lp, 0) + smin w_p) w.  (Same decisions as oasisAR1.m:64-65,83-85 up to
    //    the last-bit rounding of either form.)
    double cv, cw, cgl; int ct, cl;
    if (warm) { cv = P.v[0]; cw = P.w[0]; ct = P.t[0]; cl = P.l[0]; cgl = pow(g, (double)cl); }
    else { cv = ((double)y[0] - bsub) - lam * (1 - g); if (T == 1) cv = ((double)y[0] - bsub) - lam; cw = 1.0; ct = 1; cl = 1; cgl = g; }
    int top = 0;                                  // number of pools already on the stack (below cur)
    double pv = 0, pw = 1, pgl = 1; int pt = 0, pl = 0;            // mirror of stack entry top-1 (valid when top > 0)
    const double lam_in = lam * (1 - g);
    float ynext = (!warm && nin > 1) ? y[1] : 0.f;               // fresh samples are fetched one iteration ahead of their use
    for (int i = 1; i < nin; ++i) {
        double nv, nw; int nt, nl; double ngl;
        if (warm) { nv = P.v[i]; nw = P.w[i]; nt = P.t[i]; nl = P.l[i]; ngl = pow(g, (double)nl); }
        else {
            const float yi = ynext;
            ynext = y[i + 1 < nin ? i + 1 : i];
            nv = ((double)yi - bsub) - (i == T - 1 ? lam : lam_in); nw = 1.0; nt = i + 1; nl = 1; ngl = g;
        }
        if (nv * cw >= fma(smin, cw, cv * cgl) * nw) {                 // oasisAR1.m:64-65: no violation, advance
            P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl;
            if (!warm && top < nc) { lv[top] = cv; lw[top] = cw; lt[top] = ct; ll[top] = cl; }
            ++top;
            pv = cv; pw = cw; pt = ct; pl = cl; pgl = cgl;
            cv = nv; cw = nw; ct = nt; cl = nl; cgl = ngl;
            continue;
        }
        cv = fma(nv, cgl, cv); cw = fma(nw * cgl, cgl, cw); cl += nl; cgl *= ngl;       // :74-76 merge
        while (top > 0) {                          // :83-95 backtrack
            const double lim = pv * pgl;
            if (!(cv * pw < fma(smin, pw, lim > 0.0 ? lim : 0.0) * cw)) break;
            cv = fma(cv, pgl, pv); cw = fma(cw * pgl, pgl, pw); ct = pt; cl = pl + cl; cgl = pgl * cgl;
            --top;
            if (top > 0) {
                if (!warm && top - 1 < nc) { pv = lv[top - 1]; pw = lw[top - 1]; pt = lt[top - 1]; pl = ll[top - 1]; }
                else { pv = P.v[top - 1]; pw = P.w[top - 1]; pt = P.t[top - 1]; pl = P.l[top - 1]; }
                pgl = pow(g, (double)pl);
            }
        }
    }
    P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl;
    P.n = top + 1;
}

// The cold pass with lambda = 0 (the only lambda deconv_setup admits), written for the fewest instructions per sample: one lane of one
// wave issues an instruction every ~8 clocks whatever its kind, so 60 instructions per sample (the generic loop above under a divergent
// exec mask) are 0.24 us per sample = 2.4 ms per 10^4-frame trace.  Here
//  * branch conditions go through a ballot, which tells the compiler they are wave-uniform: scalar branches, no exec-mask bookkeeping;
//  * both thresholds are kept in the form they are compared in: thr = smin w + v g^l of the current pool (recomputed once per sample),
//    plim = smin w_p + max(v_p g^lp, 0) of the pool under it (recomputed only when that pool changes; -inf on an empty stack, so no
//    separate `top > 0` test);
//  * the stack is mirrored in LDS together with g^l: a back-track merge costs one LDS round trip and no pow().
// The comparisons are the ones of oasis_seq_t term by term (n_w = 1, lambda (1 - g) = 0), so the pools are identical.
__device__ __forceinline__ bool wave_uniform(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }   // (lane 0 is the only active lane)
__device__ void oasis_cold(const float *y, double bsub, int T, double g, double smin, Pools &P, float *lds_scr, int nc) {
    double *lv = reinterpret_cast<double *>(lds_scr), *lw = lv + nc, *lg = lw + nc;          // 32 B per mirrored pool
    int *lt = reinterpret_cast<int *>(lg + nc), *ll = lt + nc;
    double cv = (double)y[0] - bsub, cw = 1.0, cgl = g;
    int ct = 1, cl = 1, top = 0;
    double thr = fma(smin, cw, cv * cgl);
    double pv = 0, pw = 1, pgl = 1, plim = -INFINITY; int pt = 0, pl = 0;
    float ynext = T > 1 ? y[1] : 0.f;
    for (int i = 1; i < T; ++i) {
        const double nv = (double)ynext - bsub;
        ynext = y[i + 1 < T ? i + 1 : i];
        if (wave_uniform(nv * cw >= thr)) {                              // oasisAR1.m:64-65: no violation, push the current pool
            P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl;
            if (top < nc) { lv[top] = cv; lw[top] = cw; lg[top] = cgl; lt[top] = ct; ll[top] = cl; }
            ++top;
            pv = cv; pw = cw; pt = ct; pl = cl; pgl = cgl;
            { const double lim = pv * pgl; plim = fma(smin, pw, lim > 0.0 ? lim : 0.0); }
            cv = nv; cw = 1.0; ct = i + 1; cl = 1; cgl = g;
        } else {
            cv = fma(nv, cgl, cv); cw = fma(cgl, cgl, cw); cl += 1; cgl *= g;              // :74-76 merge
            while (wave_uniform(cv * pw < plim * cw)) {                  // :83-95 back-track
                cv = fma(cv, pgl, pv); cw = fma(cw * pgl, pgl, pw); ct = pt; cl = pl + cl; cgl = pgl * cgl;
                --top;
                if (top > 0) {
                    if (top - 1 < nc) { pv = lv[top - 1]; pw = lw[top - 1]; pgl = lg[top - 1]; pt = lt[top - 1]; pl = ll[top - 1]; }
                    else { pv = P.v[top - 1]; pw = P.w[top - 1]; pt = P.t[top - 1]; pl = P.l[top - 1]; pgl = pow(g, (double)pl); }
                    const double lim = pv * pgl; plim = fma(smin, pw, lim > 0.0 ? lim : 0.0);
                } else plim = -INFINITY;
            }
        }
        thr = fma(smin, cw, cv * cgl);
    }
    P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl;
    P.n = top + 1;
}

// The warm-started pass (foopsi_oasisAR1.m:155-161 hands update_g's pools back to oasisAR1) with its input pools staged in LDS by the
// whole workgroup -- v, w, t, l and g^l per pool, 32 B each -- so that lane 0 neither waits for a global load nor evaluates a pow() per pool
// (90 us per pass for ~10^2 pools).  Same tests as oasis_seq_t<true>; the stack is written to P as there.
__device__ __forceinline__ void oasis_warm(int nin, double g, double smin, Pools &P, const double *sv, const double *sw, const double *sg, const int *st, const int *sl) {
    double cv = sv[0], cw = sw[0], cgl = sg[0];
    int ct = st[0], cl = sl[0], top = 0;
    double pv = 0, pw = 1, pgl = 1, plim = -INFINITY; int pt = 0, pl = 0;
    for (int i = 1; i < nin; ++i) {
        const double nv = sv[i], nw = sw[i], ngl = sg[i];
        const int nt = st[i], nl = sl[i];
        if (wave_uniform(nv * cw >= fma(smin, cw, cv * cgl) * nw)) {
            P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl; ++top;
            pv = cv; pw = cw; pt = ct; pl = cl; pgl = cgl;
            { const double lim = pv * pgl; plim = fma(smin, pw, lim > 0.0 ? lim : 0.0); }
            cv = nv; cw = nw; ct = nt; cl = nl; cgl = ngl;
            continue;
        }
        cv = fma(nv, cgl, cv); cw = fma(nw * cgl, cgl, cw); cl += nl; cgl *= ngl;
        while (wave_uniform(cv * pw < plim * cw)) {
            cv = fma(cv, pgl, pv); cw = fma(cw * pgl, pgl, pw); ct = pt; cl = pl + cl; cgl = pgl * cgl;
            --top;
            if (top > 0) {
                pv = P.v[top - 1]; pw = P.w[top - 1]; pt = P.t[top - 1]; pl = P.l[top - 1]; pgl = pow(g, (double)pl);
                const double lim = pv * pgl; plim = fma(smin, pw, lim > 0.0 ? lim : 0.0);
            } else plim = -INFINITY;
        }
    }
    P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl;
    P.n = top + 1;
}

// wave64 inclusive prefix sum of doubles on the DPP path: four row_shr steps inside each row of 16 lanes, then row_bcast:15 / row_bcast:31
// carry the row totals across (lanes without a source add 0).  Six ds_bpermute round trips of a shuffle-based scan are ~1500 clocks, this is ~150.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_scan_incl(double x) {
    x += dpp_take<0x111, 0xf>(x);      // row_shr:1
    x += dpp_take<0x112, 0xf>(x);      // row_shr:2
    x += dpp_take<0x114, 0xf>(x);      // row_shr:4
    x += dpp_take<0x118, 0xf>(x);      // row_shr:8
    x += dpp_take<0x142, 0xa>(x);      // row_bcast:15 into rows 1 and 3
    x += dpp_take<0x143, 0xc>(x);      // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ double lane_of(double x, int j) {                // x of lane j, j wave-uniform
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), j), hi = __builtin_amdgcn_readlane((int)(b >> 32), j);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// The same pass on the 64 lanes of wave 0.  Almost every sample of a real trace takes the "merge into the current pool" path (a 10^4-frame
// trace ends with ~10^2 pools), and a run of merges has a closed form: after j merges onto the pool (v, w, g^l)
//     v_j = v + g^l sum_{m<j} y_m g^m,      w_j = w + g^2l sum_{m<j} g^2m,      g^l_j = g^l g^j,
// so one wave-wide prefix sum gives the state in front of each of the next 64 samples as if all earlier ones had merged; every lane then
// evaluates both tests of its sample (push: oasisAR1.m:64-65; back-track after the merge: :83-85) and the first lane whose test fires
// ends the run.  The samples before it are merged exactly as computed, the event itself (a push, or the back-track loop) is handled
// on uniform values by all lanes, and the next block starts behind it.  Where events are dense (blocks that end within a few samples)
// the pass drops to the one-sample step for a stretch, so it is never much slower than oasis_cold.  The run's sums are accumulated
// in a different order than the sample-by-sample fma chain (relative differences of 1e-16 in v); the tests are otherwise the same.
__device__ void oasis_cold_wave(const float *y, double bsub, int T, double g, double smin, Pools &P, float *lds_scr, int nc) {
    const int lane = threadIdx.x;                                           // called by tid < 64
    double *lv = reinterpret_cast<double *>(lds_scr), *lw = lv + nc, *lg = lw + nc;
    int *lt = reinterpret_cast<int *>(lg + nc), *ll = lt + nc;
    double gm = 1.0, qm = 0.0;                                               // g^lane, sum_{m<lane} g^2m
    { const double g2 = g * g; double q = 1.0; for (int m = 0; m < lane; ++m) { gm *= g; qm += q; q *= g2; } }
    double cv = (double)y[0] - bsub, cw = 1.0, cgl = g;
    int ct = 1, cl = 1, top = 0;
    double pv = 0, pw = 1, pgl = 1, plim = -INFINITY; int pt = 0, pl = 0;
    auto push = [&](double nv, int tn) {                                    // current pool onto the stack, sample nv opens the next one
        if (lane == 0) {
            P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl;
            if (top < nc) { lv[top] = cv; lw[top] = cw; lg[top] = cgl; lt[top] = ct; ll[top] = cl; }
        }
        ++top;
        pv = cv; pw = cw; pt = ct; pl = cl; pgl = cgl;
        { const double lim = pv * pgl; plim = fma(smin, pw, lim > 0.0 ? lim : 0.0); }
        cv = nv; cw = 1.0; ct = tn; cl = 1; cgl = g;
    };
    auto backtrack = [&]() {
        while (wave_uniform(cv * pw < plim * cw)) {
            cv = fma(cv, pgl, pv); cw = fma(cw * pgl, pgl, pw); ct = pt; cl = pl + cl; cgl = pgl * cgl;
            --top;
            if (top > 0) {
                if (top - 1 < nc) { pv = lv[top - 1]; pw = lw[top - 1]; pgl = lg[top - 1]; pt = lt[top - 1]; pl = ll[top - 1]; }
                else { pv = P.v[top - 1]; pw = P.w[top - 1]; pt = P.t[top - 1]; pl = P.l[top - 1]; pgl = pow(g, (double)pl); }
                const double lim = pv * pgl; plim = fma(smin, pw, lim > 0.0 ? lim : 0.0);
            } else plim = -INFINITY;
        }
    };
    int i = 1, credit = 0;
    while (i < T) {
        if (credit > 0) {                                                   // one-sample step (dense events)
            const double nv = (double)y[i] - bsub;
            if (wave_uniform(nv * cw >= fma(smin, cw, cv * cgl))) push(nv, i + 1);
            else { cv = fma(nv, cgl, cv); cw = fma(cgl, cgl, cw); cl += 1; cgl *= g; backtrack(); }
            ++i; --credit;
            continue;
        }
        const int idx = i + lane;
        const bool in = idx < T;
        const double nv = in ? (double)y[idx] - bsub : 0.0;
        const double xs = nv * gm;
        const double ex = dpp_take<0x138, 0xf>(wave_scan_incl(xs));         // exclusive scan of y_m g^m (wave_shr:1; lane 0 takes 0)
        const double gl_j = cgl * gm;                                        // state in front of this lane's sample
        const double v_j = fma(cgl, ex, cv), w_j = fma(cgl * cgl, qm, cw);
        const bool ev_push = in && (nv * w_j >= fma(smin, w_j, v_j * gl_j));
        const double v_n = fma(nv, gl_j, v_j), w_n = fma(gl_j, gl_j, w_j), gl_n = gl_j * g;     // ... and behind it, merged
        const bool ev_back = in && !ev_push && (v_n * pw < plim * w_n);
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(ev_push || ev_back || !in);
        const unsigned long long pushes = __builtin_amdgcn_ballot_w64(ev_push);
        const int j = __builtin_amdgcn_readfirstlane(stop ? __builtin_ctzll(stop) : 64);                    // samples i .. i+j-1 merge
        if (j == 64) { cv = lane_of(v_n, 63); cw = lane_of(w_n, 63); cgl = lane_of(gl_n, 63); cl += 64; i += 64; continue; }
        if (i + j >= T) { cv = lane_of(v_j, j); cw = lane_of(w_j, j); cgl = lane_of(gl_j, j); cl += j; i += j; continue; }   // ran off the trace
        if ((pushes >> j) & 1) {                                            // the state the firing lane tested is the one that is kept
            cv = lane_of(v_j, j); cw = lane_of(w_j, j); cgl = lane_of(gl_j, j); cl += j;
            push(lane_of(nv, j), i + j + 1);
        } else {
            cv = lane_of(v_n, j); cw = lane_of(w_n, j); cgl = lane_of(gl_n, j); cl += j + 1;
            backtrack();
        }
        i += j + 1;
        if (j < 6) credit = 24;
    }
    if (lane == 0) { P.v[top] = cv; P.w[top] = cw; P.t[top] = ct; P.l[top] = cl; }
    P.n = top + 1;
}

__device__ __forceinline__ void oasis_seq(const float *y, double bsub, int T, double g, double lam, double smin, Pools &P, int warm, float *lds_scr, int nc) {
    if (warm) oasis_seq_t<true>(y, bsub, T, g, lam, smin, P, lds_scr, 0);
    else if (lam == 0.0) oasis_cold(y, bsub, T, g, smin, P, lds_scr, (nc * 24) / 32);
    else oasis_seq_t<false>(y, bsub, T, g, lam, smin, P, lds_scr, nc);   // (the cold pass is the long one: no per-sample `warm` tests in it)
}
// the pass from singleton pools: wave 0 enters
__device__ __forceinline__ void oasis_first(const float *y, double bsub, int T, double g, double lam, double smin, Pools &P, float *lds_scr, int nc24) {
    if (lam == 0.0) { oasis_cold_wave(y, bsub, T, g, smin, P, lds_scr, (nc24 * 24) / 32); return; }
    if (threadIdx.x == 0) oasis_seq_t<false>(y, bsub, T, g, lam, smin, P, lds_scr, nc24);
    P.n = __shfl(P.n, 0);
}

// round 6: a task is DTK = 31 samples (an ODD number; 64 until round 6) -- the tasks of a long pool start 63 words apart in the LDS copy of the trace, so the lanes of a wave that walk
// their tasks in step read different banks (64 apart they all read the SAME bank), and there are about as many tasks as k_deconv has threads (512).
constexpr int DTK = 31;
// split the pools into tasks of <= DTK samples: wave 0, 64 pools per round, task slots from a wave prefix sum of the per-pool counts
// (one lane walking the pool list paid a dependent global load per pool: 65 us for 115 pools)
__device__ __forceinline__ int build_tasks(const Pools &P, const DeconvIO &io, int64_t base2) {
    const int lane = threadIdx.x;                                           // called by tid < 64 with P.n uniform
    __threadfence_block();                                                  // lane 0 wrote the pools
    int nt = 0;
    for (int p0 = 0; p0 < P.n; p0 += 64) {
        const int p = p0 + lane;
        const int l = p < P.n ? P.l[p] : 0;
        const int k = (l + DTK - 1) / DTK;
        int inc = k;
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
        int at = nt + inc - k;
        for (int off = 0; off < l; off += DTK, ++at) {
            io.tk_pool[base2 + at] = p; io.tk_off[base2 + at] = off; io.tk_len[base2 + at] = l - off < DTK ? l - off : DTK;
        }
        nt += __shfl(inc, 63);
    }
    return nt;
}

// per-pool numerators num_p = sum_j yp(t_p + j) g^j, deterministic (task partials, then a per-pool ordered sum), and -- when hh is
// given -- the denominators hh_p = sum_{j<l_p} g^2j = cumsum(h.*h)(l_p) of foopsi_oasisAR1.m:166-174 from the same sweep.  tkv (2 doubles per
// task), num and hh are flat pointers: k_deconv places them in LDS when the pool list is short enough (a Brent step is then not three
// global-memory round trips long), else in the global scratch.
__device__ __forceinline__ double ipow(double x, int n) { double r = 1.0; for (; n; n >>= 1) { if (n & 1) r *= x; x *= x; } return r; }

// td (round 6): the tasks' descriptors staged in LDS by k_deconv after every build_tasks -- {first sample, len | (off / 64) << 8, pool, tasks of the pool if this is
// its first task else 0}.  An evaluation of Brent's objective used to start with three dependent global loads per task (task -> pool -> pool start) and its
// per-pool sum with two more: ~5 us per evaluation, 28 evaluations per trace, almost all of it load latency.  nullptr: the global arrays (pool lists too long for LDS).
__device__ __forceinline__ void pool_numerators(const float *y, double bsub, double lam, double g, const Pools &P, const DeconvIO &io, int64_t base2,
                                int ntask, double *tkv, double *num, double *hh, const int4 *td = nullptr) {
    const int tid = threadIdx.x, NTH = (int)blockDim.x;
    double g64;                                                  // g^DTK: a task starts at a multiple of DTK samples into its pool
    const double g2 = g * g;
    const double g4 = g2 * g2;
    { const double g8 = g4 * g4, g16 = g8 * g8; g64 = (((g16 * g8) * g4) * g2) * g; static_assert(DTK == 31, "g^DTK is spelled out for 31"); }
    const double S63 = g2 < 1.0 ? (1.0 - g64 * g64) / (1.0 - g2) : (double)DTK;      // sum_{j < DTK} g^2j
    const double shift = bsub + lam * (1 - g);
    for (int k = tid; k < ntask; k += NTH) {
        int t0, len, o64;
        if (td) { const int4 d = td[k]; t0 = d.x; len = d.y & 255; o64 = d.y >> 8; }
        else { const int p = io.tk_pool[base2 + k], off = io.tk_off[base2 + k]; len = io.tk_len[base2 + k]; t0 = P.t[p] - 1 + off; o64 = off / DTK; }
        // g^off by squaring; the task's sum in Horner form, s = g^off (x_0 + g (x_1 + g (x_2 + ...))), x_j = yp(t0 + j) (as four chains, below) -- one fma per sample instead of a product, a
        // running power and two accumulations (round 6: an evaluation of Brent's objective was 3.7 us of vector-pipe issue on ONE wave per SIMD, 28 evaluations per
        // trace) -- and the squares' sum in closed form, sum_{j < len} g^(2 (off + j)) = g^(2 off) (1 - g^(2 len)) / (1 - g^2), S63 = the factor of a whole task
        double goff = 1.0;
        { double bs = g64; for (int e = o64; e; e >>= 1) { if (e & 1) goff *= bs; bs *= bs; } }
        // FOUR interleaved Horner chains in g^4 (samples j = r mod 4): a dependent fp64 fma every ~30 clocks was what an evaluation waited for, 63 in a row
        double H0 = 0.0, H1 = 0.0, H2 = 0.0, H3 = 0.0;
        for (int m1 = (len + 3) >> 2; m1 > 0; m1 -= 2) {         // two groups of four per batch of reads, last group first; samples behind the task's end count as 0
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = 4 * (m1 - 1 - (u >> 2)) + (u & 3); v[u] = y[t0 + (j >= 0 && j < len ? j : 0)]; }
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = 4 * (m1 - 1 - (u >> 2)) + (u & 3); x[u] = j < len ? (double)v[u] - shift : 0.0; }
            H0 = fma(H0, g4, x[0]); H1 = fma(H1, g4, x[1]); H2 = fma(H2, g4, x[2]); H3 = fma(H3, g4, x[3]);
            if (m1 >= 2) { H0 = fma(H0, g4, x[4]); H1 = fma(H1, g4, x[5]); H2 = fma(H2, g4, x[6]); H3 = fma(H3, g4, x[7]); }
        }
        const double H = (H0 + g * H1) + g2 * (H2 + g * H3);
        const double s = goff * H;
        const double h2 = goff * goff * (len == DTK ? S63 : (g2 < 1.0 ? (1.0 - ipow(g2, len)) / (1.0 - g2) : (double)len));
        tkv[2 * k] = s; tkv[2 * k + 1] = h2;
    }
    __syncthreads();
    for (int k = tid; k < ntask; k += NTH) {
        int p, nk;
        if (td) { const int4 d = td[k]; if (d.w == 0) continue; p = d.z; nk = d.w; }
        else {
            if (io.tk_off[base2 + k] != 0) continue;             // first task of its pool sums the pool's tasks in order
            p = io.tk_pool[base2 + k];
            nk = (P.l[p] + DTK - 1) / DTK;                       // (its task count from the pool length: no load-dependent loop exit)
        }
        double s = 0, h2 = 0;
        for (int q0 = 0; q0 < nk; q0 += 8) {                     // (eight partials read at a time, added in order: a long pool -- a silent stretch of the trace -- has
            double2 pv[8];                                       //  tens of tasks, and a read per addition made its first task's thread the evaluation's critical path)
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = *reinterpret_cast<const double2 *>(tkv + 2 * (k + (q0 + u < nk ? q0 + u : nk - 1)));
#pragma unroll
            for (int u = 0; u < 8; ++u) if (q0 + u < nk) { s += pv[u].x; h2 += pv[u].y; }
        }
        num[p] = s;
        if (hh) hh[p] = h2;
    }
    __syncthreads();
}

__device__ __forceinline__ double hh_of(double g, int l) {       // cumsum(h.*h)(l) = sum_{j<l} g^(2j)
    double s = 0, q = 1.0; const double g2 = g * g;
    if (l > 512) return (1.0 - pow(g2, (double)l)) / (1.0 - g2);
    for (int j = 0; j < l; ++j) { s += q; q *= g2; }
    return s;
}

// LONG: the trace does not fit LDS beside the Welch transform (T > 18436) -- it and the output staging live in a per-slot global buffer
// (L2-resident: a few hundred KB per workgroup) and LDS only holds the scratch.  Same code; every access to y goes through a pointer whose
// address space the compiler infers per instantiation.
// round 6: 512 threads per trace (256 until then).  Every parallel phase of the kernel is a short dependent chain per thread (fp64 latency, LDS round trips), and a
// workgroup has a CU to itself -- 20 to 100 traces per level on 256 CUs: with ONE wave per SIMD nothing hides those latencies (a 31-sample task of Brent's objective
// took 3 us).  Two waves per SIMD and tasks half as long; the register budget (256) stays.
constexpr int DECONV_NT = 512;
template <bool LONG, bool SPLIT = false>
__global__ void __launch_bounds__(DECONV_NT) k_deconv(DeconvCfg c, DeconvIO io) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double red[DECONV_NT / 64];
    __shared__ int sh_i[4];
    const int tid = threadIdx.x, T = c.T, NTH = DECONV_NT;
    const int slot = blockIdx.x;
    const long long tc0 = wall_clock64();                    // (option deconv_trace: thread 0 of the traced trace prints where its time goes, 100 MHz ticks)
    long long tcl = tc0;
    int k;
    if (io.jobs) {                                         // several jobs in one launch: this workgroup's trace belongs to job e.x
        const int2 e = io.jlist[slot];
        const DeconvJobDev j = io.jobs[e.x];
        k = e.y;
        io.C = j.C; io.Craw = j.Craw; io.S = j.S; io.ldc = j.ldc; io.U = j.U; io.nptr = j.nptr; io.nidx = j.nidx; io.nval = j.nval; io.aa = j.aa;
        io.pars = j.pars; io.sn_out = j.sn_out; io.b_out = j.b_out;
    } else k = io.list[slot];
    // bit 30 of a list entry: this item belongs to the LAST sweep of the update (HALS_temporal.m:99-102 keeps S and C_raw of that one) -- the sweeps of one update are
    // scheduled as ONE dependency graph over (sweep, trace) items (factor.hip, dag_schedule), so a launch can hold items of different sweeps
    c.last |= __builtin_amdgcn_readfirstlane(k >> 30) & 1;
    k &= 0x3fffffff;
    const int Tal = (T + 3) & ~3;
    float *y = LONG ? io.ybuf + (int64_t)blockIdx.x * Tal : sm;          // T raw samples (fp32), persistent
    float *scr = LONG ? sm : sm + Tal;               // scratch: select histogram | FFT re, im, twiddles, window (4 nfft) | top of the OASIS pool stack
    const size_t scr_bytes = (size_t)(LONG ? (c.ylong == 2 ? (SPLIT ? 1 : 2) : 4) * c.nfft : max(4 * c.nfft, Tal)) * sizeof(float);
    float *tabs = (LONG && c.ylong == 2) ? io.tbuf + (int64_t)blockIdx.x * 3 * c.nfft : nullptr;      // nfft >= 16384: the Welch tables do not fit beside re | im
    float *ostage = LONG ? io.obuf + (int64_t)blockIdx.x * Tal : scr;   // the solution c(t), before it is written out
    const int nc_pools = (int)(scr_bytes / 24);
    const int64_t base = (int64_t)slot * T, base2 = (int64_t)slot * 2 * T;
    Pools P; P.v = io.pv + base; P.w = io.pw + base; P.t = io.pt + base; P.l = io.pl + base; P.n = 0;
    // per-task partial sums, per-pool numerators and denominators of update_g: the tail of scr when 48 B per pool (32 for the staged warm
    // pass at its head) and 16 B per task fit, else the global scratch (then without the denominators: hh_of() per pool)
    double *tkv = io.tk_val + 2 * base2, *num = io.pnum + base, *hhs = nullptr;          // (tk_val: 4 T doubles per slot, two per task)
    bool in_lds = false;
    int ntask = 0;
    int4 *td = nullptr;                             // the tasks' descriptors in LDS (pool_numerators)
    auto place = [&]() {                            // after every build_tasks: P.n and ntask changed.  Called by every thread, behind a barrier
        in_lds = (size_t)48 * P.n + (size_t)32 * ntask + 16 <= scr_bytes;
        if (in_lds) {
            double *end = reinterpret_cast<double *>(reinterpret_cast<char *>(scr) + scr_bytes);
            tkv = end - 2 * ntask; num = tkv - P.n; hhs = num - P.n;
            td = reinterpret_cast<int4 *>((reinterpret_cast<uintptr_t>(hhs) - (size_t)16 * ntask) & ~(uintptr_t)15);
            for (int q = tid; q < ntask; q += NTH) {
                const int p = io.tk_pool[base2 + q], off = io.tk_off[base2 + q], len = io.tk_len[base2 + q];
                td[q] = make_int4(P.t[p] - 1 + off, len | ((off / DTK) << 8), p, off == 0 ? (P.l[p] + DTK - 1) / DTK : 0);
            }
            __syncthreads();
        } else { tkv = io.tk_val + 2 * base2; num = io.pnum + base; hhs = nullptr; td = nullptr; }
    };

    // ---- raw trace into LDS; HALS: ck_raw = C(k,:) + (U(k,:) - V(k,:)*C)/aa(k)  (HALS_temporal.m:62) ----
    float *ck = io.C + (int64_t)k * io.ldc;
    if (c.hals) {
        const float a = io.aa[k];
        const int n0 = io.nptr[k], n1 = io.nptr[k + 1];
        for (int t = tid; t < T; t += NTH) {
            float vc = 0.f;
            for (int j = n0; j < n1; ++j) vc = fmaf(io.nval[j], io.C[(int64_t)io.nidx[j] * io.ldc + t], vc);
            y[t] = ck[t] + (io.U[(int64_t)k * io.ldc + t] - vc) / a;
        }
    } else {
        const float *src = io.Craw + (int64_t)k * io.ldc;
        for (int t = tid; t < T; t += NTH) y[t] = src[t];
    }
    __syncthreads();
    // deconv_trace = 1000000 + k + 1: TIMING of trace k only (no 'DT' lines: a device printf costs tens of microseconds) -- the laps are kept and printed at the end
    const bool timing = c.trace == 1000000 + k + 1 && tid == 0;
    // (phase ids, named by scripts/deconv_phases.py: 0 row update + load, 1 quantile + median, 2 GetSn, 3 time constant, 4 cold OASIS pass + tasks,
    //  5 b + Brent over g, 6 warm-started pass + tasks, 7 solution + outputs, 20 = evaluations of Brent's objective in the search that follows)
    int lap_id[24]; float lap_us[24]; int nlap = 0;
    auto lap = [&](int what) {
        if (timing && nlap < 24) { const long long t = wall_clock64(); lap_id[nlap] = what; lap_us[nlap] = (float)(t - tcl) * 0.01f; ++nlap; tcl = t; }
    };
    lap(0);
    // NaN guard of deconvTemporal.m:37-40
    int bad = 0;
    for (int t = tid; t < T; t += NTH) bad |= !(y[t] == y[t]);
    bad = __syncthreads_or(bad);
    if (bad) {
        for (int t = tid; t < T; t += NTH) { ck[t] = 0.f; io.S[(int64_t)k * io.ldc + t] = 0.f; if (io.Craw) io.Craw[(int64_t)k * io.ldc + t] = 0.f; }
        return;
    }
    // ---- median (HALS_temporal.m:78) and 15 % quantile (foopsi_oasisAR1.m:93) of the raw trace ----
    double bsub = 0.0;                               // everything subtracted from the raw trace so far
    double q15 = 0.0;
    if (c.optimize_b) {
        const double q15pos = 0.15 * T - 0.5;
        const int qi = q15pos <= 0 ? 0 : (q15pos >= T - 1 ? T - 1 : (int)floor(q15pos));
        float v0, v1;
        select_pair(y, T, qi, reinterpret_cast<int *>(scr), v0, v1);
        q15 = (q15pos <= 0 || q15pos >= T - 1) ? (double)v0 : (double)v0 + (q15pos - qi) * ((double)v1 - (double)v0);
    }
    if (c.hals) {
        float m0, m1;
        select_pair(y, T, (T - 1) / 2, reinterpret_cast<int *>(scr), m0, m1);
        const float med = 0.5f * (m0 + ((T & 1) ? m0 : m1));
        double s = 0, n = 0;
        for (int t = tid; t < T; t += NTH) if (y[t] < med) { s += y[t]; n += 1; }
        s = block_sum(s, red); n = block_sum(n, red);
        bsub = s / n;                                // b = mean(ck_raw(ck_raw < median(ck_raw)))
    }
    __syncthreads();
    lap(1);
    // ---- noise level (GetSn on the raw trace: HALS_temporal.m:79, deconvTemporal.m:45) ----
    const double sn = get_sn<decltype(y), 17, SPLIT>(y, c, scr, red, true, tabs);
    lap(2);
    // ---- time constant (deconvolveCa.m:73-89) ----
    double g = (double)io.pars[k];
    if (g == 0.0) {
        g = est_g(y, bsub, T, sn, red);
        if (g < -1.0) {                              // no stable AR(1): c = s = 0, pars = 0
            for (int t = tid; t < T; t += NTH) {
                const float raw = (float)((double)y[t] - bsub);
                ck[t] = raw;                         // "if sum(abs(ck))==0, ck = ck_raw" (HALS_temporal.m:95-97, deconvTemporal.m:53-55)
                if (!c.hals || c.last) { io.S[(int64_t)k * io.ldc + t] = 0.f; io.Craw[(int64_t)k * io.ldc + t] = raw; }
            }
            if (tid == 0) { io.pars[k] = 0.f; io.sn_out[k] = (float)sn; io.b_out[k] = 0.f; }
            return;
        }
    }
    lap(3);
    const double smin = c.smin_opt < 0 ? -c.smin_opt * sn : c.smin_opt;       // deconvolveCa.m:116-118
    const double lam = c.lam;
    double mean_y = 0;
    for (int t = tid; t < T; t += NTH) mean_y += (double)y[t] - bsub;
    mean_y = block_sum(mean_y, red) / T;

    // ---- foopsi_oasisAR1.m:82-117 ----
    double b = c.optimize_b ? (q15 - bsub) : 0.0;    // :93 quantile(y, .15) of the baseline-subtracted trace
    int optimize_g = c.optimize_g;
    if (tid < 64) { oasis_first(y, bsub + b, T, g, lam, smin, P, scr, nc_pools); const int nt = build_tasks(P, io, base2); if (tid == 0) { sh_i[0] = P.n; sh_i[1] = nt; } }
    __syncthreads();
    P.n = sh_i[0]; ntask = sh_i[1]; place();
    lap(4);
    const bool trc = c.trace == k + 1 && tid == 0;
    if (trc) printf("DT first sn %.17g g %.17g b %.17g bsub %.17g smin %.17g pools %d\n", sn, g, b, bsub, smin, P.n);
    const int niter = c.optimize_b ? c.maxIter : (optimize_g ? 1 : 0);
    for (int it = 0; it < niter; ++it) {
        // sum of the current solution: c(t) = max(0, v/w) g^j on each pool
        double ssol = 0;
        for (int q = tid; q < ntask; q += NTH) {
            const int p = io.tk_pool[base2 + q], off = io.tk_off[base2 + q], len = io.tk_len[base2 + q];
            const double r = P.v[p] / P.w[p];
            double gj = pow(g, (double)off) * (r > 0 ? r : 0.0);
            for (int j = 0; j < len; ++j) { ssol += gj; gj *= g; }
        }
        ssol = block_sum(ssol, red);
        if (c.optimize_b) b = mean_y - ssol / T;     // :98 b = mean(y - solution)
        if (trc) printf("DT it %d b %.17g\n", it, b);
        if (!optimize_g) break;                      // :113-115
        const double g0 = g;
        if (g > c.gmax) {                            // :104-108
            const double sn2 = get_sn<decltype(y), 17, SPLIT>(y, c, scr, red, true, tabs);
            const double g2 = est_g(y, bsub, T, sn2, red);
            if (g2 >= -1.0) g = g2;
            if (tid < 64) { oasis_first(y, bsub + b, T, g, lam, smin, P, scr, nc_pools); const int nt = build_tasks(P, io, base2); if (tid == 0) { sh_i[0] = P.n; sh_i[1] = nt; } }
            __syncthreads();
            P.n = sh_i[0]; ntask = sh_i[1]; place();
            break;
        }
        // ---- update_g (:122-180): Brent's fminbnd of rss(g) on [0,1] ----
        double sumy2 = 0;
        for (int t = tid; t < T; t += NTH) { const double v = (double)y[t] - (bsub + b); sumy2 += v * v; }
        sumy2 = block_sum(sumy2, red);
        int nev = 0;
        auto rss = [&](double gg) -> double {
            pool_numerators(y, bsub + b, lam, gg, P, io, base2, ntask, tkv, num, hhs, td);
            double s = 0;
            for (int p = tid; p < P.n; p += NTH) { const double nm = num[p]; if (nm > 0) s += nm * nm / (hhs ? hhs[p] : hh_of(gg, P.l[p])); }
            s = block_sum(s, red);
            ++nev;
            return sumy2 - s;                        // ||y - c||^2 with c = max(num/hh, 0) h on every pool (lam = 0 form)
        };
        double xf, glast;
        {   // Brent (fminbnd), TolX = 1e-4; all threads run the same scalar code
            // NO FMA CONTRACTION in this block (hipcc's default is -ffp-contract=fast): on the second step v == w and fv == fw, the parabola is
            // degenerate and r == q EXACTLY, so p = 0, q = 0 and fminbnd falls through to a golden-section step.  Contracted, `(xf - v) * q - (xf - w) * r`
            // and `2 * (q - r)` keep the products' rounding errors (1e-17 of each other), their ratio is an arbitrary O(1) number that passes the
            // acceptance tests, and the engine took a "parabolic" step to a random point where the reference takes the golden one
            // (profiles/r03/deconv_trace_b.txt: evaluation 2 at 0.4249 instead of 0.7639) -- the source of gamma differing by up to 2e-3.
#pragma clang fp contract(off)
            const double seps = 1.4901161193847656e-08, cgold = 0.3819660112501051, tol = 1e-4;
            double a = 0.0, bb = 1.0;
            double v = a + cgold * (bb - a), w = v; xf = v;
            double d = 0.0, e = 0.0, x = xf;
            double fx = rss(x); glast = x;
            if (trc) printf("DT brent %d x %.17g f %.17g\n", 0, x, fx);
            double fv = fx, fw = fx;
            double xm = 0.5 * (a + bb), tol1 = seps * fabs(xf) + tol / 3.0, tol2 = 2.0 * tol1;
            int iter = 0;
            while (fabs(xf - xm) > (tol2 - 0.5 * (bb - a)) && iter < 500) {
                ++iter;
                bool gs = true;
                if (fabs(e) > tol1) {
                    gs = false;
                    double r = (xf - w) * (fx - fv), q = (xf - v) * (fx - fw), pq = (xf - v) * q - (xf - w) * r;
                    q = 2.0 * (q - r);
                    if (q > 0.0) pq = -pq;
                    q = fabs(q); r = e; e = d;
                    if (fabs(pq) < fabs(0.5 * q * r) && pq > q * (a - xf) && pq < q * (bb - xf)) {
                        d = pq / q; x = xf + d;
                        if ((x - a) < tol2 || (bb - x) < tol2) { const double si = (xm - xf) >= 0 ? 1.0 : -1.0; d = tol1 * si; }
                    } else gs = true;
                }
                if (gs) { e = xf >= xm ? a - xf : bb - xf; d = cgold * e; }
                const double si = d >= 0 ? 1.0 : -1.0;
                x = xf + si * fmax(fabs(d), tol1);
                const double fu = rss(x); glast = x;
                if (trc) printf("DT brent %d x %.17g f %.17g\n", iter, x, fu);
                if (fu <= fx) { if (x >= xf) a = xf; else bb = xf; v = w; fv = fw; w = xf; fw = fx; xf = x; fx = fu; }
                else {
                    if (x < xf) a = x; else bb = x;
                    if (fu <= fw || w == xf) { v = w; fv = fw; w = x; fw = fu; }
                    else if (fu <= fv || v == xf || v == w) { v = x; fv = fu; }
                }
                xm = 0.5 * (a + bb); tol1 = seps * fabs(xf) + tol / 3.0; tol2 = 2.0 * tol1;
            }
        }
        g = xf;
        if (timing && nlap < 24) { lap_id[nlap] = 20; lap_us[nlap] = (float)nev; ++nlap; }      // (id 20: evaluations of the objective in this Brent search)
        lap(5);
        // warm-started pools: v = yp' * h(g), w = cumsum(h_last.^2)(l) with h_last from the LAST rss_g call (:155-161, sic)
        pool_numerators(y, bsub + b, lam, g, P, io, base2, ntask, tkv, num, nullptr, td);   // (hhs keeps the LAST rss_g call's sums)
        const bool staged = in_lds;
        double *sv = reinterpret_cast<double *>(scr), *sw = sv + P.n, *sg = sw + P.n;
        int *st = reinterpret_cast<int *>(sg + P.n), *sl = st + P.n;
        for (int p = tid; p < P.n; p += NTH) {
            const int l = P.l[p];
            const double v = num[p], w = hhs ? hhs[p] : hh_of(glast, l);
            P.v[p] = v; P.w[p] = w;
            if (staged) { sv[p] = v; sw[p] = w; sg[p] = pow(g, (double)l); st[p] = P.t[p]; sl[p] = l; }
        }
        __syncthreads();
        if (tid < 64) {
            if (tid == 0) { if (staged) oasis_warm(P.n, g, smin, P, sv, sw, sg, st, sl); else oasis_seq(y, bsub + b, T, g, lam, smin, P, 1, scr, nc_pools); }
            P.n = __shfl(P.n, 0);
            const int nt = build_tasks(P, io, base2);
            if (tid == 0) { sh_i[0] = P.n; sh_i[1] = nt; }
        }
        __syncthreads();
        P.n = sh_i[0]; ntask = sh_i[1]; place();
        lap(6);
        if (trc) printf("DT updated g %.17g glast %.17g pools %d\n", g, glast, P.n);
        if (fabs(g - g0) / g0 < 1e-3) optimize_g = 0;            // :110-112
        if (!c.optimize_b) break;
    }
    // ---- solution (oasisAR1.m:100-109) and outputs ----
    float *so = io.S + (int64_t)k * io.ldc;
    const bool wr = !c.hals || c.last;
    double sabs = 0;
    for (int q = tid; q < ntask; q += NTH) {
        const int p = io.tk_pool[base2 + q], off = io.tk_off[base2 + q], len = io.tk_len[base2 + q];
        const double r = P.v[p] / P.w[p];
        double gj = pow(g, (double)off) * (r > 0 ? r : 0.0);
        const int t0 = P.t[p] - 1 + off;
        for (int j = 0; j < len; ++j) { double cv = gj; if (!(cv == cv) || isinf(cv)) cv = 0.0; ostage[t0 + j] = (float)cv; sabs += fabs(cv); gj *= g; }
    }
    sabs = block_sum(sabs, red);
    __syncthreads();
    const double btot = bsub + b;                     // HALS: ck_raw - b - tmp_options.b ; deconvTemporal: ck_raw - options.b
    for (int t = tid; t < T; t += NTH) {
        const float raw = (float)((double)y[t] - btot);
        ck[t] = sabs == 0.0 ? raw : ostage[t];
        if (wr) { so[t] = 0.f; io.Craw[(int64_t)k * io.ldc + t] = raw; }
    }
    __syncthreads();
    if (wr)
        for (int p = 1 + tid; p < P.n; p += NTH) {    // s(t_p) = c(t_p) - g c(t_p - 1) at pool starts
            const int t0 = P.t[p] - 1;
            so[t0] = (float)((double)ostage[t0] - g * (double)ostage[t0 - 1]);
        }
    if (tid == 0) { io.pars[k] = (float)g; io.sn_out[k] = (float)sn; io.b_out[k] = (float)b; }
    lap(7);
    if (timing) {
        const double tot = (double)(wall_clock64() - tc0) * 0.01;
        for (int i = 0; i < nlap; ++i) printf("DTT %d %d %.2f\n", k, lap_id[i], (double)lap_us[i]);
        printf("DTT %d 99 %.2f\n", k, tot);
    }
}

// ---- S5: per-pixel noise of the resident residual, sn = GetSn(Ysig)  (update_spatial_parallel.m:191-194) -------
// One workgroup per patch pixel: its trace is gathered out of Ysig4 (4 frames per 16-byte load) into LDS, then the
// same Welch estimator as for the traces.
__global__ void __launch_bounds__(256) k_sn_pixels(DeconvCfg c, const float4 *__restrict__ ysig4, int64_t d, float *__restrict__ sn) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[4];
    const int64_t m = blockIdx.x;
    const int tid = threadIdx.x, T = c.T;
    const int Tal = (T + 3) & ~3;
    float *y = lds, *scr = lds + Tal;
    for (int q = tid; q < (T + 3) / 4; q += 256) *reinterpret_cast<float4 *>(y + 4 * q) = ysig4[(int64_t)q * d + m];
    __syncthreads();
    const double v = get_sn(y, c, scr, red, false);
    if (tid == 0) sn[m] = (float)v;
}
// Long recordings (trace + transform beyond 160 KB of LDS: T > 20400): the trace stays where it is -- the windowed segments are read out of the interleaved
// video (every sample twice: the segments overlap by half) -- and LDS holds the transform alone; with nfft = 16384 (T > 36868) the twiddle table lives in a
// per-workgroup slot of global memory.  `add`: the pixel mean for the RAW video (estimate_noise), nullptr / 0 for Ysig.
template <bool SPLIT>
__global__ void __launch_bounds__(256) k_sn_pixels_long(DeconvCfg c, const float4 *__restrict__ v4, int64_t npix, const float *__restrict__ add, float *__restrict__ tabs, float *__restrict__ sn) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[4];
    for (int64_t m = blockIdx.x; m < npix; m += gridDim.x) {
        Ysig4Acc y{v4 + m, npix, add ? add[m] : 0.f};
        const double v = get_sn<Ysig4Acc, SPLIT ? 33 : 17, SPLIT>(y, c, lds, red, false, tabs ? tabs + (int64_t)blockIdx.x * 3 * c.nfft : nullptr);
        if (threadIdx.x == 0) sn[m] = (float)v;
        __syncthreads();
    }
}
// the per-pixel GetSn of a video of `npix` pixels ([T/4][npix] float4), the first T frames: picks the LDS-resident or the long flavour
static int sn_pixels_launch(cnmfe_ctx *ctx, const char *name, const float4 *v4, int64_t npix, int64_t T, const float *add_mean, float *dSn) {
    if (T < 64 || T > WELCH_TMAX) return fail(CNMFE_EUNSUPPORTED, "GetSn on the device supports 64 <= T <= %d frames (got %lld)", WELCH_TMAX, (long long)T);
    DeconvCfg c{};
    c.T = (int)T; c.P2 = 1; while (c.P2 < T) c.P2 <<= 1;
    c.L = (int)(T / 4.5); c.nov = c.L / 2;                                   // pwelch defaults (MathWorks documentation)
    c.nfft = 256; while (c.nfft < c.L) c.nfft <<= 1;
    c.nseg = (int)((T - c.nov) / (c.L - c.nov));
    c.split = c.nfft > 16384 ? 1 : 0;                                        // (73728 < T <= 147456: two 16384-point transforms per pair of segments)
    const size_t shmem = ((((size_t)T + 3) & ~size_t(3)) + 2 * (size_t)c.nfft + (size_t)c.nfft / 2) * sizeof(float);
    if (shmem <= 160 * 1024 - 256) return 1;                                 // the LDS-resident kernels (the caller launches its own: raw video / Ysig differ in the load)
    const bool tab_global = (2 * (size_t)c.nfft + (size_t)c.nfft / 2) * sizeof(float) > 160 * 1024 - 256;
    const size_t sh = (tab_global ? (c.split ? 1 : 2) * (size_t)c.nfft : 2 * (size_t)c.nfft + (size_t)c.nfft / 2) * sizeof(float);
    const unsigned nwg = (unsigned)std::min<int64_t>(npix, 1024);
    float *tabs = nullptr;
    if (tab_global) { RET(ctx->dscr.tbuf.ensure((size_t)nwg * 3 * c.nfft * sizeof(float))); tabs = ctx->dscr.tbuf.as<float>(); }
    if (c.split) {
        if (sh > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_sn_pixels_long<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        LAUNCH(ctx, name, k_sn_pixels_long<true>, dim3(nwg), dim3(256), sh, c, v4, npix, add_mean, tabs, dSn);
        return 0;
    }
    if (sh > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_sn_pixels_long<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    LAUNCH(ctx, name, k_sn_pixels_long<false>, dim3(nwg), dim3(256), sh, c, v4, npix, add_mean, tabs, dSn);
    return 0;
}

int sn_pixels_run(cnmfe_ctx *ctx, Patch *P, float *sn_out) {
    const int64_t T = P->T;
    {   // long recordings: the transform alone in LDS
        DevBuf &dSnL = ctx->tmp[14];
        RET(dSnL.ensure((size_t)P->d * sizeof(float)));
        const int rcl = sn_pixels_launch(ctx, "spatial_sn_pixels", P->ysig.as<float4>(), P->d, T, nullptr, dSnL.as<float>());
        if (rcl < 0) return rcl;
        if (rcl == 0) {
            CK(hipMemcpyAsync(sn_out, dSnL.p, (size_t)P->d * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
            return ctx_check_errflag(ctx);
        }
    }
    DeconvCfg c{};
    c.T = (int)T; c.P2 = 1; while (c.P2 < T) c.P2 <<= 1;
    c.L = (int)(T / 4.5); c.nov = c.L / 2;                                   // pwelch defaults (MathWorks documentation)
    c.nfft = 256; while (c.nfft < c.L) c.nfft <<= 1;
    c.nseg = (int)((T - c.nov) / (c.L - c.nov));
    const size_t shmem = ((((size_t)T + 3) & ~size_t(3)) + 2 * (size_t)c.nfft + (size_t)c.nfft / 2) * sizeof(float);
    if (shmem > 160 * 1024 - 256) return fail(CNMFE_EUNSUPPORTED, "trace of %lld frames does not fit the GetSn kernel's LDS (trace + Welch transform in 160 KB: T <= 20400)", (long long)T);
    if (shmem > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_sn_pixels, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    DevBuf &dSn = ctx->tmp[14];
    RET(dSn.ensure((size_t)P->d * sizeof(float)));
    LAUNCH(ctx, "spatial_sn_pixels", k_sn_pixels, dim3((unsigned)P->d), dim3(256), shmem, c, P->ysig.as<float4>(), P->d, dSn.as<float>());
    CK(hipMemcpyAsync(sn_out, dSn.p, (size_t)P->d * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    return ctx_check_errflag(ctx);
}

// ---- P.sn = estimate_noise(obj) (Sources2D.m:328-379, method 'psd'): GetSn of the first frames of the RAW video, per block pixel ------
// The resident video is centred, so the pixel mean goes back in (pwelch does not detrend: what the mean leaks through the Hamming window
// into [0.25, 0.5] is part of the reference's number).
__global__ void __launch_bounds__(256) k_sn_video(DeconvCfg c, const float4 *__restrict__ yc4, int64_t d_b, const float *__restrict__ ymean_f, float *__restrict__ sn) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[4];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, T = c.T;
    const int Tal = (T + 3) & ~3;
    float *y = lds, *scr = lds + Tal;
    const float mu = ymean_f[q];
    for (int i = tid; i < (T + 3) / 4; i += 256) { const float4 v = yc4[(int64_t)i * d_b + q]; *reinterpret_cast<float4 *>(y + 4 * i) = make_float4(v.x + mu, v.y + mu, v.z + mu, v.w + mu); }
    __syncthreads();
    const double v = get_sn(y, c, scr, red, false);
    if (tid == 0) sn[q] = (float)v;
}

int sn_video_run(cnmfe_ctx *ctx, Patch *P, int64_t nframes, float *sn_out) {
    const int64_t T = nframes;
    if (T < 64 || T > P->T) return fail(CNMFE_EUNSUPPORTED, "estimate_noise on the device supports 64 <= frames <= T (got %lld of %lld)", (long long)T, (long long)P->T);
    {
        DevBuf &dSnL = ctx->tmp[14];
        RET(dSnL.ensure((size_t)P->d_b * sizeof(float)));
        const int rcl = sn_pixels_launch(ctx, "estimate_noise", P->Yc4.as<float4>(), P->d_b, T, P->ymean_f.as<float>(), dSnL.as<float>());
        if (rcl < 0) return rcl;
        if (rcl == 0) {
            CK(hipMemcpyAsync(sn_out, dSnL.p, (size_t)P->d_b * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
            CK(hipStreamSynchronize(ctx->st()));
            return 0;
        }
    }
    DeconvCfg c{};
    c.T = (int)T; c.P2 = 1; while (c.P2 < T) c.P2 <<= 1;
    c.L = (int)(T / 4.5); c.nov = c.L / 2;                                   // pwelch defaults (MathWorks documentation)
    c.nfft = 256; while (c.nfft < c.L) c.nfft <<= 1;
    c.nseg = (int)((T - c.nov) / (c.L - c.nov));
    const size_t shmem = ((((size_t)T + 3) & ~size_t(3)) + 2 * (size_t)c.nfft + (size_t)c.nfft / 2) * sizeof(float);
    if (shmem > 160 * 1024 - 256) return fail(CNMFE_EUNSUPPORTED, "%lld frames do not fit the GetSn kernel's LDS (trace + Welch transform in 160 KB: <= 20400)", (long long)T);
    if (shmem > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_sn_video, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    DevBuf &dSn = ctx->tmp[14];
    RET(dSn.ensure((size_t)P->d_b * sizeof(float)));
    LAUNCH(ctx, "estimate_noise", k_sn_video, dim3((unsigned)P->d_b), dim3(256), shmem, c, P->Yc4.as<float4>(), P->d_b, P->ymean_f.as<float>(), dSn.as<float>());
    CK(hipMemcpyAsync(sn_out, dSn.p, (size_t)P->d_b * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    return 0;
}

// ---- host side -------------------------------------------------------------------------------------------

int deconv_setup(const cnmfe_deconv_opts *o, int64_t T, int in_sweep, DeconvCfg &c, size_t &shmem) {
    if (!o) return fail(CNMFE_EINVAL, "null deconvolution options");
    if (o->type != 1 || o->method != 1) return fail(CNMFE_EUNSUPPORTED, "only type 'ar1' / method 'foopsi' is built (demo_large_data_1p.m:38-43)");
    if (T < 64 || T > WELCH_TMAX) return fail(CNMFE_EUNSUPPORTED, "deconvolution supports 64 <= T <= %d frames (got %lld)", WELCH_TMAX, (long long)T);
    c.T = (int)T; c.P2 = 1; while (c.P2 < T) c.P2 <<= 1;
    c.L = (int)(T / 4.5); c.nov = c.L / 2;
    c.nfft = 256; while (c.nfft < c.L) c.nfft <<= 1;
    c.nseg = (int)((T - c.nov) / (c.L - c.nov));
    c.split = c.nfft > 16384 ? 1 : 0;
    c.maxIter = in_sweep ? 20 : (o->maxIter > 0 ? o->maxIter : 10);          // HALS_temporal.m:92 passes 'maxIter', 20
    c.optimize_b = o->optimize_b; c.optimize_g = o->optimize_pars;
    c.smin_opt = o->smin; c.lam = o->lambda; c.gmax = exp(-1.0 / (o->max_tau > 0 ? o->max_tau : 100.0));
    c.hals = in_sweep; c.last = 0; c.trace = 0;
    if (o->lambda != 0.0) return fail(CNMFE_EUNSUPPORTED, "lambda != 0 is not built");
    const size_t scr = std::max<size_t>(4 * (size_t)c.nfft, ((size_t)T + 3) & ~size_t(3));     // FFT + tables | output staging; pool mirrors use what there is
    shmem = ((((size_t)T + 3) & ~size_t(3)) + scr) * sizeof(float);
    c.ylong = 0;
    if (shmem > 160 * 1024 - 256) {                  // long recording: trace and output staging in global memory, LDS = the scratch alone
        c.ylong = 1;
        shmem = 4 * (size_t)c.nfft * sizeof(float);
        if (shmem > 160 * 1024 - 256) {              // nfft = 16384 (36868 < T <= 73728): re | im alone in LDS (128 KB), the twiddle / window tables in global memory
            c.ylong = 2;
            shmem = (c.split ? 1 : 2) * (size_t)c.nfft * sizeof(float);
        }
        if (shmem > 160 * 1024 - 256) return fail(CNMFE_EUNSUPPORTED, "trace of %lld frames: the Welch transform (nfft = %d) does not fit the deconvolution kernel's LDS (T <= %d)", (long long)T, c.nfft, WELCH_TMAX);
    }
    return 0;
}

int deconv_launch(cnmfe_ctx *ctx, DeconvCfg &c, size_t shmem, DeconvIO io, const int *d_list, int n, DeconvScratch &s) {
    if (n <= 0) return 0;
    const int64_t T = c.T;
    RET(s.pv.ensure((size_t)n * T * 8)); RET(s.pw.ensure((size_t)n * T * 8));
    RET(s.pt.ensure((size_t)n * T * 4)); RET(s.pl.ensure((size_t)n * T * 4));
    RET(s.tkp.ensure((size_t)n * 2 * T * 4)); RET(s.tko.ensure((size_t)n * 2 * T * 4)); RET(s.tkl.ensure((size_t)n * 2 * T * 4));
    RET(s.tkv.ensure((size_t)n * 4 * T * 8)); RET(s.pnum.ensure((size_t)n * T * 8));
    io.list = d_list;
    io.pv = s.pv.as<double>(); io.pw = s.pw.as<double>(); io.pt = s.pt.as<int>(); io.pl = s.pl.as<int>();
    io.tk_pool = s.tkp.as<int>(); io.tk_off = s.tko.as<int>(); io.tk_len = s.tkl.as<int>(); io.tk_val = s.tkv.as<double>(); io.pnum = s.pnum.as<double>();
    io.ybuf = io.obuf = nullptr;
    if (c.ylong) {
        const size_t Tal = ((size_t)T + 3) & ~size_t(3);
        RET(s.ybuf.ensure((size_t)n * Tal * 4)); RET(s.obuf.ensure((size_t)n * Tal * 4));
        io.ybuf = s.ybuf.as<float>(); io.obuf = s.obuf.as<float>();
        if (c.ylong == 2) { RET(s.tbuf.ensure((size_t)n * 3 * c.nfft * 4)); io.tbuf = s.tbuf.as<float>(); }
        if (c.split) {
            if (shmem > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_deconv<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            LAUNCH(ctx, c.hals ? "temporal_hals_deconv_level" : "deconv_temporal", (k_deconv<true, true>), dim3(n), dim3(DECONV_NT), shmem, c, io);
            return 0;
        }
        if (shmem > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_deconv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        LAUNCH(ctx, c.hals ? "temporal_hals_deconv_level" : "deconv_temporal", k_deconv<true>, dim3(n), dim3(DECONV_NT), shmem, c, io);
        return 0;
    }
    if (shmem > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_deconv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    LAUNCH(ctx, c.hals ? "temporal_hals_deconv_level" : "deconv_temporal", k_deconv<false>, dim3(n), dim3(DECONV_NT), shmem, c, io);
    return 0;
}

int temporal_deconv_sweeps(cnmfe_ctx *ctx, const cnmfe_deconv_opts *dopts, int64_t T, int K, int maxIter, const std::vector<std::vector<int>> &levels,
                           const int *dLvl, const std::vector<int> &off, float *dC, float *dCraw, float *dS, int64_t ldc, const float *dU,
                           const int *dNptr, const int *dNidx, const float *dNval, const float *dAa, float *dPars, float *dSn) {
    DeconvCfg c; size_t shmem;
    RET(deconv_setup(dopts, T, 1, c, shmem));
    DeconvScratch &scr = ctx->dscr; DevBuf &dB = ctx->scr[20];
    RET(dB.ensure((size_t)K * sizeof(float)));
    DeconvIO io;
    io.C = dC; io.Craw = dCraw; io.S = dS; io.ldc = ldc; io.U = dU; io.nptr = dNptr; io.nidx = dNidx; io.nval = dNval; io.aa = dAa;
    io.pars = dPars; io.sn_out = dSn; io.b_out = dB.as<float>();
    // maxIter < 0: `levels` are the levels of the dependency graph over the items of ALL sweeps (factor.hip, dag_schedule): one pass, the last sweep's items carry bit 30
    for (int it = 0; it < (maxIter < 0 ? 1 : maxIter); ++it) {
        c.last = maxIter >= 0 && it == maxIter - 1;
        for (size_t l = 0; l < levels.size(); ++l)
            RET(deconv_launch(ctx, c, shmem, io, dLvl + off[l], (int)levels[l].size(), scr));
    }
    return 0;                                                // (the caller waits when it takes results back to the host)
}

// the in-sweep deconvolution of several temporal jobs: one launch per level over all of them
int temporal_deconv_sweeps_jobs(cnmfe_ctx *ctx, const cnmfe_deconv_opts *dopts, int64_t T, int maxIter, const std::vector<TemporalJob *> &jobs,
                                const int2 *dList, const std::vector<int> &off, DevBuf &dTab) {
    DeconvCfg c; size_t shmem;
    RET(deconv_setup(dopts, T, 1, c, shmem));
    std::vector<DeconvJobDev> tab(jobs.size());
    for (size_t ji = 0; ji < jobs.size(); ++ji) {
        TemporalJob *j = jobs[ji];
        tab[ji] = DeconvJobDev{j->dC.as<float>(), j->dCraw.as<float>(), j->dS.as<float>(), j->ldc, j->dU.as<float>(), j->dNptr.as<int>(), j->dNidx.as<int>(),
                               j->dNval.as<float>(), j->dAa.as<float>(), j->dPars.as<float>(), j->dSn.as<float>(), j->dB.as<float>()};
    }
    RET(to_dev(ctx, dTab, tab.data(), tab.size()));
    DeconvIO io;
    io.C = io.Craw = io.S = nullptr; io.ldc = 0; io.U = nullptr; io.nptr = io.nidx = nullptr; io.nval = io.aa = nullptr; io.pars = io.sn_out = io.b_out = nullptr;
    io.jobs = dTab.as<DeconvJobDev>();
    const size_t nlev = off.size() - 1;
    for (int it = 0; it < (maxIter < 0 ? 1 : maxIter); ++it) {          // (maxIter < 0: the graph's levels over all sweeps, as temporal_deconv_sweeps)
        c.last = maxIter >= 0 && it == maxIter - 1;
        for (size_t l = 0; l < nlev; ++l) {
            io.jlist = dList + off[l];
            RET(deconv_launch(ctx, c, shmem, io, nullptr, off[l + 1] - off[l], ctx->dscr));
        }
    }
    return 0;
}

int deconv_all_run(cnmfe_ctx *ctx, int32_t K, int64_t T, float *C_raw, int c_order, const cnmfe_deconv_opts *opts,
                   float *C_out, float *S_out, float *pars_out, float *sn_out) {
    DeconvCfg c; size_t shmem;
    RET(deconv_setup(opts, T, 0, c, shmem));
    c.trace = (int)ctx->opt("deconv_trace", 0);
    DevBuf *S_ = ctx->scr;
    DevBuf &dCraw = S_[0], &dC = S_[1], &dS = S_[2], &dPars = S_[3], &dSn = S_[4], &dB = S_[20], &dList = S_[5]; DeconvScratch &scr = ctx->dscr;
    int64_t ldc;
    RET(upload_traces(ctx, dCraw, C_raw, K, T, c_order, &ldc));
    RET(dC.ensure((size_t)K * ldc * sizeof(float))); RET(dS.ensure((size_t)K * ldc * sizeof(float)));
    CK(hipMemsetAsync(dC.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
    CK(hipMemsetAsync(dS.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
    RET(dPars.ensure((size_t)K * sizeof(float))); RET(dSn.ensure((size_t)K * sizeof(float))); RET(dB.ensure((size_t)K * sizeof(float)));
    CK(hipMemsetAsync(dPars.p, 0, (size_t)K * sizeof(float), ctx->st()));         // fresh time-constant estimate for every trace
    CK(hipMemsetAsync(dSn.p, 0, (size_t)K * sizeof(float), ctx->st()));
    std::vector<int> list(K);
    for (int k = 0; k < K; ++k) list[k] = k;
    RET(to_dev(ctx, dList, list.data(), list.size()));
    DeconvIO io;
    io.C = dC.as<float>(); io.Craw = dCraw.as<float>(); io.S = dS.as<float>(); io.ldc = ldc;
    io.U = nullptr; io.nptr = nullptr; io.nidx = nullptr; io.nval = nullptr; io.aa = nullptr;
    io.pars = dPars.as<float>(); io.sn_out = dSn.as<float>(); io.b_out = dB.as<float>();
    // batches bound the pool/task scratch (7 arrays of T per trace)
    const int batch = 512;
    for (int k0 = 0; k0 < K; k0 += batch)
        RET(deconv_launch(ctx, c, shmem, io, dList.as<int>() + k0, std::min(batch, K - k0), scr));
    RET(download_traces(ctx, dC.as<float>(), ldc, C_out, K, T, c_order));
    RET(download_traces(ctx, dCraw.as<float>(), ldc, C_raw, K, T, c_order));
    RET(download_traces(ctx, dS.as<float>(), ldc, S_out, K, T, c_order));
    if (pars_out) CK(hipMemcpyAsync(pars_out, dPars.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    if (sn_out) CK(hipMemcpyAsync(sn_out, dSn.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    return 0;
}

// deconvTemporal on the bound matrix (cnmfe_deconv_temporal_bound): C_raw read and rewritten (C_raw - b) where it lies, C into a buffer of the context that is
// then SWAPPED with `bound`; the host copies go out on the copy stream behind an event (as cnmfe_stitch_finish_async)
int deconv_bound_run(cnmfe_ctx *ctx, const cnmfe_deconv_opts *opts, float *C_out, float *C_raw_out, float *S_out, float *pars_out, float *sn_out) {
    if (!ctx->bound_valid || ctx->bound_K <= 0) return fail(CNMFE_ESTATE, "no bound trace matrix (cnmfe_traces_bind / cnmfe_stitch_finish)");
    const int32_t K = ctx->bound_K; const int64_t T = ctx->bound_T, ldc = (T + 3) & ~int64_t(3);
    DeconvCfg c; size_t shmem;
    RET(deconv_setup(opts, T, 0, c, shmem));
    c.trace = (int)ctx->opt("deconv_trace", 0);
    if (ctx->copy_pending) { CK(hipStreamWaitEvent(ctx->st(), ctx->ev_copy_done, 0)); ctx->copy_pending = false; }   // the last downloads still read bound / dcv_*
    DevBuf &dC = ctx->dcv_c, &dS = ctx->dcv_s, &dPars = ctx->dcv_pars, &dSn = ctx->dcv_sn, &dB = ctx->scr[20], &dList = ctx->scr[5];
    RET(dC.ensure((size_t)K * ldc * sizeof(float))); RET(dS.ensure((size_t)K * ldc * sizeof(float)));
    CK(hipMemsetAsync(dC.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
    CK(hipMemsetAsync(dS.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
    RET(dPars.ensure((size_t)K * sizeof(float))); RET(dSn.ensure((size_t)K * sizeof(float))); RET(dB.ensure((size_t)K * sizeof(float)));
    CK(hipMemsetAsync(dPars.p, 0, (size_t)K * sizeof(float), ctx->st()));         // fresh time-constant estimate for every trace
    CK(hipMemsetAsync(dSn.p, 0, (size_t)K * sizeof(float), ctx->st()));
    std::vector<int> list(K);
    for (int k = 0; k < K; ++k) list[k] = k;
    RET(to_dev(ctx, dList, list.data(), list.size()));
    DeconvIO io;
    io.C = dC.as<float>(); io.Craw = ctx->bound.as<float>(); io.S = dS.as<float>(); io.ldc = ldc;
    io.U = nullptr; io.nptr = nullptr; io.nidx = nullptr; io.nval = nullptr; io.aa = nullptr;
    io.pars = dPars.as<float>(); io.sn_out = dSn.as<float>(); io.b_out = dB.as<float>();
    const int batch = 512;
    for (int k0 = 0; k0 < K; k0 += batch)
        RET(deconv_launch(ctx, c, shmem, io, dList.as<int>() + k0, std::min(batch, K - k0), ctx->dscr));
    ctx->bound.swap(dC); ++ctx->bound_gen;                  // bound = C (row-major), dcv_c = C_raw - b
    ctx->bound_order = CNMFE_ROWMAJOR;
    if (C_out || C_raw_out || S_out || pars_out || sn_out) {
        if (!ctx->copy_stream) {
            CK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            CK(hipEventCreateWithFlags(&ctx->ev_bound_ready, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ctx->ev_copy_done, hipEventDisableTiming));
        }
        CK(hipEventRecord(ctx->ev_bound_ready, ctx->st()));
        CK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_bound_ready, 0));
        const size_t rb = (size_t)T * sizeof(float), pb = (size_t)ldc * sizeof(float);
        if (C_out) CK(hipMemcpy2DAsync(C_out, rb, ctx->bound.p, pb, rb, K, hipMemcpyDeviceToHost, ctx->copy_stream));
        if (C_raw_out) CK(hipMemcpy2DAsync(C_raw_out, rb, dC.p, pb, rb, K, hipMemcpyDeviceToHost, ctx->copy_stream));
        if (S_out) CK(hipMemcpy2DAsync(S_out, rb, dS.p, pb, rb, K, hipMemcpyDeviceToHost, ctx->copy_stream));
        if (pars_out) CK(hipMemcpyAsync(pars_out, dPars.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, ctx->copy_stream));
        if (sn_out) CK(hipMemcpyAsync(sn_out, dSn.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, ctx->copy_stream));
        CK(hipEventRecord(ctx->ev_copy_done, ctx->copy_stream));
        ctx->copy_pending = true;
        RET(ctx->copy_batch_mark());
    }
    return 0;
}

// loads this translation unit's code object now (HIP loads it at the first launch of one of its kernels -- milliseconds each that would otherwise fall into the first iteration): cnmfe_create
int tu_warm_deconv() { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)k_sn_pixels) == hipSuccess ? 0 : -1; }

}  // namespace cnmfe
