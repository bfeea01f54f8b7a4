// C-ABI surface + context / data-plane / ring-pattern plumbing.   (gfx950 only)
#include "common.hpp"
#include <mutex>
#include <hip/hip_fp16.h>
#include <math.h>
#include <dlfcn.h>

namespace cnmfe {
thread_local char g_err[1024] = "";

// ---------------------------------------------------------------------------------
// dtype conversion on upload
// ---------------------------------------------------------------------------------
template <class S> __device__ inline float to_f32(S v) { return (float)v; }
template <> __device__ inline float to_f32<__half>(__half v) { return __half2float(v); }

template <class S>
__global__ void k_convert(const S *__restrict__ src, float *__restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = to_f32<S>(src[i]);
}

// temporal mean of every block pixel, double accumulation (fit_ring_model.m:42, P.Ymean)
__global__ void k_ymean(const float *__restrict__ Y, int64_t d_b, int64_t T, double *__restrict__ ym, float *__restrict__ ymf) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= d_b) return;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t t = 0;
    for (; t + 3 < T; t += 4) {
        s0 += Y[(t + 0) * d_b + q]; s1 += Y[(t + 1) * d_b + q];
        s2 += Y[(t + 2) * d_b + q]; s3 += Y[(t + 3) * d_b + q];
    }
    for (; t < T; ++t) s0 += Y[t * d_b + q];
    double m = ((s0 + s1) + (s2 + s3)) / (double)T;
    ym[q] = m; ymf[q] = (float)m;
}

// resident layout: Yc4[c][q] = (Y[4c..4c+3][q] - Ymean[q]) as one float4 (frames past T are 0)
__global__ void k_center4(const float *__restrict__ Y, const float *__restrict__ ymf, int64_t d_b, int64_t T, float4 *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= d_b) return;
    const int64_t c = blockIdx.y;
    const float m = ymf[q];
    const int64_t t = 4 * c;
    float4 v;
    v.x = Y[t * d_b + q] - m;
    v.y = t + 1 < T ? Y[(t + 1) * d_b + q] - m : 0.f;
    v.z = t + 2 < T ? Y[(t + 2) * d_b + q] - m : 0.f;
    v.w = t + 3 < T ? Y[(t + 3) * d_b + q] - m : 0.f;
    out[c * d_b + q] = v;
}

// 576 bytes of private (scratch) memory per lane, touched: launched once when a context is created.  The runtime sets a queue's scratch arena up at the
// first launch of a kernel that needs one -- the ring solve spills 76 bytes per lane -- and on a fresh box that first set-up took 0.5-0.9 s (profiles/r04/
// cold_start.txt: the stall sat on the first bg_ring_solve of the first process or two of every lease, on the first kernel with scratch in round 3).
__global__ void k_scratch_warm(int *out, int n) {
    volatile int a[144];
    for (int i = 0; i < 144; ++i) a[i] = i * n;
    int s = 0;
    for (int i = 0; i < n; ++i) s += a[(i * 7 + n) % 144];
    if (out) *out = s;
}

// W = 1/count on the in-FOV ring neighbours (initComponents_parallel.m:229-233)
__global__ void k_ring_init(float *__restrict__ W, int64_t d, int nr, int p, const int *__restrict__ dr, const int *__restrict__ dc,
                            int r0, int c0, int d1, int d2) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= d) return;
    int r = r0 + (int)(m % nr), c = c0 + (int)(m / nr);     // absolute 1-based
    int cnt = 0;
    for (int i = 0; i < p; ++i) { int rr = r + dr[i], cc = c + dc[i]; cnt += (rr >= 1 && rr <= d1 && cc >= 1 && cc <= d2); }
    float v = cnt ? 1.0f / (float)cnt : 0.0f;
    for (int i = 0; i < p; ++i) { int rr = r + dr[i], cc = c + dc[i];
        W[(int64_t)i * d + m] = (rr >= 1 && rr <= d1 && cc >= 1 && cc <= d2) ? v : 0.0f; }
}

// K x T column-major (k fastest) -> [k][ldc] row-major
__global__ void k_transpose_in(const float *__restrict__ src, float *__restrict__ dst, int K, int64_t T, int64_t ldc) {
    __shared__ float tile[32][33];
    int64_t t0 = (int64_t)blockIdx.x * 32; int k0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int64_t t = t0 + j; int k = k0 + threadIdx.x;
        tile[j][threadIdx.x] = (t < T && k < K) ? src[t * K + k] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int k = k0 + j; int64_t t = t0 + threadIdx.x;
        if (k < K && t < T) dst[(int64_t)k * ldc + t] = tile[threadIdx.x][j];
    }
}
__global__ void k_transpose_out(const float *__restrict__ src, int64_t ldc, float *__restrict__ dst, int K, int64_t T) {
    __shared__ float tile[32][33];
    int64_t t0 = (int64_t)blockIdx.x * 32; int k0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int k = k0 + j; int64_t t = t0 + threadIdx.x;
        tile[j][threadIdx.x] = (k < K && t < T) ? src[(int64_t)k * ldc + t] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int64_t t = t0 + j; int k = k0 + threadIdx.x;
        if (t < T && k < K) dst[t * K + k] = tile[threadIdx.x][j];
    }
}

__global__ void __launch_bounds__(256) k_gather_rows(const float4 *__restrict__ src, int64_t ldc4, const int *__restrict__ idx, float4 *__restrict__ dst) {
    const float4 *s = src + (int64_t)idx[blockIdx.x] * ldc4;
    float4 *d = dst + (int64_t)blockIdx.x * ldc4;
    for (int64_t c = threadIdx.x; c < ldc4; c += 256) d[c] = s[c];
}

int upload_traces(cnmfe_ctx *ctx, DevBuf &dst, const float *C, int32_t K, int64_t T, int order, int64_t *ldc_out) {
    int64_t ldc = (T + 3) & ~int64_t(3);
    *ldc_out = ldc;
    RET(dst.ensure(std::max<int64_t>(1, (int64_t)K * ldc) * sizeof(float)));
    if (K == 0) return 0;
    if (order == CNMFE_BOUND) {                            // the matrix bound with cnmfe_traces_bind: a device copy, no PCIe transfer
        if (!ctx->bound_valid || ctx->bound_K != K || ctx->bound_T != T)
            return fail(CNMFE_ESTATE, "no bound trace matrix of %d x %lld (cnmfe_traces_bind)", K, (long long)T);
        CK(hipMemcpyAsync(dst.p, ctx->bound.p, (size_t)K * ldc * sizeof(float), hipMemcpyDeviceToDevice, ctx->st()));
        return 0;
    }
    if (!C) return fail(CNMFE_EINVAL, "null trace matrix");
    if (order == CNMFE_BOUND_ROWS) {                       // K rows of the bound matrix, picked on the device
        if (!ctx->bound_valid || ctx->bound_T != T) return fail(CNMFE_ESTATE, "no bound trace matrix with %lld frames (cnmfe_traces_bind)", (long long)T);
        const int32_t *rows = reinterpret_cast<const int32_t *>(C);
        for (int32_t k = 0; k < K; ++k)
            if (rows[k] < 0 || rows[k] >= ctx->bound_K) return fail(CNMFE_EINVAL, "row %d of the bound trace matrix (%d rows) does not exist", rows[k], ctx->bound_K);
        DevBuf &dIdx = ctx->tmp[15];
        RET(to_dev(ctx, dIdx, rows, (size_t)K));
        LAUNCH(ctx, "gather_rows", k_gather_rows, dim3((unsigned)K), dim3(256), 0, ctx->bound.as<float4>(), ldc >> 2, dIdx.as<int>(), dst.as<float4>());
        return 0;                                          // (the index array went through the pinned arena: the caller may drop it)
    }
    CK(hipMemsetAsync(dst.p, 0, (size_t)K * ldc * sizeof(float), ctx->st()));
    if (order == CNMFE_ROWMAJOR) {
        CK(hipMemcpy2DAsync(dst.p, ldc * sizeof(float), C, T * sizeof(float), T * sizeof(float), K, hipMemcpyDefault, ctx->st()));
    } else {
        RET(ctx->stage.ensure((size_t)K * T * sizeof(float)));
        CK(hipMemcpyAsync(ctx->stage.p, C, (size_t)K * T * sizeof(float), hipMemcpyHostToDevice, ctx->st()));
        dim3 g((unsigned)((T + 31) / 32), (unsigned)((K + 31) / 32)), b(32, 8);
        LAUNCH(ctx, "transpose_in", k_transpose_in, g, b, 0, ctx->stage.as<float>(), dst.as<float>(), K, T, ldc);
    }
    CK(hipStreamSynchronize(ctx->st()));                 // the caller's matrix was read in place: it is free again when this returns
    return 0;
}

int download_traces(cnmfe_ctx *ctx, const float *dC, int64_t ldc, float *C, int32_t K, int64_t T, int order) {
    if (K == 0 || !C) return 0;
    if (order == CNMFE_BOUND || order == CNMFE_BOUND_ROWS) order = ctx->bound_order;   // outputs of a call on the bound matrix come back in ITS layout
    if (order == CNMFE_ROWMAJOR) {
        CK(hipMemcpy2DAsync(C, T * sizeof(float), dC, ldc * sizeof(float), T * sizeof(float), K, hipMemcpyDeviceToHost, ctx->st()));
    } else {
        RET(ctx->stage.ensure((size_t)K * T * sizeof(float)));
        dim3 g((unsigned)((T + 31) / 32), (unsigned)((K + 31) / 32)), b(32, 8);
        LAUNCH(ctx, "transpose_out", k_transpose_out, g, b, 0, dC, ldc, ctx->stage.as<float>(), K, T);
        CK(hipMemcpyAsync(C, ctx->stage.p, (size_t)K * T * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    }
    CK(hipStreamSynchronize(ctx->st()));
    return 0;
}

// copy of a pinned-arena slot to device memory by a kernel (to_dev, common.hpp): both ends are 256-byte granular (PinArena::take, DevBuf::ensure)
// (blockIdx.y = segment: the held-back uploads of cnmfe_ctx::st go out as one dispatch)
__global__ void __launch_bounds__(256) k_pin_copy(PinSegs s) {
    const int g = blockIdx.y;
    const uint4 *__restrict__ src = s.src[g]; uint4 *__restrict__ dst = s.dst[g];
    const unsigned n16 = s.n16[g];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
// every live context, so that a buffer about to be freed can have ALL held-back uploads sent first whoever holds them (DevBuf::ensure / ~DevBuf); the
// mutex also serialises the segment lists against a second host thread driving the same context
static std::mutex g_pin_mu;
static std::vector<cnmfe_ctx *> g_pin_live;
void pin_register(cnmfe_ctx *ctx, bool live) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = std::find(g_pin_live.begin(), g_pin_live.end(), ctx);
    if (live && it == g_pin_live.end()) g_pin_live.push_back(ctx);
    if (!live && it != g_pin_live.end()) g_pin_live.erase(it);
}
static void flush_locked(cnmfe_ctx *ctx) {
    if (!ctx->npseg) return;
    PinSegs &pseg = ctx->pseg;
    unsigned mx = 0;
    for (int i = 0; i < ctx->npseg; ++i) mx = std::max(mx, pseg.n16[i]);
    for (int i = ctx->npseg; i < PIN_NSEG; ++i) { pseg.src[i] = nullptr; pseg.dst[i] = nullptr; pseg.n16[i] = 0; }
    const unsigned nb = std::min<unsigned>((mx + 255) / 256, 256);
    const int n = ctx->npseg; ctx->npseg = 0;
    hipLaunchKernelGGL(k_pin_copy, dim3(nb, (unsigned)n), dim3(256), 0, ctx->stream_, pseg);      // (a failed launch is reported by the next LAUNCH's hipGetLastError)
}
// Before device memory [p, p + bytes) is freed: the held-back uploads INTO it must go out first.  Only the context that owns the allocation ever uploads into it,
// and a context is driven by one thread -- so this flushes the caller's own context and nobody else's (flushing every context from here let one thread reset
// another thread's npseg, which cnmfe_ctx::st() reads without the lock: ADVICE round 3).
void pin_flush_range(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    const char *lo = (const char *)p, *hi = lo + bytes;
    int cur = -1;
    for (cnmfe_ctx *c : g_pin_live) {
        bool hit = false;
        for (int i = 0; i < c->npseg && !hit; ++i) { const char *d_ = (const char *)c->pseg.dst[i]; hit = d_ >= lo && d_ < hi; }
        if (!hit) continue;
        if (cur < 0) (void)hipGetDevice(&cur);
        if (c->device != cur) (void)hipSetDevice(c->device);
        flush_locked(c);
        if (c->device != cur) (void)hipSetDevice(cur);
    }
}
void pin_flush_all() {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    int cur = -1;
    for (cnmfe_ctx *c : g_pin_live) {
        if (!c->npseg) continue;
        if (cur < 0) (void)hipGetDevice(&cur);
        if (c->device != cur) (void)hipSetDevice(c->device);         // a launch goes to the current device's streams only
        flush_locked(c);
        if (c->device != cur) (void)hipSetDevice(cur);
    }
}
int pinned_to_dev(cnmfe_ctx *ctx, void *dst, const void *src_pinned, size_t bytes) {
    if (bytes > (size_t(8) << 20)) { CK(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, ctx->st())); return 0; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (ctx->npseg == PIN_NSEG) flush_locked(ctx);
    const int i = ctx->npseg++;
    ctx->pseg.src[i] = (const uint4 *)src_pinned; ctx->pseg.dst[i] = (uint4 *)dst; ctx->pseg.n16[i] = (unsigned)((bytes + 15) / 16);
    return 0;
}
}  // namespace cnmfe
int cnmfe_ctx::copy_batch_mark() {
    hipEvent_t e;
    if (!copy_ev_pool.empty()) { e = copy_ev_pool.back(); copy_ev_pool.pop_back(); }
    else CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, copy_stream));
    copy_gens.push_back({++copy_gen, e});
    return 0;
}
void cnmfe_ctx::flush_copies() {
    std::lock_guard<std::mutex> lk(cnmfe::g_pin_mu);
    cnmfe::flush_locked(this);
}
namespace cnmfe {

bool bound_rows_of(const cnmfe_ctx *ctx, const float *C, int c_order, int32_t K, std::vector<int32_t> &rows) {
    rows.clear();
    if (!ctx->bound_valid || K <= 0) return false;
    if (c_order == CNMFE_BOUND) {
        if (K != ctx->bound_K) return false;
        rows.resize((size_t)K);
        for (int32_t k = 0; k < K; ++k) rows[k] = k;
        return true;
    }
    if (c_order == CNMFE_BOUND_ROWS && C) {
        const int32_t *r = reinterpret_cast<const int32_t *>(C);
        for (int32_t k = 0; k < K; ++k) if (r[k] < 0 || r[k] >= ctx->bound_K) return false;
        rows.assign(r, r + K);
        return true;
    }
    return false;
}

int ensure_ymean(cnmfe_ctx *ctx, Patch *P) {
    if (P->ymean_valid) return 0;
    if (P->frames_uploaded < P->T) return fail(CNMFE_ESTATE, "block has %lld of %lld frames uploaded", (long long)P->frames_uploaded, (long long)P->T);
    RET(P->ymean_d.ensure(P->d_b * sizeof(double)));
    RET(P->ymean_f.ensure(P->d_b * sizeof(float)));
    LAUNCH(ctx, "ymean", k_ymean, dim3((unsigned)((P->d_b + 255) / 256)), dim3(256), 0,
           P->Y.as<float>(), P->d_b, P->T, P->ymean_d.as<double>(), P->ymean_f.as<float>());
    // build the resident centred / 4-frame-interleaved copy, then drop the upload staging
    P->Tc = (P->T + 3) / 4;
    RET(P->Yc4.ensure((size_t)P->Tc * P->d_b * sizeof(float4)));
    LAUNCH(ctx, "center4", k_center4, dim3((unsigned)((P->d_b + 255) / 256), (unsigned)P->Tc), dim3(256), 0,
           P->Y.as<float>(), P->ymean_f.as<float>(), P->d_b, P->T, P->Yc4.as<float4>());
    CK(hipStreamSynchronize(ctx->st()));
    (void)hipFree(P->Y.p); P->Y.p = nullptr; P->Y.cap = 0;
    P->ymean_valid = true;
    return 0;
}

// pmax = max_i #{j : W(i,j) > 0}  (fit_ring_model.m:60)
__global__ void k_ring_pmax(const float *__restrict__ W, int64_t d, int p, int *__restrict__ pmax) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    if (m < d) {                                              // (16 independent loads in flight per thread: one load per trip made the kernel 96 memory latencies long)
        int i = 0;
        for (; i + 16 <= p; i += 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = W[(int64_t)(i + u) * d + m];
#pragma unroll
            for (int u = 0; u < 16; ++u) c += w[u] > 0.f;
        }
        for (; i < p; ++i) c += W[(int64_t)i * d + m] > 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(c, o); c = v > c ? v : c; }
    if ((threadIdx.x & 63) == 0) atomicMax(pmax, c);
}
static bool first_run_of_row(const Patch *P, const float *row0) {
    std::vector<float> u;
    const int r = P->prect[0], c = P->prect[2];
    int nvalid = 0;
    for (int i = 0; i < P->p; ++i) {
        const int rr = r + P->dr[i], cc = c + P->dc[i];
        if (rr < 1 || rr > P->d1 || cc < 1 || cc > P->d2) continue;
        ++nvalid; u.push_back(row0[i]);
    }
    if (nvalid < P->d_b) u.push_back(0.f);
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    return u.size() == 2;
}
// pinned layout: int pmax | float row1[p]
int ring_stats_enqueue(cnmfe_ctx *ctx, Patch *P) {
    const size_t bytes = 64 + (size_t)P->p * sizeof(float);
    if (!P->stat_host) CK(hipHostMalloc(&P->stat_host, bytes, hipHostMallocDefault));
    if (!P->stat_ev) CK(hipEventCreateWithFlags(&P->stat_ev, hipEventDisableTiming));
    RET(P->stat_dev.ensure(64));
    CK(hipMemsetAsync(P->stat_dev.p, 0, 64, ctx->st()));
    LAUNCH(ctx, "ring_pmax", k_ring_pmax, dim3((unsigned)((P->d + 255) / 256)), dim3(256), 0, P->W.as<float>(), P->d, P->p, P->stat_dev.as<int>());
    CK(hipMemcpyAsync(P->stat_host, P->stat_dev.p, sizeof(int), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipMemcpy2DAsync((char *)P->stat_host + 64, sizeof(float), P->W.p, P->d * sizeof(float), sizeof(float), P->p, hipMemcpyDeviceToHost, ctx->st()));
    CK(hipEventRecord(P->stat_ev, ctx->st()));
    P->stat_valid = true;
    return 0;
}
int ring_stats_get(cnmfe_ctx *ctx, Patch *P, int *pmax, bool *first) {
    if (!P->stat_valid) RET(ring_stats_enqueue(ctx, P));      // W came from somewhere else (ring_init, cnmfe_ring_set_csr): evaluate now
    CK(hipEventSynchronize(P->stat_ev));
    if (pmax) *pmax = *reinterpret_cast<const int *>(P->stat_host);
    if (first) *first = first_run_of_row(P, reinterpret_cast<const float *>((const char *)P->stat_host + 64));
    return 0;
}

// length(unique(W_old(1,:)))==2 on row 1 of the resident W, implicit zeros of the sparse row included
int ring_first_run(cnmfe_ctx *ctx, Patch *P, bool *first) { return ring_stats_get(ctx, P, nullptr, first); }
int ctx_errflag(cnmfe_ctx *ctx, int **dflag) {
    if (!ctx->errflag.p) {
        RET(ctx->errflag.ensure(sizeof(int)));
        CK(hipMemsetAsync(ctx->errflag.p, 0, sizeof(int), ctx->st()));
    }
    *dflag = ctx->errflag.as<int>();
    return 0;
}
// what a raised error word means: clears it, drops every table a later call would trust, and reports
// `taken`: the device word was already read AND cleared at the point of the stream the flag `h` stands for (a ticket's k_flag_take) -- clearing it again here
// would wipe what kernels queued behind that point have raised since
static int errflag_raise(cnmfe_ctx *ctx, int h, bool taken = false) {
    if (!taken) CK(hipMemsetAsync(ctx->errflag.p, 0, sizeof(int), ctx->st()));
    // whatever raised the flag left truncated or inconsistent tables behind (the footprint terms beside a residual, a P table): nothing kept with the patches may
    // be trusted by a later call -- the next residual sweeps again, the next spatial update builds its own table
    for (auto &kv : ctx->patches) { Patch *q = kv.second; q->ysig_valid = false; q->ysig_virtual = false; q->pend = false; q->res_ac = false; q->res_kind = 0; q->pt_valid = false; }
    ctx->bgs_patch = -1;
    if (h & 2) return fail(CNMFE_EUNSUPPORTED, "a pixel's ring touches more than 256 footprints of A_prev (flag %d)", h);
    if (h & 4) return fail(CNMFE_EUNSUPPORTED, "bg_ssub > 1: a pixel's interpolation window meets more than 64 footprints of A_prev (flag %d)", h);
    return fail(CNMFE_ESTATE, "a kernel met an inconsistent table (flag %d): the ring regression needed a block pair the covariance table does not hold", h);
}
// waits for the stream and reports what its kernels raised since the last call
int ctx_check_errflag(cnmfe_ctx *ctx) {
    if (!ctx->errflag.p) { CK(hipStreamSynchronize(ctx->st())); return 0; }
    int h = 0;
    CK(hipMemcpyAsync(&h, ctx->errflag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    if (!h) return 0;
    return errflag_raise(ctx, h);
}
// a ticket's share of the error word: what the kernels in front of it raised since the last take, read and cleared in ONE step at the ticket's place in the stream.
// (A plain copy at record time + a clear at wait time reported one fault once per outstanding ticket -- every raise invalidating state the caller had rebuilt in
//  between -- and the clear wiped faults raised behind the ticket.)
__global__ void k_flag_take(int *__restrict__ flag, int *__restrict__ out) { *out = atomicExch(flag, 0); }
// a ticket's download as ONE kernel: n16 x 16 bytes from device memory into PINNED host memory (mapped into the device's address space like the upload arena that
// k_pin_copy reads), and the error word taken into the ticket's pinned slot.  Rounds 4-6 (first half) queued hipMemcpyAsync(device -> pinned) + k_flag_take + a
// 4-byte hipMemcpyAsync: in the second to fourth iteration after an upload those copy calls BLOCKED the host for 0.4-0.7 ms each (one rank of eight at c4: an
// iteration of 10 ms instead of 3.3; sixteen patches: 30 instead of 18 -- scripts/gpu/r6_call28.sh), and they were three dispatches per patch
__global__ void __launch_bounds__(256) k_fetch_take(const uint4 *__restrict__ src, uint4 *__restrict__ dst, unsigned n16, int *__restrict__ flag, int *__restrict__ flag_out) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) *flag_out = atomicExch(flag, 0);
}

// ---- T5: stitch accumulator (update_temporal_parallel.m:264-280) ------------------------------------------------------------------
// acc[row][t] += aa_m(j) * C_raw_m(j, t), row = ind_m[j]; the weight sum lives in column ld - 4 of the same row
__global__ void __launch_bounds__(256) k_stitch_add(const float *__restrict__ craw, int64_t ldc, const float *__restrict__ aa, const int *__restrict__ ind,
                                                    float *__restrict__ acc, int64_t ld, int64_t T) {
    const int j = blockIdx.y;
    const float w = aa[j];
    float *row = acc + (int64_t)ind[j] * ld;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < T) row[t] += w * craw[(int64_t)j * ldc + t];
    if (t == 0) row[ld - 4] += w;
}
// one workgroup per row: C_raw(k,:) = acc(k,:) / max(aa, (aa == 0)) (:279-280), minus its minimum if asked (:285); written to the bound trace matrix
__global__ void __launch_bounds__(256) k_stitch_finish(const float *__restrict__ acc, int64_t ld, int64_t T, int subtract_min, float *__restrict__ out, int64_t ldc) {
    const int k = blockIdx.x;
    const float *row = acc + (int64_t)k * ld;
    float w = row[ld - 4];
    if (w == 0.f) w = 1.f;
    float mn = INFINITY;
    for (int64_t t = threadIdx.x; t < T; t += 256) mn = fminf(mn, row[t] / w);
    __shared__ float red[256];
    red[threadIdx.x] = mn; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    const float sub = subtract_min ? red[0] : 0.f;
    for (int64_t t = threadIdx.x; t < ldc; t += 256) out[(int64_t)k * ldc + t] = t < T ? row[t] / w - sub : 0.f;
}

// RCCL, resolved at first use (single-process multi-GPU stitch only): the library is optional for single-GPU hosts
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr; int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int load() {
        if (h) return 0;
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return fail(CNMFE_EUNSUPPORTED, "cnmfe_stitch_temporal over several GPUs needs RCCL (librccl.so not found: %s)", dlerror());
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll"); AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy"); GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!CommInitAll || !AllReduce || !GroupStart || !GroupEnd || !CommDestroy) { h = nullptr; return fail(CNMFE_EUNSUPPORTED, "librccl.so lacks the ncclCommInitAll / ncclAllReduce / ncclGroup* symbols"); }
        return 0;
    }
};
static Rccl g_rccl;
static int stitch_finish_one(cnmfe_ctx *ctx, int subtract_min, float *C_raw_out, int c_order, bool async_copy = false) {
    if (!ctx->stitch_open) return fail(CNMFE_ESTATE, "cnmfe_stitch_begin has not been called");
    const int32_t K = ctx->stitch_K; const int64_t T = ctx->stitch_T, ldc = (T + 3) & ~int64_t(3);
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    RET(ctx->bound.ensure((size_t)std::max<int64_t>(1, (int64_t)K * ldc) * sizeof(float)));
    if (ctx->copy_pending) { CK(hipStreamWaitEvent(ctx->st(), ctx->ev_copy_done, 0)); ctx->copy_pending = false; }   // the last download still reads `bound`
    if (K > 0) LAUNCH(ctx, "stitch_finish", k_stitch_finish, dim3((unsigned)K), dim3(256), 0, ctx->stitch.as<float>(), ctx->stitch_ld, T, subtract_min, ctx->bound.as<float>(), ldc);
    ctx->bound_K = K; ctx->bound_T = T; ctx->bound_order = (c_order == CNMFE_COLMAJOR) ? CNMFE_COLMAJOR : CNMFE_ROWMAJOR; ctx->bound_valid = K > 0;
    ++ctx->bound_gen;
    ctx->stitch_open = false;
    if (C_raw_out && async_copy && K > 0) {
        // the copy goes out on its own stream behind an event: the compute stream is free for the next call's kernels at once
        if (!ctx->copy_stream) {
            CK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            CK(hipEventCreateWithFlags(&ctx->ev_bound_ready, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ctx->ev_copy_done, hipEventDisableTiming));
        }
        CK(hipEventRecord(ctx->ev_bound_ready, ctx->st()));
        CK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_bound_ready, 0));
        CK(hipMemcpy2DAsync(C_raw_out, T * sizeof(float), ctx->bound.p, ldc * sizeof(float), T * sizeof(float), K, hipMemcpyDeviceToHost, ctx->copy_stream));
        CK(hipEventRecord(ctx->ev_copy_done, ctx->copy_stream));
        ctx->copy_pending = true;
        RET(ctx->copy_batch_mark());
        return 0;
    }
    if (C_raw_out) RET(download_traces(ctx, ctx->bound.as<float>(), ldc, C_raw_out, K, T, c_order == CNMFE_COLMAJOR ? CNMFE_COLMAJOR : CNMFE_ROWMAJOR));
    return 0;
}

// ring offsets: get_nhood.m:1-25, then sorted by (dc, dr) == MATLAB sparse column order
static void ring_offsets(int radius, int k, std::vector<int32_t> &dr, std::vector<int32_t> &dc) {
    dr.clear(); dc.clear();
    for (int c = -radius; c <= radius; ++c)
        for (int r = -radius; r <= radius; ++r) {
            int d2 = c * c + r * r;                         // R>=radius && R<radius+1 on integers (:10-11)
            if (d2 >= radius * radius && d2 < (radius + 1) * (radius + 1)) { dr.push_back(r); dc.push_back(c); }
        }
    int n = (int)dr.size();
    if (k <= 0 || k > n) return;                            // :17
    std::vector<int> ids(n);
    for (int i = 0; i < n; ++i) ids[i] = i;
    std::vector<double> ang(n);
    for (int i = 0; i < n; ++i) ang[i] = atan2((double)dr[i], (double)dc[i]);    // :20
    std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return ang[a] < ang[b]; });   // :21
    std::vector<std::pair<int, int>> sel;
    for (int j = 0; j < k; ++j) {
        double x = (k == 1) ? 1.0 : 1.0 + (double)(n - 1) * j / (double)(k - 1);               // linspace(1,n,k)
        int idx = (int)floor(x + 0.5) - 1;                                                        // round()
        sel.push_back({dc[ids[idx]], dr[ids[idx]]});
    }
    std::sort(sel.begin(), sel.end());
    sel.erase(std::unique(sel.begin(), sel.end()), sel.end());
    dr.clear(); dc.clear();
    for (auto &s : sel) { dc.push_back(s.first); dr.push_back(s.second); }
}
}  // namespace cnmfe

using namespace cnmfe;

cnmfe_ctx::~cnmfe_ctx() {
    if (rccl_comm && g_rccl.CommDestroy) { (void)hipSetDevice(device); g_rccl.CommDestroy(rccl_comm); rccl_comm = nullptr; }
    for (auto &kv : patches) delete kv.second;
    prof.drain();
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    if (ev_bound_ready) (void)hipEventDestroy(ev_bound_ready);
    if (ev_copy_done) (void)hipEventDestroy(ev_copy_done);
    for (auto &g : copy_gens) (void)hipEventDestroy(g.second);
    for (auto e : copy_ev_pool) (void)hipEventDestroy(e);
    cnmfe::pin_register(this, false);
    for (auto *j : tjobs) delete j;
    for (auto e : tickets) (void)hipEventDestroy(e);
    if (ticket_flags) (void)hipHostFree(ticket_flags);
    if (stream_) (void)hipStreamDestroy(stream_);
    for (auto *L : lanes) delete L;                           // (the inactive lanes' streams, events and scratch)
    if (ev_fork) (void)hipEventDestroy(ev_fork);
}

// ---- execution lanes (common.hpp, cnmfe_lane) ----
void cnmfe_ctx::swap_lane(cnmfe_lane &L) {
    std::swap(stream_, L.stream_); std::swap(pseg, L.pseg); std::swap(npseg, L.npseg); pin.swap(L.pin); std::swap(spatial_nnz, L.spatial_nnz);
    for (int i = 0; i < 32; ++i) vp[i].swap(L.vp[i]);          // (the high-water sizes hw_* stay with the context: what one lane's patches needed, the other lanes' buffers get at once)
    std::swap(last_ldc, L.last_ldc);
    ysig_low.swap(L.ysig_low); up_tmp.swap(L.up_tmp); bgs_r.swap(L.bgs_r); bgs_b.swap(L.bgs_b); bgs_upr.swap(L.bgs_upr); bgs_upc.swap(L.bgs_upc);
    std::swap(bgs_patch, L.bgs_patch); std::swap(bgs_d1s, L.bgs_d1s); std::swap(bgs_dF, L.bgs_dF);
    bf.swap(L.bf); dig_smax.swap(L.dig_smax); dig_rspart.swap(L.dig_rspart); dig_scale.swap(L.dig_scale); tdig.swap(L.tdig); tscale.swap(L.tscale); gk.swap(L.gk);
    win_items.swap(L.win_items); bf2.swap(L.bf2); outl_cnt.swap(L.outl_cnt); outl_sel.swap(L.outl_sel); cov.swap(L.cov); rowsum.swap(L.rowsum);
    for (int i = 0; i < 16; ++i) tmp[i].swap(L.tmp[i]);
    for (int i = 0; i < 7; ++i) inc[i].swap(L.inc[i]);
    for (int i = 0; i < 4; ++i) stg[i].swap(L.stg[i]);
    wcodes.swap(L.wcodes); solve_fill.swap(L.solve_fill); stage.swap(L.stage);
    for (int i = 0; i < 24; ++i) scr[i].swap(L.scr[i]);
    dscr.swap(L.dscr);
}
int cnmfe_ctx::activate(int lane) {
    if (lanes.empty() || lane == cur_lane) {
        if (!lanes.empty() && lane != 0 && lanes[lane]->fork_seen != fork_gen) {      // (active, but a call on lane 0 has queued shared work since this lane last looked)
            CK(hipStreamWaitEvent(stream_, ev_fork, 0)); lanes[lane]->fork_seen = fork_gen;
        }
        if (!lanes.empty() && lane != 0) lanes[lane]->dirty = true;
        return 0;
    }
    if (lane < 0 || lane >= (int)lanes.size()) return fail(CNMFE_ESTATE, "lane %d of %d", lane, (int)lanes.size());
    if (npseg) flush_copies();                               // an inactive lane holds no held-back uploads: whoever reads their targets from another lane finds them queued
    swap_lane(*lanes[cur_lane]);                             // the active lane's members into its storage ...
    swap_lane(*lanes[lane]);                                 // ... and this lane's out of its own
    cur_lane = lane;
    if (lane != 0) {
        if (lanes[lane]->fork_seen != fork_gen) { CK(hipStreamWaitEvent(stream_, ev_fork, 0)); lanes[lane]->fork_seen = fork_gen; }
        lanes[lane]->dirty = true;
    }
    return 0;
}
int cnmfe_ctx::join_lanes() {
    if (lanes.empty()) return 0;
    // the other lanes first say where they are (each on its own stream, its held-back uploads sent), then lane 0 waits for those points
    for (int l = 1; l < (int)lanes.size(); ++l) {
        if (!lanes[l]->dirty) continue;
        RET(activate(l));
        CK(hipEventRecord(lanes[l]->ev, st()));
    }
    RET(activate(0));
    for (int l = 1; l < (int)lanes.size(); ++l)
        if (lanes[l]->dirty) { CK(hipStreamWaitEvent(st(), lanes[l]->ev, 0)); lanes[l]->dirty = false; }
    return 0;
}
int cnmfe_ctx::fork_mark() {
    if (lanes.empty()) return 0;
    RET(activate(0));
    CK(hipEventRecord(ev_fork, st()));
    ++fork_gen;
    return 0;
}
static int lanes_set(cnmfe_ctx *ctx, int64_t n) {
    if (n < 1 || n > 4) return fail(CNMFE_EINVAL, "option lanes = %lld: 1 .. 4", (long long)n);
    if (!ctx->patches.empty() && (int)std::max<size_t>(1, ctx->lanes.size()) != n) return fail(CNMFE_ESTATE, "option lanes is set before the first patch is created");
    if (n == 1 || (int)ctx->lanes.size() == n) return 0;
    CK(hipSetDevice(ctx->device));
    for (int l = 0; l < n; ++l) {
        cnmfe_lane *L = new cnmfe_lane();
        ctx->lanes.push_back(L);
        CK(hipEventCreateWithFlags(&L->ev, hipEventDisableTiming));
        if (l == 0) continue;                                // (lane 0 is the active one: its members are the context's own)
        CK(hipStreamCreate(&L->stream_));
        hipLaunchKernelGGL(k_scratch_warm, dim3(1), dim3(64), 0, L->stream_, (int *)nullptr, 3);       // (its hardware queue now, not inside the first update)
        CK(hipStreamSynchronize(L->stream_));
        if (L->pin.init(size_t(64) << 20) != 0) return fail(CNMFE_EHIP, "pinned staging arena (64 MB) of lane %d could not be allocated", l);
    }
    CK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    CK(hipEventRecord(ctx->ev_fork, ctx->st()));
    return 0;
}

extern "C" {

const char *cnmfe_last_error(void) { return g_err; }
const char *cnmfe_version(void) { return "cnmfe-mi355x 0.1 (gfx950)"; }

cnmfe_ctx *cnmfe_create(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(CNMFE_EHIP, "no HIP device visible (%s)", hipGetErrorString(e)); return nullptr; }
    if (device < 0 || device >= n) { fail(CNMFE_EINVAL, "device %d out of range (0..%d)", device, n - 1); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(CNMFE_EHIP, "hipSetDevice(%d) failed", device); return nullptr; }
    cnmfe_ctx *ctx = new cnmfe_ctx();
    ctx->device = device;
    if (hipStreamCreate(&ctx->stream_) != hipSuccess) { fail(CNMFE_EHIP, "hipStreamCreate failed"); delete ctx; return nullptr; }
    // the download stream of the lazy traces and its events now, not inside the first temporal update (creating a stream is a new hardware queue: 20-25 ms of the first
    // iteration after an upload went there, scripts/gpu/r6_call32.sh)
    if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_bound_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_copy_done, hipEventDisableTiming) != hipSuccess) { fail(CNMFE_EHIP, "copy stream / events could not be created"); delete ctx; return nullptr; }
    pin_register(ctx, true);
    if (ctx->pin.init(size_t(64) << 20) != 0) { fail(CNMFE_EHIP, "pinned staging arena (64 MB) could not be allocated"); delete ctx; return nullptr; }
    // every translation unit's code object is loaded now, not at the first launch of one of its kernels inside the first iteration (tens of milliseconds in all)
    hipLaunchKernelGGL(k_scratch_warm, dim3(1), dim3(64), 0, ctx->stream_, (int *)nullptr, 3);
    hipLaunchKernelGGL(k_scratch_warm, dim3(1), dim3(64), 0, ctx->copy_stream, (int *)nullptr, 3);      // (a stream gets its hardware queue at its first use)
    (void)hipStreamSynchronize(ctx->stream_); (void)hipStreamSynchronize(ctx->copy_stream); (void)hipGetLastError();
    (void)tu_warm_resid(); (void)tu_warm_bg(); (void)tu_warm_factor(); (void)tu_warm_deconv(); (void)tu_warm_ssub(); (void)tu_warm_vproj();
    // CNMFE_OPTS="name=value,name=value": tunables of cnmfe_set_option preset for every context of the process (A/B runs of the test suite and the bench
    // without touching their code); names this build does not know are ignored -- the same environment serves builds with different option sets
    if (const char *env = getenv("CNMFE_OPTS")) {
        std::string s(env), applied, ignored;
        size_t pos = 0;
        while (pos < s.size()) {
            size_t end = s.find(',', pos); if (end == std::string::npos) end = s.size();
            const std::string kv = s.substr(pos, end - pos); pos = end + 1;
            const size_t eq = kv.find('=');
            if (eq == std::string::npos || eq == 0) continue;
            const int rc_ = cnmfe_set_option(ctx, kv.substr(0, eq).c_str(), strtoll(kv.c_str() + eq + 1, nullptr, 10));
            (rc_ == 0 ? applied : ignored) += (rc_ == 0 ? (applied.empty() ? "" : ",") : (ignored.empty() ? "" : ",")) + kv;
        }
        static bool said = false;                              // once per process: a preset that silently changes every context is worth one line
        if (!said) { said = true; fprintf(stderr, "[cnmfe] CNMFE_OPTS applied to every context: %s%s%s\n", applied.empty() ? "(none)" : applied.c_str(), ignored.empty() ? "" : "; not known to this build, ignored: ", ignored.c_str()); }
    }
    return ctx;
}

void cnmfe_destroy(cnmfe_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)ctx->join_lanes();
    (void)hipStreamSynchronize(ctx->st());
    delete ctx;
}

int cnmfe_synchronize(cnmfe_ctx *ctx) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    RET(ctx->join_lanes());                                  // (every lane's work in front of lane 0's stream)
    CK(hipStreamSynchronize(ctx->st()));
    if (ctx->copy_stream) CK(hipStreamSynchronize(ctx->copy_stream));
    return ctx_check_errflag(ctx);
}

int cnmfe_set_option(cnmfe_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return fail(CNMFE_EINVAL, "null argument");
    // behaviour switches (include/cnmfe.h) and, behind them, the probes of scripts/ (diagnostics: solve_probe, r1_probe, deconv_trace, host_trace, debug)
    static const char *known[] = {"r1_variant", "r1_delta", "r1_lazy", "r1_defer", "r1_virtual", "gram_incremental", "prealloc", "solve_packed", "sweep_dag", "win_i8_planes", "solve_staged", "solve_inv", "solve_inv_terms", "gram_i8", "win_i8", "proj_tiled", "proj_i8", "proj_i8_planes", "ssub_virtual",
                                  "solve_probe", "r1_probe", "deconv_trace", "host_trace", "debug", nullptr};
    if (!strcmp(name, "lanes")) { RET(lanes_set(ctx, value)); ctx->opts[name] = value; return 0; }
    for (int i = 0; known[i]; ++i) if (!strcmp(known[i], name)) { ctx->opts[name] = value; if (!strcmp(name, "host_trace")) ctx->trace_level = (int)value; return 0; }
    return fail(CNMFE_EINVAL, "unknown option '%s'", name);
}

int cnmfe_patch_create(cnmfe_ctx *ctx, int patch_id, const int32_t pr[4], const int32_t br[4], int32_t d1, int32_t d2, int64_t T) {
    if (!ctx || !pr || !br) return fail(CNMFE_EINVAL, "null argument");
    if (d1 <= 0 || d2 <= 0 || T <= 0) return fail(CNMFE_EINVAL, "bad dims d1=%d d2=%d T=%lld", d1, d2, (long long)T);
    if (!(1 <= br[0] && br[0] <= pr[0] && pr[0] <= pr[1] && pr[1] <= br[1] && br[1] <= d1 &&
          1 <= br[2] && br[2] <= pr[2] && pr[2] <= pr[3] && pr[3] <= br[3] && br[3] <= d2))
        return fail(CNMFE_EINVAL, "patch [%d %d %d %d] must lie inside block [%d %d %d %d] inside the %dx%d FOV",
                    pr[0], pr[1], pr[2], pr[3], br[0], br[1], br[2], br[3], d1, d2);
    CK(hipSetDevice(ctx->device));
    if (ctx->patches.count(patch_id)) { delete ctx->patches[patch_id]; ctx->patches.erase(patch_id); }
    Patch *P = new Patch();
    P->lane = ctx->lanes.empty() ? 0 : (ctx->patches_created++ % (int)ctx->lanes.size());
    memcpy(P->prect, pr, sizeof(P->prect)); memcpy(P->brect, br, sizeof(P->brect));
    P->d1 = d1; P->d2 = d2; P->T = T;
    P->nr = pr[1] - pr[0] + 1; P->nc = pr[3] - pr[2] + 1;
    P->nr_b = br[1] - br[0] + 1; P->nc_b = br[3] - br[2] + 1;
    P->roff = pr[0] - br[0]; P->coff = pr[2] - br[2];
    P->d = (int64_t)P->nr * P->nc; P->d_b = (int64_t)P->nr_b * P->nc_b;
    int rc = P->Y.ensure((size_t)P->d_b * T * sizeof(float));
    if (rc) { delete P; return rc; }
    if (hipMemsetAsync(P->Y.p, 0, (size_t)P->d_b * T * sizeof(float), ctx->st()) != hipSuccess) { delete P; return fail(CNMFE_EHIP, "hipMemsetAsync of the upload staging failed"); }
    P->frame_seen.assign((size_t)T, 0);
    P->Tc = (T + 3) / 4;
    ctx->patches[patch_id] = P;
    return 0;
}

int cnmfe_upload_block(cnmfe_ctx *ctx, int patch_id, const void *Y, int dtype, int memspace, int64_t t0, int64_t nt) {
    if (!ctx || !Y) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (t0 < 0 || nt <= 0 || t0 + nt > P->T) return fail(CNMFE_EINVAL, "frames [%lld,%lld) outside [0,%lld)", (long long)t0, (long long)(t0 + nt), (long long)P->T);
    CK(hipSetDevice(ctx->device));
    size_t esz = dtype == CNMFE_F32 ? 4 : dtype == CNMFE_F64 ? 8 : dtype == CNMFE_U16 ? 2 : dtype == CNMFE_U8 ? 1 : dtype == CNMFE_F16 ? 2 : 0;
    if (!esz) return fail(CNMFE_EINVAL, "unknown dtype %d", dtype);
    if (memspace != CNMFE_HOST && memspace != CNMFE_DEVICE) return fail(CNMFE_EINVAL, "unknown memspace %d", memspace);
    if (!P->Y.p) {                                   // a re-upload after the block was finalised: start over with a zeroed staging copy
        RET(P->Y.ensure((size_t)P->d_b * P->T * sizeof(float)));
        CK(hipMemsetAsync(P->Y.p, 0, (size_t)P->d_b * P->T * sizeof(float), ctx->st()));
        P->frames_uploaded = 0;
        P->frame_seen.assign((size_t)P->T, 0);
    }
    for (int64_t t = t0; t < t0 + nt; ++t)           // every frame arrives exactly once: a repeated or overlapping chunk is a caller error, not a silent overwrite
        if (P->frame_seen[(size_t)t]) return fail(CNMFE_EINVAL, "frame %lld of patch %d was already uploaded (chunks must not overlap; re-create the patch to replace its video)", (long long)t, patch_id);
    float *dst = P->Y.as<float>() + t0 * P->d_b;
    int64_t n = nt * P->d_b;
    if (dtype == CNMFE_F32) {
        CK(hipMemcpyAsync(dst, Y, (size_t)n * 4, memspace == CNMFE_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->st()));
    } else {
        // stage in slabs of <= 64 Mi elements, convert on device
        const int64_t slab = int64_t(1) << 26;
        for (int64_t o = 0; o < n; o += slab) {
            int64_t m = std::min(slab, n - o);
            const void *src = (const char *)Y + (size_t)o * esz;
            if (memspace == CNMFE_HOST) {
                RET(ctx->stage.ensure((size_t)m * esz));
                CK(hipMemcpyAsync(ctx->stage.p, src, (size_t)m * esz, hipMemcpyHostToDevice, ctx->st()));
                src = ctx->stage.p;
            }
            dim3 g((unsigned)std::min<int64_t>((m + 255) / 256, 65535)), b(256);
            switch (dtype) {
                case CNMFE_F64: LAUNCH(ctx, "convert", k_convert<double>, g, b, 0, (const double *)src, dst + o, m); break;
                case CNMFE_U16: LAUNCH(ctx, "convert", k_convert<uint16_t>, g, b, 0, (const uint16_t *)src, dst + o, m); break;
                case CNMFE_U8:  LAUNCH(ctx, "convert", k_convert<uint8_t>, g, b, 0, (const uint8_t *)src, dst + o, m); break;
                default:        LAUNCH(ctx, "convert", k_convert<__half>, g, b, 0, (const __half *)src, dst + o, m); break;
            }
            if (memspace == CNMFE_HOST) CK(hipStreamSynchronize(ctx->st()));   // staging buffer reuse
        }
    }
    CK(hipStreamSynchronize(ctx->st()));
    std::fill(P->frame_seen.begin() + t0, P->frame_seen.begin() + t0 + nt, (uint8_t)1);
    P->frames_uploaded += nt;
    P->ymean_valid = false; P->ysig_valid = false; P->base_valid = false; P->base_alt_valid = false; P->sys_valid = false; P->sys_alt_valid = false; P->kinv_valid = false; P->kinv_lam_valid = false; P->kinv_fits = 0; P->pt_valid = false; P->dig_valid = false; P->digp_valid = false; P->yt4_valid = false;
    return 0;
}

int cnmfe_get_ymean(cnmfe_ctx *ctx, int patch_id, double *out) {
    if (!ctx || !out) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    CK(hipSetDevice(ctx->device));
    RET(ensure_ymean(ctx, P));
    CK(hipMemcpyAsync(out, P->ymean_d.p, P->d_b * sizeof(double), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    return 0;
}

int cnmfe_ring_init(cnmfe_ctx *ctx, int patch_id, int32_t radius, int32_t num_neighbors) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (radius < 1 || radius > 64) return fail(CNMFE_EINVAL, "ring radius %d out of range", radius);
    CK(hipSetDevice(ctx->device));
    // validate on locals first: a rejected call leaves the patch's ring (offsets, p, W) exactly as it was
    std::vector<int32_t> ndr, ndc;
    ring_offsets(radius, num_neighbors, ndr, ndc);
    const int32_t np_ = (int32_t)ndr.size();
    if (np_ < 1) return fail(CNMFE_EINVAL, "ring of radius %d has no neighbours", radius);
    if (np_ > PMAX_RING) return fail(CNMFE_EUNSUPPORTED, "ring with %d neighbours exceeds the supported %d", np_, PMAX_RING);
    if (radius > 24) return fail(CNMFE_EUNSUPPORTED, "ring radius %d: the block-pair covariance table of the ring regression covers radii <= 24", radius);
    // every in-FOV ring neighbour of a patch pixel must be inside the block
    int h = radius;
    if ((P->prect[0] - P->brect[0] < std::min(h, P->prect[0] - 1)) || (P->brect[1] - P->prect[1] < std::min(h, P->d1 - P->prect[1])) ||
        (P->prect[2] - P->brect[2] < std::min(h, P->prect[2] - 1)) || (P->brect[3] - P->prect[3] < std::min(h, P->d2 - P->prect[3])))
        return fail(CNMFE_EINVAL, "block halo is narrower than the ring radius %d", radius);
    P->ring_ready = false;                                  // from here on a failure leaves "no ring", never a mismatched one
    P->dr.swap(ndr); P->dc.swap(ndc); P->p = np_; P->radius = radius;
    RET(to_dev(ctx, P->ring_dr, P->dr.data(), P->dr.size()));
    RET(to_dev(ctx, P->ring_dc, P->dc.data(), P->dc.size()));
    RET(P->W.ensure((size_t)P->p * P->d * sizeof(float)));
    RET(P->b0.ensure(P->d * sizeof(double)));
    CK(hipMemsetAsync(P->b0.p, 0, P->d * sizeof(double), ctx->st()));          // initComponents_parallel.m:221
    LAUNCH(ctx, "ring_init", k_ring_init, dim3((unsigned)((P->d + 255) / 256)), dim3(256), 0,
           P->W.as<float>(), P->d, P->nr, P->p, P->ring_dr.as<int>(), P->ring_dc.as<int>(), P->prect[0], P->prect[2], P->d1, P->d2);
    CK(hipStreamSynchronize(ctx->st()));
    P->stat_valid = false;
    if (P->stat_host) { (void)hipHostFree(P->stat_host); P->stat_host = nullptr; }      // sized by the number of ring offsets
    P->ring_ready = true; P->ysig_valid = false; P->base_valid = false; P->base_alt_valid = false; P->sys_valid = false; P->sys_alt_valid = false; P->kinv_valid = false; P->kinv_lam_valid = false; P->kinv_fits = 0; P->pt_valid = false;   // (the kept covariance tables cover the sub-tiles THIS ring needs)
    return 0;
}

static int64_t ring_count_nnz(const Patch *P) {
    int64_t nnz = 0;
    for (int i = 0; i < P->p; ++i) {
        // rows r with 1 <= r + dr <= d1 within the patch rows, same for columns
        int64_t rlo = std::max<int64_t>(P->prect[0], 1 - P->dr[i]), rhi = std::min<int64_t>(P->prect[1], P->d1 - P->dr[i]);
        int64_t clo = std::max<int64_t>(P->prect[2], 1 - P->dc[i]), chi = std::min<int64_t>(P->prect[3], P->d2 - P->dc[i]);
        if (rhi >= rlo && chi >= clo) nnz += (rhi - rlo + 1) * (chi - clo + 1);
    }
    return nnz;
}

int cnmfe_ring_nnz(cnmfe_ctx *ctx, int patch_id, int64_t *nnz, int32_t *p) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    if (nnz) *nnz = ring_count_nnz(P);
    if (p) *p = P->p;
    return 0;
}

int cnmfe_ring_get_csr(cnmfe_ctx *ctx, int patch_id, int64_t *rowptr, int32_t *col, float *val) {
    if (!ctx || !rowptr || !col || !val) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    CK(hipSetDevice(ctx->device));
    std::vector<float> W((size_t)P->p * P->d);
    CK(hipMemcpyAsync(W.data(), P->W.p, W.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    int64_t e = 0;
    for (int64_t m = 0; m < P->d; ++m) {
        rowptr[m] = e;
        int r = P->prect[0] + (int)(m % P->nr), c = P->prect[2] + (int)(m / P->nr);
        for (int i = 0; i < P->p; ++i) {
            int rr = r + P->dr[i], cc = c + P->dc[i];
            if (rr < 1 || rr > P->d1 || cc < 1 || cc > P->d2) continue;
            col[e] = (int32_t)((int64_t)(cc - P->brect[2]) * P->nr_b + (rr - P->brect[0]));
            val[e] = W[(size_t)i * P->d + m];
            ++e;
        }
    }
    rowptr[P->d] = e;
    return 0;
}

int cnmfe_ring_set_values(cnmfe_ctx *ctx, int patch_id, const float *val) {
    if (!ctx || !val) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    CK(hipSetDevice(ctx->device));
    std::vector<float> W((size_t)P->p * P->d, 0.f);
    int64_t e = 0;
    for (int64_t m = 0; m < P->d; ++m) {
        int r = P->prect[0] + (int)(m % P->nr), c = P->prect[2] + (int)(m / P->nr);
        for (int i = 0; i < P->p; ++i) {
            int rr = r + P->dr[i], cc = c + P->dc[i];
            if (rr < 1 || rr > P->d1 || cc < 1 || cc > P->d2) continue;
            W[(size_t)i * P->d + m] = val[e++];
        }
    }
    CK(hipMemcpyAsync(P->W.p, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    P->stat_valid = false;
    P->ysig_valid = false;
    return 0;
}

int cnmfe_ring_first_run(cnmfe_ctx *ctx, int patch_id, int *first_run) {
    if (!ctx || !first_run) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    CK(hipSetDevice(ctx->device));
    bool f = false;
    RET(ring_first_run(ctx, P, &f));
    *first_run = f ? 1 : 0;
    return 0;
}

int cnmfe_b0_get(cnmfe_ctx *ctx, int patch_id, float *b0) {
    if (!ctx || !b0) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    CK(hipSetDevice(ctx->device));
    std::vector<double> tmp(P->d);
    CK(hipMemcpyAsync(tmp.data(), P->b0.p, P->d * sizeof(double), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    for (int64_t i = 0; i < P->d; ++i) b0[i] = (float)tmp[i];
    return 0;
}

int cnmfe_b0_set(cnmfe_ctx *ctx, int patch_id, const float *b0) {
    if (!ctx || !b0) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    CK(hipSetDevice(ctx->device));
    std::vector<double> tmp(P->d);
    for (int64_t i = 0; i < P->d; ++i) tmp[i] = (double)b0[i];
    CK(hipMemcpyAsync(P->b0.p, tmp.data(), P->d * sizeof(double), hipMemcpyHostToDevice, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    P->ysig_valid = false;
    return 0;
}

static int check_csc(const char *what, int32_t K, int64_t nrow, const int64_t *colptr, const int32_t *rowidx) {
    if (K < 0) return fail(CNMFE_EINVAL, "%s: K=%d", what, K);
    if (K == 0) return 0;
    if (!colptr) return fail(CNMFE_EINVAL, "%s: null colptr", what);
    if (colptr[0] != 0) return fail(CNMFE_EINVAL, "%s: colptr[0] != 0", what);
    for (int32_t k = 0; k < K; ++k) if (colptr[k + 1] < colptr[k]) return fail(CNMFE_EINVAL, "%s: colptr not monotone at %d", what, k);
    int64_t nnz = colptr[K];
    if (nnz && !rowidx) return fail(CNMFE_EINVAL, "%s: null rowidx", what);
    for (int32_t k = 0; k < K; ++k)
        for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) {
            if (rowidx[e] < 0 || rowidx[e] >= nrow) return fail(CNMFE_EINVAL, "%s: row index %d out of [0,%lld)", what, rowidx[e], (long long)nrow);
            if (e > colptr[k] && rowidx[e] <= rowidx[e - 1]) return fail(CNMFE_EINVAL, "%s: rows of column %d not strictly ascending", what, k);
        }
    return 0;
}
extern "C++" { namespace cnmfe { int check_csc_pub(const char *what, int32_t ncol, int64_t nrow, const int64_t *colptr, const int32_t *rowidx) { return check_csc(what, ncol, nrow, colptr, rowidx); } } }

int cnmfe_ring_solve_stats(cnmfe_ctx *ctx, int patch_id, int64_t out[4]) {
    if (!ctx || !out) return fail(CNMFE_EINVAL, "null argument");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_EINVAL, "no patch %d", patch_id);
    out[0] = out[1] = out[2] = out[3] = -1;
    if (!P->kinv_valid || !P->kinv_list.p) return 0;
    CK(hipSetDevice(ctx->device));
    int h[4];
    CK(hipMemcpyAsync(h, P->kinv_list.as<int>() + 2 * P->d, sizeof(h), hipMemcpyDeviceToHost, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));
    for (int i = 0; i < 4; ++i) out[i] = h[i];
    return 0;
}

int cnmfe_fit_ring_model(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                         const float *A_val, const float *C, int c_order, double thresh_outlier, int with_projection,
                         float *b0_out, int64_t info[4]) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    RET(check_csc("A", K, P->d_b, A_colptr, A_rowidx));
    if (K > 0 && ((!A_val && A_colptr[K] > 0) || (!C && c_order != CNMFE_BOUND))) return fail(CNMFE_EINVAL, "null A_val / C");
    CK(hipSetDevice(ctx->device));
    RET(ensure_ymean(ctx, P));
    int64_t dummy[4];
    int rc = bg_fit_ring(ctx, P, K, A_colptr, A_rowidx, A_val, C, c_order, with_projection, b0_out, info ? info : dummy, 0, thresh_outlier);
    P->ysig_valid = false;
    return rc;
}

int cnmfe_fit_reserve(cnmfe_ctx *ctx, int patch_id) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    CK(hipSetDevice(ctx->device));
    if (ctx->opt("prealloc", 1) == 0) return 0;
    return bg_reserve(ctx, P);
}

int cnmfe_set_noise(cnmfe_ctx *ctx, int patch_id, const float *sn_block) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!sn_block) return fail(CNMFE_EINVAL, "null sn_block");
    CK(hipSetDevice(ctx->device));
    RET(to_dev(ctx, P->sn_b, sn_block, (size_t)P->d_b));
    CK(hipStreamSynchronize(ctx->st()));
    P->sn_ready = true;
    return 0;
}

int cnmfe_residual(cnmfe_ctx *ctx, int patch_id, int32_t Ksel, const int64_t *A_colptr, const int32_t *A_rowidx,
                   const float *A_val, const float *C, int c_order, float *Ysig_out, int out_memspace) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P || !P->ring_ready) return fail(CNMFE_ESTATE, "ring of patch %d not initialised", patch_id);
    RET(check_csc("A_prev", Ksel, P->d_b, A_colptr, A_rowidx));
    if (Ksel > 0 && ((!A_val && A_colptr[Ksel] > 0) || (!C && c_order != CNMFE_BOUND))) return fail(CNMFE_EINVAL, "null A_val / C");
    CK(hipSetDevice(ctx->device));
    RET(ensure_ymean(ctx, P));
    return residual_run(ctx, P, patch_id, Ksel, A_colptr, A_rowidx, A_val, C, c_order, Ysig_out, out_memspace);
}

int cnmfe_get_sn(cnmfe_ctx *ctx, int patch_id, float *sn_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    RET(residual_materialize(ctx, P));                       // a pending footprint term must be in Ysig for this consumer
    if (!sn_out) return fail(CNMFE_EINVAL, "null sn_out");
    CK(hipSetDevice(ctx->device));
    return sn_pixels_run(ctx, P, sn_out);
}

int cnmfe_estimate_noise(cnmfe_ctx *ctx, int patch_id, int64_t nframes, float *sn_block_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!sn_block_out) return fail(CNMFE_EINVAL, "null sn_block_out");
    CK(hipSetDevice(ctx->device));
    RET(ensure_ymean(ctx, P));
    return sn_video_run(ctx, P, nframes, sn_block_out);
}

int cnmfe_update_spatial(cnmfe_ctx *ctx, int patch_id, int algorithm, int32_t K, const int64_t *A_colptr,
                         const int32_t *A_rowidx, const float *A_val, const float *C, int c_order,
                         const int64_t *IND_colptr, const int32_t *IND_rowidx, const float *sn, int32_t param, float *A_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    // (a pending footprint term enters through the projection, spatial_run; only a term that cannot is folded into Ysig first)
    if (algorithm < CNMFE_SPATIAL_HALS || algorithm > CNMFE_SPATIAL_NNLS) return fail(CNMFE_EINVAL, "unknown spatial algorithm %d", algorithm);
    if (K <= 0) return fail(CNMFE_EINVAL, "K=%d", K);
    RET(check_csc("A", K, P->d, A_colptr, A_rowidx));
    RET(check_csc("IND", K, P->d, IND_colptr, IND_rowidx));
    if (!C && c_order != CNMFE_BOUND) return fail(CNMFE_EINVAL, "null C");
    if (algorithm == CNMFE_SPATIAL_HALS_THRESH && !sn) return fail(CNMFE_EINVAL, "HALS_THRESH needs sn");
    if (param <= 0) return fail(CNMFE_EINVAL, "maxIter/maxN must be positive");
    CK(hipSetDevice(ctx->device));
    ctx->spatial_lane = P->lane;                            // (the fetch calls carry no patch id: they look at this lane's scratch)
    return spatial_run(ctx, P, algorithm, K, A_colptr, A_rowidx, A_val, C, c_order, IND_colptr, IND_rowidx, sn, param, A_out);
}

int cnmfe_update_spatial_fetch(cnmfe_ctx *ctx, float *A_out, int64_t nnz) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (!A_out && nnz) return fail(CNMFE_EINVAL, "null A_out");
    CK(hipSetDevice(ctx->device));
    RET(ctx->activate(ctx->spatial_lane));
    return spatial_fetch(ctx, A_out, nnz);
}

// A ticket = an event behind the work queued so far.  The device error word travels with it: a caller that waits for its ticket only -- not for the
// stream, cnmfe_synchronize -- still hears what the kernels in front of the ticket had to report (a ring over too many footprints, an inconsistent table)
// before it uses what they computed.
static int ticket_record(cnmfe_ctx *ctx, int64_t *ticket, const void *src = nullptr, void *dst_pinned = nullptr, size_t bytes = 0) {
    size_t t = 0;
    while (t < ctx->tickets.size() && ctx->ticket_busy[t]) ++t;
    if (t == ctx->tickets.size()) {
        hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->tickets.push_back(e); ctx->ticket_busy.push_back(0);
    }
    const bool flagged = ctx->errflag.p && t < cnmfe_ctx::TICKET_FLAGS;
    if (flagged) {
        if (!ctx->ticket_flags) { CK(hipHostMalloc((void **)&ctx->ticket_flags, cnmfe_ctx::TICKET_FLAGS * sizeof(int), hipHostMallocDefault)); memset(ctx->ticket_flags, 0, cnmfe_ctx::TICKET_FLAGS * sizeof(int)); }
        ctx->ticket_flags[t] = 0;
    }
    if (bytes) {                                             // the download and the flag in one kernel (k_fetch_take): both ends 16-byte granular with room (DevBuf: 256 B, the caller's block: a power of two)
        const unsigned n16 = (unsigned)((bytes + 15) / 16), nb = std::min<unsigned>((n16 + 255) / 256, 64);
        hipLaunchKernelGGL(k_fetch_take, dim3(nb), dim3(256), 0, ctx->st(), (const uint4 *)src, (uint4 *)dst_pinned, n16, flagged ? ctx->errflag.as<int>() : (int *)nullptr,
                           flagged ? ctx->ticket_flags + t : (int *)nullptr);
    } else if (flagged) {
        RET(ctx->ticket_dev.ensure(cnmfe_ctx::TICKET_FLAGS * sizeof(int)));
        hipLaunchKernelGGL(k_flag_take, dim3(1), dim3(1), 0, ctx->st(), ctx->errflag.as<int>(), ctx->ticket_dev.as<int>() + t);
        CK(hipMemcpyAsync(&ctx->ticket_flags[t], ctx->ticket_dev.as<int>() + t, sizeof(int), hipMemcpyDeviceToHost, ctx->st()));
    }
    CK(hipEventRecord(ctx->tickets[t], ctx->st()));
    ctx->ticket_busy[t] = 1;
    *ticket = (int64_t)t;
    return 0;
}

int cnmfe_update_spatial_fetch_async(cnmfe_ctx *ctx, float *A_out_pinned, int64_t nnz, int64_t *ticket) {
    if (!ctx || !ticket) return fail(CNMFE_EINVAL, "null context / ticket");
    if (!A_out_pinned && nnz) return fail(CNMFE_EINVAL, "null A_out");
    CK(hipSetDevice(ctx->device));
    RET(ctx->activate(ctx->spatial_lane));
    if (ctx->spatial_nnz < 0 || nnz != ctx->spatial_nnz) return fail(CNMFE_ESTATE, "no deferred spatial update of %lld values (last one: %lld)", (long long)nnz, (long long)ctx->spatial_nnz);
    // A_out_pinned: PINNED host memory (cnmfe_host_alloc) of at least nnz floats rounded up to 16 bytes -- the copy is a kernel writing through the mapping
    return ticket_record(ctx, ticket, ctx->scr[6].p, A_out_pinned, (size_t)nnz * sizeof(float));
}

int cnmfe_ticket_wait(cnmfe_ctx *ctx, int64_t ticket) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (ticket < 0 || (size_t)ticket >= ctx->tickets.size() || !ctx->ticket_busy[ticket]) return fail(CNMFE_ESTATE, "ticket %lld is not outstanding", (long long)ticket);
    CK(hipSetDevice(ctx->device));
    CK(hipEventSynchronize(ctx->tickets[ticket]));
    ctx->ticket_busy[ticket] = 0;
    if ((size_t)ticket >= cnmfe_ctx::TICKET_FLAGS) return ctx_check_errflag(ctx);          // (more tickets outstanding than flag slots: the whole stream's word, at the price of a drain)
    if (ctx->ticket_flags && ctx->ticket_flags[ticket]) {
        const int h = ctx->ticket_flags[ticket];
        ctx->ticket_flags[ticket] = 0;
        return errflag_raise(ctx, h, true);
    }
    return 0;
}

int cnmfe_update_spatial_fetch_connected(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                                         float *A_out, uint8_t *keep_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (K <= 0 || d1 <= 0 || d2 <= 0) return fail(CNMFE_EINVAL, "bad K / d1 / d2");
    RET(check_csc("IND", K, (int64_t)d1 * d2, IND_colptr, IND_rowidx));
    if (IND_colptr[K] && (!A_out || !keep_out)) return fail(CNMFE_EINVAL, "null A_out / keep_out");
    CK(hipSetDevice(ctx->device));
    RET(ctx->activate(ctx->spatial_lane));
    return spatial_fetch_connected(ctx, d1, d2, K, IND_colptr, IND_rowidx, A_out, keep_out);
}

int cnmfe_update_spatial_fetch_connected_async(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                                               float *A_out_pinned, uint8_t *keep_out_pinned, int64_t *ticket) {
    if (!ctx || !ticket) return fail(CNMFE_EINVAL, "null context / ticket");
    if (K <= 0 || d1 <= 0 || d2 <= 0) return fail(CNMFE_EINVAL, "bad K / d1 / d2");
    RET(check_csc("IND", K, (int64_t)d1 * d2, IND_colptr, IND_rowidx));
    if (IND_colptr[K] && (!A_out_pinned || !keep_out_pinned)) return fail(CNMFE_EINVAL, "null A_out / keep_out");
    CK(hipSetDevice(ctx->device));
    RET(ctx->activate(ctx->spatial_lane));
    RET(spatial_fetch_connected(ctx, d1, d2, K, IND_colptr, IND_rowidx, A_out_pinned, keep_out_pinned, false));
    return ticket_record(ctx, ticket);
}

int cnmfe_hals_temporal(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                        const float *A_val, const float *C_in, int c_order, int32_t maxIter,
                        float *C_out, float *C_raw_out, float *aa_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    if (K <= 0) return fail(CNMFE_EINVAL, "K=%d", K);
    RET(check_csc("A", K, P->d, A_colptr, A_rowidx));
    if ((!A_val && A_colptr[K] > 0) || (!C_in && c_order != CNMFE_BOUND)) return fail(CNMFE_EINVAL, "null A_val / C_in");   // all-zero columns are legal: aa = 0, row left alone (HALS_temporal.m:51)
    if (maxIter <= 0) return fail(CNMFE_EINVAL, "maxIter must be positive");
    CK(hipSetDevice(ctx->device));
    return temporal_run(ctx, P, K, A_colptr, A_rowidx, A_val, C_in, c_order, maxIter, C_out, C_raw_out, aa_out, nullptr, nullptr, nullptr, nullptr);
}

int cnmfe_hals_temporal_deconv(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                               const float *A_val, const float *C_in, int c_order, int32_t maxIter, const cnmfe_deconv_opts *opts,
                               float *kernel_pars, float *C_out, float *C_raw_out, float *S_out, float *sn_out, float *aa_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    if (K <= 0) return fail(CNMFE_EINVAL, "K=%d", K);
    RET(check_csc("A", K, P->d, A_colptr, A_rowidx));
    if ((!A_val && A_colptr[K] > 0) || (!C_in && c_order != CNMFE_BOUND) || !opts || !kernel_pars) return fail(CNMFE_EINVAL, "null A_val / C_in / opts / kernel_pars");
    if (maxIter <= 0) return fail(CNMFE_EINVAL, "maxIter must be positive");
    CK(hipSetDevice(ctx->device));
    return temporal_run(ctx, P, K, A_colptr, A_rowidx, A_val, C_in, c_order, maxIter, C_out, C_raw_out, aa_out, opts, kernel_pars, S_out, sn_out);
}

int cnmfe_fast_temporal(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                        int c_order, float *C_raw_out, float *aa_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    RET(residual_materialize(ctx, P));                       // a pending footprint term must be in Ysig for this consumer
    if (K <= 0) return fail(CNMFE_EINVAL, "K=%d", K);
    RET(check_csc("A", K, P->d, A_colptr, A_rowidx));
    if (!A_val && A_colptr[K] > 0) return fail(CNMFE_EINVAL, "null A_val");
    CK(hipSetDevice(ctx->device));
    return fast_temporal_run(ctx, P, K, A_colptr, A_rowidx, A_val, c_order, C_raw_out, aa_out);
}

int cnmfe_compute_rss(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                      const float *C, int c_order, const float *b0_block, const float *b0_new, double *rss_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    if (P->res_kind != 1) return fail(CNMFE_EUNSUPPORTED, "compute_RSS needs the residual of cnmfe_residual (bg_ssub = 1)");
    if (K < 0 || !b0_block || !b0_new || !rss_out) return fail(CNMFE_EINVAL, "bad K / null b0_block / b0_new / rss_out");
    if (K > 0) {
        RET(check_csc("A", K, P->d, A_colptr, A_rowidx));
        if ((!A_val && A_colptr[K] > 0) || (!C && c_order != CNMFE_BOUND)) return fail(CNMFE_EINVAL, "null A_val / C");
    }
    CK(hipSetDevice(ctx->device));
    return rss_run(ctx, P, K, A_colptr, A_rowidx, A_val, C, c_order, b0_block, b0_new, rss_out);
}

int cnmfe_reconstruct_background(cnmfe_ctx *ctx, int patch_id, const float *b0_block, const float *b0_new, int64_t frame0, int64_t nframes,
                                  float *Ybg_out, int out_memspace) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    if (P->res_kind != 1) return fail(CNMFE_EUNSUPPORTED, "reconstruct_background needs the residual of cnmfe_residual (bg_ssub = 1)");
    if (!b0_block || !b0_new || !Ybg_out) return fail(CNMFE_EINVAL, "null b0_block / b0_new / Ybg_out");
    if (frame0 < 0 || nframes <= 0 || frame0 + nframes > P->T || nframes > 65535) return fail(CNMFE_EINVAL, "frames [%lld, %lld) outside [0, %lld) or more than 65535 at once", (long long)frame0, (long long)(frame0 + nframes), (long long)P->T);
    CK(hipSetDevice(ctx->device));
    return bg_reconstruct_run(ctx, P, b0_block, b0_new, frame0, nframes, Ybg_out, out_memspace);
}

int cnmfe_stitch_begin(cnmfe_ctx *ctx, int32_t K, int64_t T) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (K < 0 || T <= 0) return fail(CNMFE_EINVAL, "bad K / T");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    const int64_t ld = ((T + 3) & ~int64_t(3)) + 4;
    RET(ctx->stitch.ensure((size_t)std::max<int64_t>(1, (int64_t)K * ld) * sizeof(float)));
    CK(hipMemsetAsync(ctx->stitch.p, 0, (size_t)std::max<int64_t>(1, (int64_t)K * ld) * sizeof(float), ctx->st()));
    ctx->stitch_K = K; ctx->stitch_T = T; ctx->stitch_ld = ld; ctx->stitch_open = true;
    ctx->tjobs_used = 0;                                   // the temporal jobs of the last update are over (their buffers stay for this one's)
    return 0;
}

// ---- several patches per context: set every patch's temporal update up first (cnmfe_hals_temporal_job), sweep them together, then add each to the stitch ----
int cnmfe_hals_temporal_job(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                            const float *C_in, int c_order, int32_t maxIter, const cnmfe_deconv_opts *opts, const float *kernel_pars, int32_t *job_out) {
    if (!ctx || !job_out) return fail(CNMFE_EINVAL, "null context / job_out");
    Patch *P = get_patch(ctx, patch_id);
    if (!P) return fail(CNMFE_ESTATE, "patch %d not created", patch_id);
    if (!P->ysig_valid) return fail(CNMFE_ESTATE, "cnmfe_residual has not been run for patch %d", patch_id);
    if (!ctx->stitch_open) return fail(CNMFE_ESTATE, "cnmfe_stitch_begin has not been called (it opens a round of temporal jobs)");
    if (K <= 0) return fail(CNMFE_EINVAL, "K=%d", K);
    RET(check_csc("A", K, P->d, A_colptr, A_rowidx));
    if ((!A_val && A_colptr[K] > 0) || (!C_in && c_order != CNMFE_BOUND) || (opts && !kernel_pars)) return fail(CNMFE_EINVAL, "null A_val / C_in / kernel_pars");
    if (maxIter <= 0) return fail(CNMFE_EINVAL, "maxIter must be positive");
    CK(hipSetDevice(ctx->device));
    if (ctx->tjobs_used == (int)ctx->tjobs.size()) ctx->tjobs.push_back(new TemporalJob());
    TemporalJob *job = ctx->tjobs[ctx->tjobs_used];
    job->K = 0; job->swept = false;
    RET(temporal_run(ctx, P, K, A_colptr, A_rowidx, A_val, C_in, c_order, maxIter, nullptr, nullptr, nullptr, opts, const_cast<float *>(kernel_pars), nullptr, nullptr, job));
    *job_out = ctx->tjobs_used++;
    return 0;
}
int cnmfe_temporal_jobs_sweep(cnmfe_ctx *ctx) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    return temporal_sweep_jobs(ctx);
}
int cnmfe_stitch_add_job(cnmfe_ctx *ctx, int32_t job_id, int32_t K_m, const int32_t *ind_m) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (!ctx->stitch_open) return fail(CNMFE_ESTATE, "cnmfe_stitch_begin has not been called");
    if (job_id < 0 || job_id >= ctx->tjobs_used) return fail(CNMFE_EINVAL, "temporal job %d does not exist (%d set up since cnmfe_stitch_begin)", job_id, ctx->tjobs_used);
    TemporalJob *job = ctx->tjobs[job_id];
    if (!job->swept) return fail(CNMFE_ESTATE, "temporal job %d has not been swept (cnmfe_temporal_jobs_sweep)", job_id);
    if (K_m != job->K || !ind_m || job->T != ctx->stitch_T) return fail(CNMFE_EINVAL, "temporal job %d holds %d rows x %lld frames", job_id, job->K, (long long)job->T);
    std::vector<char> seen((size_t)ctx->stitch_K, 0);
    for (int32_t j = 0; j < K_m; ++j) {
        if (ind_m[j] < 0 || ind_m[j] >= ctx->stitch_K) return fail(CNMFE_EINVAL, "row %d outside the %d-row accumulator", ind_m[j], ctx->stitch_K);
        if (seen[ind_m[j]]) return fail(CNMFE_EINVAL, "row %d listed twice", ind_m[j]);
        seen[ind_m[j]] = 1;
    }
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    RET(to_dev(ctx, ctx->scr[23], ind_m, (size_t)K_m));
    LAUNCH(ctx, "stitch_add", k_stitch_add, dim3((unsigned)((ctx->stitch_T + 255) / 256), (unsigned)K_m), dim3(256), 0, job->dCraw.as<float>(), job->ldc,
           job->dAa.as<float>(), ctx->scr[23].as<int>(), ctx->stitch.as<float>(), ctx->stitch_ld, ctx->stitch_T);
    return 0;
}

int cnmfe_stitch_add(cnmfe_ctx *ctx, int32_t K_m, const int32_t *ind_m) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (!ctx->stitch_open) return fail(CNMFE_ESTATE, "cnmfe_stitch_begin has not been called");
    if (K_m == 0) return 0;
    if (K_m < 0 || !ind_m) return fail(CNMFE_EINVAL, "bad K_m / null ind_m");
    if (!ctx->last_t_valid || ctx->last_t_K != K_m || ctx->last_t_T != ctx->stitch_T)
        return fail(CNMFE_ESTATE, "no temporal result of %d rows x %lld frames on the device (the last cnmfe_hals_temporal / cnmfe_fast_temporal call)", K_m, (long long)ctx->stitch_T);
    std::vector<char> seen((size_t)ctx->stitch_K, 0);
    for (int32_t j = 0; j < K_m; ++j) {
        if (ind_m[j] < 0 || ind_m[j] >= ctx->stitch_K) return fail(CNMFE_EINVAL, "row %d outside the %d-row accumulator", ind_m[j], ctx->stitch_K);
        if (seen[ind_m[j]]) return fail(CNMFE_EINVAL, "row %d listed twice", ind_m[j]);
        seen[ind_m[j]] = 1;
    }
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    RET(to_dev(ctx, ctx->scr[23], ind_m, (size_t)K_m));
    LAUNCH(ctx, "stitch_add", k_stitch_add, dim3((unsigned)((ctx->stitch_T + 255) / 256), (unsigned)K_m), dim3(256), 0, ctx->last_craw.as<float>(), ctx->last_t_ldc,
           ctx->last_aa.as<float>(), ctx->scr[23].as<int>(), ctx->stitch.as<float>(), ctx->stitch_ld, ctx->stitch_T);
    // (ind_m went through the pinned arena; the call returns with the accumulation queued behind the patch's HALS sweeps)
    ctx->last_t_valid = false;
    return 0;
}

int cnmfe_stitch_buffer(cnmfe_ctx *ctx, float **dev_acc, int64_t *ld) {
    if (!ctx || !dev_acc || !ld) return fail(CNMFE_EINVAL, "null argument");
    if (!ctx->stitch_open) return fail(CNMFE_ESTATE, "cnmfe_stitch_begin has not been called");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    CK(hipStreamSynchronize(ctx->st()));                 // the caller's collective runs on ITS stream: everything added so far must have landed
    *dev_acc = ctx->stitch.as<float>(); *ld = ctx->stitch_ld;
    return 0;
}

// The same hand-over without the drain: the accumulator and the stream its additions are queued on.  A caller that enqueues its collective ON that stream (or
// orders its own stream behind it with an event) keeps the host out of the exchange: cnmfe_stitch_finish* then follows in stream order.
int cnmfe_stitch_buffer_stream(cnmfe_ctx *ctx, float **dev_acc, int64_t *ld, void **hip_stream) {
    if (!ctx || !dev_acc || !ld || !hip_stream) return fail(CNMFE_EINVAL, "null argument");
    if (!ctx->stitch_open) return fail(CNMFE_ESTATE, "cnmfe_stitch_begin has not been called");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    *dev_acc = ctx->stitch.as<float>(); *ld = ctx->stitch_ld; *hip_stream = (void *)ctx->st();      // (st(): the held-back small uploads go out first)
    return 0;
}

// host helper: a CSC matrix (ncol columns, rows sorted per column) from n (row, column, value) triplets in any order -- the assembly of the gathered rows of A
// (update_spatial_parallel.m:324-334: every rank's patches contribute disjoint rows) without a sort of the whole triplet list: counting pass per column, then
// an insertion sort of each column's few entries by row.  Duplicate (row, column) pairs are an error (the patches are disjoint).
int cnmfe_csc_from_triplets(int64_t n, const int32_t *rows, const int32_t *cols, const float *vals, int32_t ncol, int64_t nrow,
                            int64_t *out_colptr, int32_t *out_rowidx, float *out_val) {
    if (n < 0 || ncol < 0 || !out_colptr || (n > 0 && (!rows || !cols || !vals || !out_rowidx || !out_val))) return fail(CNMFE_EINVAL, "cnmfe_csc_from_triplets: null argument");
    for (int32_t k = 0; k <= ncol; ++k) out_colptr[k] = 0;
    for (int64_t e = 0; e < n; ++e) {
        if (cols[e] < 0 || cols[e] >= ncol || rows[e] < 0 || rows[e] >= nrow) return fail(CNMFE_EINVAL, "cnmfe_csc_from_triplets: entry %lld = (%d, %d) outside %lld x %d", (long long)e, rows[e], cols[e], (long long)nrow, ncol);
        ++out_colptr[cols[e] + 1];
    }
    for (int32_t k = 0; k < ncol; ++k) out_colptr[k + 1] += out_colptr[k];
    std::vector<int64_t> cur(out_colptr, out_colptr + ncol);
    for (int64_t e = 0; e < n; ++e) { const int64_t at = cur[cols[e]]++; out_rowidx[at] = rows[e]; out_val[at] = vals[e]; }
    std::vector<std::pair<int32_t, float>> tmp;
    for (int32_t k = 0; k < ncol; ++k) {                    // rows ascending within the column
        const int64_t a = out_colptr[k], b = out_colptr[k + 1];
        bool sorted = true;
        for (int64_t i = a + 1; i < b && sorted; ++i) sorted = out_rowidx[i - 1] <= out_rowidx[i];
        if (!sorted) {
            if (b - a <= 32) {                              // (a rank's part arrives sorted within the column: a few runs to merge)
                for (int64_t i = a + 1; i < b; ++i) {
                    const int32_t r = out_rowidx[i]; const float v = out_val[i];
                    int64_t j = i;
                    while (j > a && out_rowidx[j - 1] > r) { out_rowidx[j] = out_rowidx[j - 1]; out_val[j] = out_val[j - 1]; --j; }
                    out_rowidx[j] = r; out_val[j] = v;
                }
            } else {                                        // (a long unsorted column must not cost n^2)
                tmp.resize((size_t)(b - a));
                for (int64_t i = a; i < b; ++i) tmp[(size_t)(i - a)] = {out_rowidx[i], out_val[i]};
                std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, float> &x, const std::pair<int32_t, float> &y) { return x.first < y.first; });
                for (int64_t i = a; i < b; ++i) { out_rowidx[i] = tmp[(size_t)(i - a)].first; out_val[i] = tmp[(size_t)(i - a)].second; }
            }
        }
        for (int64_t i = a + 1; i < b; ++i) if (out_rowidx[i] == out_rowidx[i - 1]) return fail(CNMFE_EINVAL, "cnmfe_csc_from_triplets: entry (%d, %d) given twice", out_rowidx[i], k);
    }
    return 0;
}

// host helper: per footprint (column of the d x K CSC matrix A, pixels column-major in a d1-row image) the sum, the centre of mass (utilities/com.m:20-28, clamped
// to [0, d1] x [0, d2] as :26-28 do) and the second central moments of determine_search_location.m:73 -- sums over a column's entries in storage order, every
// product formed as (value * coordinate) [* coordinate] in double precision: the same numbers as NumPy's bincount formulation of the host mirror, which spent 4-5 ms
// here at 500 neurons.  Outputs: K doubles each; a footprint whose values sum to 0 gets empty[k] = 1 and moments over a unit denominator (:52).
int cnmfe_footprint_moments(int32_t K, int32_t d1, int32_t d2, const int64_t *colptr, const int32_t *rowidx, const float *val, double *s_out, uint8_t *empty,
                            double *cmx, double *cmy, double *vxx, double *vxy, double *vyy) {
#pragma clang fp contract(off)
    if (K < 0 || d1 < 1 || d2 < 1 || !colptr || !s_out || !empty || !cmx || !cmy || !vxx || !vxy || !vyy || (colptr[K] > 0 && (!rowidx || !val)))
        return fail(CNMFE_EINVAL, "cnmfe_footprint_moments: null argument");
    for (int32_t k = 0; k < K; ++k) {
        double s = 0.0, sx = 0.0, sy = 0.0;
        for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) {
            const double v = (double)val[e], x = (double)(rowidx[e] % d1 + 1), y = (double)(rowidx[e] / d1 + 1);
            s += v; sx += v * x; sy += v * y;
        }
        const bool em = s == 0.0;
        const double ss = em ? 1.0 : s;
        double mx = sx / ss, my = sy / ss;
        mx = std::min(std::max(mx, 0.0), (double)d1); my = std::min(std::max(my, 0.0), (double)d2);
        double xx = 0.0, xy = 0.0, yy = 0.0;
        for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) {
            const double v = (double)val[e], dx = (double)(rowidx[e] % d1 + 1) - mx, dy = (double)(rowidx[e] / d1 + 1) - my;
            xx += v * dx * dx; xy += v * dx * dy; yy += v * dy * dy;
        }
        s_out[k] = s; empty[k] = em ? 1 : 0; cmx[k] = mx; cmy[k] = my; vxx[k] = xx / ss; vxy[k] = xy / ss; vyy[k] = yy / ss;
    }
    return 0;
}

// host helper: the 'ellipse' search masks of determine_search_location.m:76-100 for K neurons whose centres of mass, principal axes and clamped axis
// variances the caller has computed (com.m:20-28, determine_search_location.m:73-82: moments + a 2 x 2 eig, K small problems that stay in the caller's
// linear algebra).  Pixel (r, c), 1-based, belongs to mask k iff
//     sqrt(((r - cmx) V00 + (c - cmy) V10)^2 / d11 + ((r - cmx) V01 + (c - cmy) V11)^2 / d22) <= dist          (:84, evaluated exactly in this order)
// within the (2R+1)^2 window around (floor(cmx), floor(cmy)) and the image.  The set is convex, so every image column holds one run of rows: its ends are
// estimated from the quadratic and then settled with the exact expression above (the estimate only decides where the exact tests start), which makes the
// result identical to the exhaustive evaluation at O(window) instead of O(window^2) tests per neuron -- 127 k mask entries for 500 neurons in ~0.2 ms, where
// the vectorised NumPy formulation took 7-20 ms on the host's critical path between the ring fit and the spatial update.
// vk = (V00, V10, V01, V11) per neuron; empty[k] != 0: no mask (:102-104).  out_colptr[K + 1] is always written; out_rowidx (global 0-based pixel
// (c - 1) d1 + (r - 1), ascending per neuron) only if cap >= the total; *nnz_out = the total.
int cnmfe_search_ellipse(int32_t K, int32_t d1, int32_t d2, const double *cmx, const double *cmy, const double *vk, const double *d11, const double *d22,
                         const uint8_t *empty, double dist, int32_t R, int64_t cap, int64_t *out_colptr, int32_t *out_rowidx, int64_t *nnz_out) {
#pragma clang fp contract(off)
    if (K < 0 || d1 < 1 || d2 < 1 || R < 0 || !out_colptr || !nnz_out || (K > 0 && (!cmx || !cmy || !vk || !d11 || !d22)))
        return fail(CNMFE_EINVAL, "cnmfe_search_ellipse: null argument");
    const int W = 2 * R + 1;
    std::vector<int32_t> lo((size_t)std::max(1, K) * W), hi((size_t)std::max(1, K) * W);      // per neuron and window column: first / last row (1-based), lo > hi: none
    int64_t total = 0;
    out_colptr[0] = 0;
    for (int32_t k = 0; k < K; ++k) {
        int64_t nk = 0;
        const double mx = cmx[k], my = cmy[k], v00 = vk[4 * k], v10 = vk[4 * k + 1], v01 = vk[4 * k + 2], v11 = vk[4 * k + 3], a11 = d11[k], a22 = d22[k];
        const double fx = std::floor(mx), fy = std::floor(my);
        auto inside = [&](double r, double ey) -> bool {
            const double ex = r - mx;
            const double p1 = ex * v00 + ey * v10, p2 = ex * v01 + ey * v11;
            return std::sqrt(p1 * p1 / a11 + p2 * p2 / a22) <= dist;
        };
        // the quadratic form a ex^2 + 2 b ex ey + c ey^2 <= dist^2 (estimates only)
        const double qa = v00 * v00 / a11 + v01 * v01 / a22, qb = v00 * v10 / a11 + v01 * v11 / a22, qc = v10 * v10 / a11 + v11 * v11 / a22;
        for (int j = 0; j < W; ++j) {
            int32_t &l = lo[(size_t)k * W + j], &h = hi[(size_t)k * W + j];
            l = 1; h = 0;
            if (empty && empty[k]) continue;
            const double c = fy + (double)(j - R);
            if (c < 1.0 || c > (double)d2) continue;
            const double wmin = std::max(fx - (double)R, 1.0), wmax = std::min(fx + (double)R, (double)d1);
            if (wmin > wmax) continue;
            const double ey = c - my;
            double ctr = mx, half = 0.0;
            if (qa > 0.0 && std::isfinite(qa)) {
                ctr = mx - qb * ey / qa;
                const double disc = qb * ey * qb * ey - qa * (qc * ey * ey - dist * dist);
                half = disc > 0.0 ? std::sqrt(disc) / qa : 0.0;
            }
            if (!std::isfinite(ctr)) ctr = mx;
            if (!std::isfinite(half)) half = (double)W;
            // a pixel of the run: the one nearest the vertex, or a neighbour (a run narrower than a pixel contains one of them or nothing)
            double seed = std::min(std::max(std::floor(ctr + 0.5), wmin), wmax), t = 0.0;
            bool found = false;
            for (int o = 0; o < 3 && !found; ++o) {
                const double r = seed + (o == 0 ? 0.0 : (o == 1 ? -1.0 : 1.0));
                if (r >= wmin && r <= wmax && inside(r, ey)) { t = r; found = true; }
            }
            if (!found) continue;
            double rl = std::min(t, std::max(std::ceil(ctr - half), wmin)), rh = std::max(t, std::min(std::floor(ctr + half), wmax));
            while (rl < t && !inside(rl, ey)) rl += 1.0;
            while (rl - 1.0 >= wmin && inside(rl - 1.0, ey)) rl -= 1.0;
            while (rh > t && !inside(rh, ey)) rh -= 1.0;
            while (rh + 1.0 <= wmax && inside(rh + 1.0, ey)) rh += 1.0;
            l = (int32_t)rl; h = (int32_t)rh;
            nk += (int64_t)(h - l + 1);
        }
        total += nk;
        out_colptr[k + 1] = total;
    }
    *nnz_out = total;
    if (!out_rowidx || cap < total) return 0;
    int64_t at = 0;
    for (int32_t k = 0; k < K; ++k) {
        const double fy = std::floor(cmy[k]);
        for (int j = 0; j < W; ++j) {
            const int32_t l = lo[(size_t)k * W + j], h = hi[(size_t)k * W + j];
            if (l > h) continue;
            const int64_t c = (int64_t)fy + (j - R);
            for (int32_t r = l; r <= h; ++r) out_rowidx[at++] = (int32_t)((c - 1) * d1 + (r - 1));
        }
    }
    return 0;
}

int cnmfe_stitch_finish(cnmfe_ctx *ctx, int subtract_min, float *C_raw_out, int c_order) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    return stitch_finish_one(ctx, subtract_min, C_raw_out, c_order);
}

int cnmfe_stitch_finish_async(cnmfe_ctx *ctx, int subtract_min, float *C_raw_pinned) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (!C_raw_pinned) return fail(CNMFE_EINVAL, "null C_raw_pinned");
    return stitch_finish_one(ctx, subtract_min, C_raw_pinned, CNMFE_ROWMAJOR, true);
}

int cnmfe_stitch_dims(cnmfe_ctx *ctx, int32_t *K, int64_t *T) {
    if (!ctx || !K || !T) return fail(CNMFE_EINVAL, "null context / K / T");
    if (!ctx->stitch_open) return fail(CNMFE_ESTATE, "cnmfe_stitch_begin has not been called");
    *K = ctx->stitch_K; *T = ctx->stitch_T;
    return 0;
}

int cnmfe_stitch_wait(cnmfe_ctx *ctx) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (ctx->ev_copy_done && ctx->copy_stream) CK(hipStreamSynchronize(ctx->copy_stream));
    for (auto &g : ctx->copy_gens) ctx->copy_ev_pool.push_back(g.second);
    ctx->copy_gens.clear();
    return 0;
}

int cnmfe_copy_generation(cnmfe_ctx *ctx, int64_t *gen) {
    if (!ctx || !gen) return fail(CNMFE_EINVAL, "null context / gen");
    *gen = ctx->copy_gen;
    return 0;
}

int cnmfe_copy_wait(cnmfe_ctx *ctx, int64_t gen) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (gen > ctx->copy_gen) return fail(CNMFE_ESTATE, "download batch %lld has not been queued (last: %lld)", (long long)gen, (long long)ctx->copy_gen);
    size_t n = 0;
    while (n < ctx->copy_gens.size() && ctx->copy_gens[n].first <= gen) ++n;       // (ascending; batches complete in order)
    if (n == 0) return 0;                                  // already known to be complete
    CK(hipEventSynchronize(ctx->copy_gens[n - 1].second));
    for (size_t i = 0; i < n; ++i) ctx->copy_ev_pool.push_back(ctx->copy_gens[i].second);
    ctx->copy_gens.erase(ctx->copy_gens.begin(), ctx->copy_gens.begin() + n);
    return 0;
}

void *cnmfe_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(CNMFE_EHIP, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void cnmfe_host_free(void *p) { if (p) (void)hipHostFree(p); }

// a CSC matrix without its stored zeros (and, with `keep`, without the entries whose flag is 0): what MATLAB's sparse assignment does implicitly
int cnmfe_csc_drop_zeros(int32_t ncol, const int64_t *colptr, const int32_t *rowidx, const float *val, const uint8_t *keep,
                         int64_t *out_colptr, int32_t *out_rowidx, float *out_val, int64_t *nnz_out) {
    if (ncol < 0 || !colptr || !out_colptr || !nnz_out) return fail(CNMFE_EINVAL, "cnmfe_csc_drop_zeros: null argument");
    const int64_t nnz = colptr[ncol];
    if (nnz > 0 && (!rowidx || !val || !out_rowidx || !out_val)) return fail(CNMFE_EINVAL, "cnmfe_csc_drop_zeros: null argument");
    int64_t n = 0;
    out_colptr[0] = 0;
    for (int32_t k = 0; k < ncol; ++k) {
        if (keep) { for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) if (val[e] != 0.f && keep[e]) { out_rowidx[n] = rowidx[e]; out_val[n] = val[e]; ++n; } }
        else      { for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) if (val[e] != 0.f) { out_rowidx[n] = rowidx[e]; out_val[n] = val[e]; ++n; } }
        out_colptr[k + 1] = n;
    }
    *nnz_out = n;
    return 0;
}

// A(mask, cols) of a CSC matrix (what the reference writes as sparse indexing, e.g. update_spatial_parallel.m:87-91,96-97): host code, no device involved.
int cnmfe_csc_select_rows(const int64_t *colptr, const int32_t *rowidx, const float *val, const int32_t *lut, int64_t ncand, const int64_t *cand,
                          int keep_all, int64_t cap, int64_t *out_ind, int64_t *out_colptr, int32_t *out_rowidx, float *out_val, int64_t *nkept) {
    if (!colptr || !rowidx || !val || !lut || (ncand > 0 && !cand) || !out_ind || !out_colptr || !nkept || (cap > 0 && (!out_rowidx || !out_val)))
        return fail(CNMFE_EINVAL, "cnmfe_csc_select_rows: null argument");
    int64_t n = 0, nc = 0;
    out_colptr[0] = 0;
    for (int64_t j = 0; j < ncand; ++j) {
        const int64_t k = cand[j];
        if (j > 0 && k <= cand[j - 1]) return fail(CNMFE_EINVAL, "cnmfe_csc_select_rows: columns must be ascending");
        const int64_t n0 = n;
        double s = 0.0;
        for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) {
            const int32_t loc = lut[rowidx[e]];
            if (loc < 0) continue;
            if (n >= cap) return fail(CNMFE_EINVAL, "cnmfe_csc_select_rows: output capacity %lld exceeded", (long long)cap);
            out_rowidx[n] = loc; out_val[n] = val[e]; s += (double)val[e]; ++n;
        }
        if (keep_all || s > 0.0) { out_ind[nc] = k; out_colptr[++nc] = n; }
        else n = n0;                                       // sum(A(mask, k)) > 0 fails: the column is not selected
    }
    *nkept = nc;
    return 0;
}

// rows_of twice in one pass (sources2d.py, update_temporal_parallel: `ind = find(sum(A(block, :), 1) > 0)`, A(block, ind) and A(patch, ind) of one patch): the columns
// are the candidates whose BLOCK entries sum to > 0, the second matrix holds the same columns' PATCH entries (a patch's pixels are block pixels)
int cnmfe_csc_select_block_patch(const int64_t *colptr, const int32_t *rowidx, const float *val, const int32_t *lut_block, const int32_t *lut_patch, int64_t ncand,
                                 const int64_t *cand, int64_t cap, int64_t *out_ind, int64_t *blk_colptr, int32_t *blk_rowidx, float *blk_val,
                                 int64_t *pat_colptr, int32_t *pat_rowidx, float *pat_val, int64_t *nkept) {
    if (!colptr || !rowidx || !val || !lut_block || !lut_patch || (ncand > 0 && !cand) || !out_ind || !blk_colptr || !pat_colptr || !nkept ||
        (cap > 0 && (!blk_rowidx || !blk_val || !pat_rowidx || !pat_val))) return fail(CNMFE_EINVAL, "cnmfe_csc_select_block_patch: null argument");
    int64_t nb = 0, np_ = 0, nc = 0;
    blk_colptr[0] = 0; pat_colptr[0] = 0;
    for (int64_t j = 0; j < ncand; ++j) {
        const int64_t k = cand[j];
        if (j > 0 && k <= cand[j - 1]) return fail(CNMFE_EINVAL, "cnmfe_csc_select_block_patch: columns must be ascending");
        const int64_t nb0 = nb, np0 = np_;
        double s = 0.0;
        for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) {
            const int32_t lb = lut_block[rowidx[e]];
            if (lb < 0) continue;
            if (nb >= cap) return fail(CNMFE_EINVAL, "cnmfe_csc_select_block_patch: output capacity %lld exceeded", (long long)cap);
            blk_rowidx[nb] = lb; blk_val[nb] = val[e]; s += (double)val[e]; ++nb;
            const int32_t lp = lut_patch[rowidx[e]];
            if (lp >= 0) { pat_rowidx[np_] = lp; pat_val[np_] = val[e]; ++np_; }
        }
        if (s > 0.0) { out_ind[nc] = k; ++nc; blk_colptr[nc] = nb; pat_colptr[nc] = np_; }
        else { nb = nb0; np_ = np0; }
    }
    *nkept = nc;
    return 0;
}

// per non-empty column of a sorted CSC footprint matrix over a d1-row image: its index and the bounding box of its entries (image rows / columns, 0-based)
int cnmfe_csc_bbox(int32_t K, int32_t d1, const int64_t *colptr, const int32_t *rowidx, int64_t *nz, int32_t *rmin, int32_t *rmax, int32_t *cmin, int32_t *cmax, int64_t *nnz_cols) {
    if (K < 0 || d1 <= 0 || !colptr || (colptr[K] > 0 && !rowidx) || !nnz_cols || (K > 0 && (!nz || !rmin || !rmax || !cmin || !cmax))) return fail(CNMFE_EINVAL, "cnmfe_csc_bbox: bad argument");
    int64_t n = 0;
    for (int32_t k = 0; k < K; ++k) {
        const int64_t a = colptr[k], b = colptr[k + 1];
        if (b <= a) continue;
        int32_t lo = INT32_MAX, hi = -1;
        for (int64_t e = a; e < b; ++e) { const int32_t r = rowidx[e] % d1; lo = std::min(lo, r); hi = std::max(hi, r); }
        nz[n] = k; rmin[n] = lo; rmax[n] = hi; cmin[n] = rowidx[a] / d1; cmax[n] = rowidx[b - 1] / d1;      // (entries ascend in pixel index = image column major)
        ++n;
    }
    *nnz_cols = n;
    return 0;
}

int cnmfe_stitch_temporal(cnmfe_ctx *const *ctxs, int n, int subtract_min, float *C_raw_out, int c_order) {
    if (!ctxs || n <= 0) return fail(CNMFE_EINVAL, "no contexts");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->stitch_open) return fail(CNMFE_ESTATE, "context %d: cnmfe_stitch_begin has not been called", i);
        if (ctxs[i]->stitch_K != ctxs[0]->stitch_K || ctxs[i]->stitch_T != ctxs[0]->stitch_T) return fail(CNMFE_EINVAL, "context %d accumulates a different K x T", i);
        for (int j = 0; j < i; ++j) if (ctxs[j]->device == ctxs[i]->device) return fail(CNMFE_EINVAL, "contexts %d and %d share GPU %d (one context per GPU)", j, i, ctxs[i]->device);
    }
    // n == 1 still goes through RCCL when the communicator exists or CNMFE_STITCH_RCCL=1 asks for it (a 1-GPU box exercising the multi-GPU path)
    const char *force = getenv("CNMFE_STITCH_RCCL");
    if (n > 1 || (force && force[0] == '1')) {
        RET(g_rccl.load());
        bool have = true;
        for (int i = 0; i < n; ++i) have = have && ctxs[i]->rccl_comm && ctxs[i]->rccl_n == n && ctxs[i]->rccl_rank == i;
        if (!have) {
            std::vector<int> devs(n); std::vector<void *> comms(n, nullptr);
            for (int i = 0; i < n; ++i) { devs[i] = ctxs[i]->device; if (ctxs[i]->rccl_comm) { g_rccl.CommDestroy(ctxs[i]->rccl_comm); ctxs[i]->rccl_comm = nullptr; } }
            const int rc = g_rccl.CommInitAll(comms.data(), n, devs.data());
            if (rc != 0) return fail(CNMFE_EHIP, "ncclCommInitAll over %d GPU(s) failed: %s", n, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
            for (int i = 0; i < n; ++i) { ctxs[i]->rccl_comm = comms[i]; ctxs[i]->rccl_n = n; ctxs[i]->rccl_rank = i; }
        }
        const size_t count = (size_t)ctxs[0]->stitch_K * (size_t)ctxs[0]->stitch_ld;
        if (count) {
            int rc = g_rccl.GroupStart();
            for (int i = 0; i < n && rc == 0; ++i) {
                CK(hipSetDevice(ctxs[i]->device));
                rc = g_rccl.AllReduce(ctxs[i]->stitch.p, ctxs[i]->stitch.p, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, ctxs[i]->rccl_comm, ctxs[i]->st());
            }
            const int rc2 = g_rccl.GroupEnd();
            if (rc != 0 || rc2 != 0) return fail(CNMFE_EHIP, "ncclAllReduce of the stitch accumulator failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc ? rc : rc2) : "?");
        }
    }
    for (int i = 0; i < n; ++i) RET(stitch_finish_one(ctxs[i], subtract_min, i == 0 ? C_raw_out : nullptr, c_order));
    for (int i = 0; i < n; ++i) { CK(hipSetDevice(ctxs[i]->device)); CK(hipStreamSynchronize(ctxs[i]->st())); }
    return 0;
}

int cnmfe_traces_bind(cnmfe_ctx *ctx, int32_t K, int64_t T, const float *C, int c_order) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    ctx->bound_valid = false; ++ctx->bound_gen;
    if (K == 0 || !C) return 0;                            // unbind
    if (K < 0 || T <= 0) return fail(CNMFE_EINVAL, "bad K / T");
    if (c_order != CNMFE_ROWMAJOR && c_order != CNMFE_COLMAJOR) return fail(CNMFE_EINVAL, "bad c_order");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    int64_t ldc;
    if (ctx->copy_pending) { CK(hipStreamWaitEvent(ctx->st(), ctx->ev_copy_done, 0)); ctx->copy_pending = false; }   // a lazy download may still read `bound` (as stitch_finish_one)
    RET(upload_traces(ctx, ctx->bound, C, K, T, c_order, &ldc));
    CK(hipStreamSynchronize(ctx->st()));
    ctx->bound_K = K; ctx->bound_T = T; ctx->bound_order = c_order; ctx->bound_valid = true;
    return 0;
}

int cnmfe_deconv_temporal(cnmfe_ctx *ctx, int32_t K, int64_t T, float *C_raw, int c_order, const cnmfe_deconv_opts *opts,
                          float *C_out, float *S_out, float *kernel_pars_out, float *sn_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (K < 0 || T <= 0) return fail(CNMFE_EINVAL, "bad K / T");
    if (K == 0) return 0;
    if (!C_raw || !opts || !C_out) return fail(CNMFE_EINVAL, "null C_raw / opts / C_out");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    return deconv_all_run(ctx, K, T, C_raw, c_order, opts, C_out, S_out, kernel_pars_out, sn_out);
}

int cnmfe_deconv_temporal_bound(cnmfe_ctx *ctx, const cnmfe_deconv_opts *opts, float *C_out, float *C_raw_out, float *S_out, float *kernel_pars_out, float *sn_out) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (!opts) return fail(CNMFE_EINVAL, "null deconvolution options");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    return deconv_bound_run(ctx, opts, C_out, C_raw_out, S_out, kernel_pars_out, sn_out);
}

int cnmfe_post_process_spatial(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *A_colptr,
                               const int32_t *A_rowidx, const float *A_val, uint8_t *keep) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    if (d1 <= 0 || d2 <= 0) return fail(CNMFE_EINVAL, "bad dims");
    RET(check_csc("A", K, (int64_t)d1 * d2, A_colptr, A_rowidx));
    if (K == 0) return 0;
    if ((!A_val && A_colptr[K] > 0) || !keep) return fail(CNMFE_EINVAL, "null A_val / keep");
    CK(hipSetDevice(ctx->device));
    GlobalScope gs_(ctx); if (gs_.rc) return gs_.rc;          // (lanes: lane 0, behind the other lanes; they wait for what this queues)
    return postproc_run(ctx, d1, d2, K, A_colptr, A_rowidx, A_val, keep);
}

int cnmfe_profile_enable(cnmfe_ctx *ctx, int on) { if (!ctx) return fail(CNMFE_EINVAL, "null context"); ctx->prof.drain(); ctx->prof.on = on == 2 ? 2 : (on != 0); return 0; }
int cnmfe_profile_reset(cnmfe_ctx *ctx) { if (!ctx) return fail(CNMFE_EINVAL, "null context"); ctx->prof.reset(); return 0; }
int cnmfe_profile_count(cnmfe_ctx *ctx) { if (!ctx) return fail(CNMFE_EINVAL, "null context"); ctx->prof.drain(); return (int)ctx->prof.names.size(); }
int cnmfe_profile_get(cnmfe_ctx *ctx, int i, char *name, int cap, double *total_ms, int64_t *calls) {
    if (!ctx) return fail(CNMFE_EINVAL, "null context");
    ctx->prof.drain();
    if (i < 0 || i >= (int)ctx->prof.names.size()) return fail(CNMFE_EINVAL, "profile index %d out of range", i);
    if (name && cap > 0) { strncpy(name, ctx->prof.names[i].c_str(), cap - 1); name[cap - 1] = 0; }
    if (total_ms) *total_ms = ctx->prof.total_ms[i];
    if (calls) *calls = ctx->prof.calls[i];
    return 0;
}

}  // extern "C"
