// Shared internals of the CNMF-E HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <string>
#include <vector>
#include <cstring>
#include <chrono>
#include <cmath>
#include <map>
#include <algorithm>
#include "../../include/cnmfe.h"

namespace cnmfe {

// a 16-byte load of data that is read ONCE per pass (the digit planes of the video in k_win_proj_i8): non-temporal, so the stream does not evict the trace planes every
// workgroup re-reads (1.97 -> 1.94 ms at H, three runs each on one lease; the fp32 stream of k_vp_proj_b showed no difference and keeps plain loads)
typedef unsigned nt_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const uint4 *p) {
    const nt_u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}


extern thread_local char g_err[1024];
inline int fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return cnmfe::fail(CNMFE_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); } while (0)
#define RET(x) do { int r_ = (x); if (r_ != 0) return r_; } while (0)

void pin_flush_all();         // api.hip: enqueue the small uploads every context still holds back (see cnmfe_ctx::st)
void pin_flush_range(const void *p, size_t bytes);   // ... those into [p, p + bytes) -- before that device memory is freed (only its owner's context holds any)

// ---- owned device buffer -----------------------------------------------------
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    ~DevBuf() { if (p) { pin_flush_range(p, cap); (void)hipFree(p); } }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        // a buffer that GROWS gets half as much again: index lists and value arrays follow nnz(A) / nnz(IND), which creep up over the first iterations, and every
        // hipFree drains the device -- 0.8-0.9 ms of the host blocked inside an upload while the fit's kernels ran (host_trace, one rank's share of c4)
        const size_t grown = (p && bytes < (size_t(64) << 20)) ? std::max(bytes, cap + cap / 2) : bytes;       // (small buffers only: a table or a video that grows is not over-allocated)
        static const bool trace_alloc_ = getenv("CNMFE_TRACE_ALLOC") != nullptr;          // (diagnostic: every re-allocation of a grown buffer on stderr -- hipFree drains the device)
        if (trace_alloc_ && p) fprintf(stderr, "[alloc] grow %zu -> %zu bytes (hipFree + hipMalloc)\n", cap, grown);
        if (p) { pin_flush_range(p, cap); (void)hipFree(p); p = nullptr; cap = 0; }     // (an upload into the old allocation may still be held back)
        size_t want = (grown + 255) & ~size_t(255);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(CNMFE_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        cap = want; return 0;
    }
    // for buffers that change hands (swap) between the context's scratch and the patches: every one of them grows to the largest request seen (`hw`),
    // otherwise a buffer sized for one patch's traces is re-allocated -- hipFree drains the device -- whenever it lands under a patch with more
    int ensure_hw(size_t bytes, size_t &hw) { if (bytes > hw) hw = bytes; return ensure(hw); }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); }
};

// ---- per-kernel event timing ---------------------------------------------------
struct Profiler {
    int on = 0;                        // 0 off, 1 every kernel, 2 only the kernels a roofline is quoted for (a pair of events costs a few microseconds of
                                       // host AND device time per launch: thousands of small launches per iteration feel it, cnmfe_profile_enable)
    static bool selected(const char *n) {
        static const char *const sel[] = {"residual_r1", "bg_ring_solve", "bg_win_proj", "bg_cov_correct", "spatial_proj_U", "temporal_proj_U"};
        for (const char *s : sel) if (!strcmp(n, s)) return true;
        return false;
    }
    bool want(const char *n) const { return on == 1 || (on == 2 && selected(n)); }
    struct Rec { hipEvent_t a, b; int id; };
    std::vector<std::string> names;
    std::vector<double> total_ms; std::vector<int64_t> calls;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    int id_of(const char *n) {
        for (size_t i = 0; i < names.size(); ++i) if (names[i] == n) return (int)i;
        names.push_back(n); total_ms.push_back(0); calls.push_back(0); return (int)names.size() - 1;
    }
    hipEvent_t ev() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
                      hipEvent_t e; (void)hipEventCreate(&e); return e; }
    void begin(const char *n, hipStream_t s, Rec &r) { r.id = id_of(n); r.a = ev(); r.b = ev(); (void)hipEventRecord(r.a, s); }
    void end(hipStream_t s, Rec &r) { (void)hipEventRecord(r.b, s); pending.push_back(r); }
    void drain() {
        for (auto &r : pending) {
            (void)hipEventSynchronize(r.b); float ms = 0; (void)hipEventElapsedTime(&ms, r.a, r.b);
            total_ms[r.id] += ms; calls[r.id] += 1; pool.push_back(r.a); pool.push_back(r.b);
        }
        pending.clear();
    }
    void reset() { drain(); for (auto &t : total_ms) t = 0; for (auto &c : calls) c = 0; }
    ~Profiler() { drain(); for (auto e : pool) (void)hipEventDestroy(e); }
};

// launch wrapper: LAUNCH(ctx, "name", kernel, grid, block, shmem, args...)
#define LAUNCH(ctx, name, kern, grid, block, shmem, ...) do { \
    cnmfe::Profiler::Rec pr_; bool pon_ = (ctx)->prof.on && (ctx)->prof.want(name); \
    const bool ltr_ = (ctx)->trace_level >= 2; \
    const auto lt0_ = ltr_ ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point(); \
    hipStream_t lst_ = (ctx)->st(); \
    if (pon_) (ctx)->prof.begin(name, lst_, pr_); \
    const auto lt1_ = ltr_ ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point(); \
    hipLaunchKernelGGL(kern, grid, block, shmem, lst_, __VA_ARGS__); \
    if (ltr_) { const auto lt2_ = std::chrono::steady_clock::now(); const double la_ = std::chrono::duration<double, std::milli>(lt1_ - lt0_).count(), lb_ = std::chrono::duration<double, std::milli>(lt2_ - lt1_).count(); \
        if (la_ + lb_ > 2.0) fprintf(stderr, "[host_trace] launch %s: flush of held-back uploads %.2f ms, launch call %.2f ms\n", name, la_, lb_); } \
    if (pon_) (ctx)->prof.end(lst_, pr_); \
    hipError_t le_ = hipGetLastError(); \
    if (le_ != hipSuccess) return cnmfe::fail(CNMFE_EHIP, "launch %s failed: %s", name, hipGetErrorString(le_)); \
} while (0)

constexpr int PMAX_RING = 128;    // max ring neighbours supported (r=18 -> 120)
constexpr int BLK = 16;           // 16x16-pixel covariance blocks
constexpr int BLKPX = BLK * BLK;

// ---- sparse helpers (host) ------------------------------------------------------
struct HostCSR { std::vector<int32_t> rowptr;   /* nnz < 2^31 everywhere (checked by the callers): the device wants 32-bit row pointers anyway */ std::vector<int32_t> col; std::vector<float> val; std::vector<int32_t> src; };
// CSC (ncol columns, nrow rows) -> CSR with `src` = index into the CSC arrays
inline void csc_to_csr(int64_t nrow, int32_t ncol, const int64_t *colptr, const int32_t *rowidx, const float *val, HostCSR &out) {
    int64_t nnz = colptr[ncol];
    out.rowptr.assign(nrow + 1, 0); out.col.resize(nnz); out.val.resize(nnz); out.src.resize(nnz);
    for (int64_t e = 0; e < nnz; ++e) out.rowptr[rowidx[e] + 1]++;
    for (int64_t r = 0; r < nrow; ++r) out.rowptr[r + 1] += out.rowptr[r];
    std::vector<int32_t> cur(out.rowptr.begin(), out.rowptr.end() - 1);
    for (int32_t k = 0; k < ncol; ++k)
        for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) {
            int32_t pos = cur[rowidx[e]]++; out.col[pos] = k; out.val[pos] = val ? val[e] : 1.0f; out.src[pos] = (int32_t)e;
        }
}

// ---- per-patch resident state ----------------------------------------------------
struct Patch {
    int32_t prect[4], brect[4];        // 1-based inclusive [r0 r1 c0 c1]
    int32_t d1 = 0, d2 = 0;
    int32_t nr = 0, nc = 0, nr_b = 0, nc_b = 0;
    int32_t roff = 0, coff = 0;        // patch origin inside the block (0-based)
    int64_t T = 0, d = 0, d_b = 0;
    DevBuf Y;                          // upload staging: T x d_b fp32, frame-major; released once Yc4 is built
    DevBuf Yc4;                        // the RESIDENT video: centred (Y - Ymean), 4-frame interleaved: [ceil(T/4)][d_b] float4
    int64_t Tc = 0;                    // ceil(T/4)
    DevBuf ymean_d;                    // d_b double
    DevBuf ymean_f;                    // d_b float
    bool ymean_valid = false;
    int64_t frames_uploaded = 0;       // distinct frames that have arrived
    std::vector<uint8_t> frame_seen;   // T flags: a frame counts once, overlapping or repeated chunks are rejected (cnmfe_upload_block)
    // ring
    int32_t radius = 0, p = 0;         // p ring offsets
    std::vector<int32_t> dr, dc;       // ring offsets, (dc, dr)-sorted == MATLAB find() order
    DevBuf ring_dr, ring_dc;           // int32[p]
    DevBuf W;                          // p x d fp32, offset-major; 0 where the neighbour is outside the FOV
    DevBuf b0;                         // d fp64 (kept in double on the device; the ABI converts)
    bool ring_ready = false;
    DevBuf sn_b;                       // d_b fp32 noise levels of the block pixels (cnmfe_set_noise): only the outlier branch of the ring fit reads them
    bool sn_ready = false;
    // the background-subtracted video of this patch, Ysig4[ceil(T/4)][d] float4, as left by the last cnmfe_residual; valid until the
    // video, W or b0 change.  It stays resident per patch: a further cnmfe_residual under the same W, b0 differs from it only by the
    // footprint term (W*A)(C - mean C), whose applied instance is kept beside it (ELL rows + centred traces) -- see residual_run
    DevBuf ysig;
    bool ysig_valid = false;
    // sweep-free ("virtual") residual, round 4: nothing between two background fits needs Ysig itself, only its projections.  After a fit cnmfe_residual only
    // RECORDS the request (ysig_valid and ysig_virtual set, the footprint term pending as before): the spatial update takes Ysig C' out of the table
    // P = Yc Cc' below (U = P - W P, vproj.hip), the temporal update projects the centred video through B = A - W'A.  Whoever needs Ysig itself (GetSn,
    // compute_RSS, an export, fast_temporal ...) calls residual_realize first, which runs the sweep.
    bool ysig_virtual = false;
    // P(j,k) = sum_t Yc_j(t) Cc_k(t) in fp64 per 16x16 block of the block region and list slot: pt_tab[(pt_lp[b] + slot) * 256 + lp_of(pixel)], pt_slot[b * pt_K + k]
    // = slot or -1.  The ring fit's window projection computes exactly this (all frames, the same centred traces) and leaves it here; otherwise the
    // spatial update builds it.  Columns are rows pt_rows[] of the bound trace matrix of generation pt_gen (-1: some other matrix -- not reusable).
    DevBuf pt_tab, pt_lp, pt_slot;
    bool pt_valid = false; int32_t pt_K = 0; int64_t pt_gen = -1;
    std::vector<int32_t> pt_rows; std::vector<int> pt_lp_h; std::vector<short> pt_slot_h;
    bool res_ac = false; int res_kind = 0; int64_t res_ldc = 0;    // res_kind: who wrote Ysig and the term beside it: 1 = cnmfe_residual, 2 = cnmfe_residual_ssub (0: no term kept)
    DevBuf resCnt, resK, resV, resCc, resCm;              // resCm: the means the centred traces were taken about (fp64, per trace)
    // a footprint term asked for by the last cnmfe_residual but not yet folded into Ysig: cnmfe_hals_temporal only needs A' Ysig and adds
    // A' (W A)(C - mean C) algebraically (factor.hip), every other consumer calls residual_materialize first
    bool pend = false, pend_ac = false; int64_t pend_ldc = 0; int32_t pend_K = 0, res_K = 0;
    DevBuf pendCnt, pendK, pendV, pendCc, pendCm;
    // incremental ring regression (bg.hip): the block-pair covariance table and row sums of the centred VIDEO (no footprints subtracted),
    // valid for one frame stride until the video changes
    DevBuf cov_base, rowsum_base;
    bool base_valid = false, derived = false;             // derived: a low-resolution patch of bg_ssub > 1 (its video is rebuilt every call)
    int base_kstride = 0;
    // a second table of the same video at ANOTHER frame stride: the stride follows pmax (fit_ring_model.m:84-87: k = floor(T / min(T, 100 pmax))), and a
    // recording whose T / (100 pmax) sits at an integer -- T = 20000 with pmax around 66 -- alternates between two strides from fit to fit; one kept table
    // meant a full fp64 rebuild (10-15 ms per patch) on every flip.  Allocated only when a second stride turns up.
    DevBuf cov_base_alt, rowsum_base_alt; bool base_alt_valid = false; int base_alt_kstride = 0;
    // the video's normal equations per patch pixel, packed in the ring solve's register-tile order (ring_solve_packed.hpp): gathered once from cov_base
    // (and once more for a second frame stride), 43 KB per pixel at p = 96
    DevBuf sys, sys_alt; bool sys_valid = false, sys_alt_valid = false;
    // round 6 (ring_solve_inv.hpp): the explicit inverses of those systems at a ridge lam0 per pixel (k_ring_inverse: built behind the patch's first packed fit,
    // the bytes of `sys` once more), the ridge every fit leaves per pixel, and the list of pixels a fit's fast path (k_ring_apply) left to k_ring_solve6
    // (kinv_list: [d] pixels, then 4 counters).  They belong to `sys`: invalid whenever it is
    DevBuf kinv, kinv_lam, kinv_list; bool kinv_valid = false, kinv_lam_valid = false; int kinv_fits = 0;
    // the centred video as 32-bit fixed-point digit planes [blk][frame/16][plane][256 px] x 16 B + per-pixel scales (gram_i8.hpp), kept for the fits' window projection
    // (win_proj_i8.hpp) when the memory allows: frame stride 1 only
    DevBuf dig, dig_sc; int64_t dig_T16 = 0; bool dig_valid = false;
    DevBuf yt4; bool yt4_valid = false;                   // vproj.hip: the centred video tiled by 16 x 16 block (k_tile_video), for the temporal projection
    DevBuf digp; bool digp_valid = false;                 // vproj_i8.hpp: the digit planes once more, pixel-major (k_dig_pixmajor), for the temporal projection on the int8 pipe
    // bg_ssub > 1 (ssub.hip, round 5: the sweep-free residual of res_kind 2): the low-resolution residual patch and the factor of the last cnmfe_residual_ssub, and
    // imresize's bicubic UPSAMPLING taps of the block region's rows / columns (low-resolution index + weight, ss_Pr / ss_Pc per row / column) on the host and the
    // device, their ranges per row / column (made monotone: ss_rlo[r] <= every tap index of the rows >= r, ss_rhi[r] >= those of the rows <= r), and the
    // TRANSPOSED taps (per low-resolution row / column the block-region rows / columns it feeds, CSR) for up' A of the temporal projection
    int ss_res = -1, ss_ssub = 0, ss_Pr = 0, ss_Pc = 0, ss_d1s = 0, ss_d2s = 0;
    bool ss_taps = false;
    std::vector<int> ss_tr_idx, ss_tc_idx, ss_rlo, ss_rhi, ss_clo, ss_chi, ss_trp_h, ss_tri_h, ss_tcp_h, ss_tci_h;
    std::vector<float> ss_tr_w, ss_tc_w;
    DevBuf ss_ir, ss_wr, ss_ic, ss_wc, ss_trp, ss_tri, ss_trw, ss_tcp, ss_tci, ss_tcw, ss_dlt;
    // what the next fit asks of W before it can queue anything (pmax of fit_ring_model.m:60, row 1 for the first-run test of :25), copied to
    // pinned memory behind the fit that produced W: the next fit reads it without draining the stream (ring_stats_*, api.hip)
    DevBuf stat_dev; void *stat_host = nullptr; hipEvent_t stat_ev = nullptr; bool stat_valid = false;
    int lane = 0;                                          // the execution lane (stream + scratch set) of this patch's calls: cnmfe_ctx::activate, option "lanes"
    ~Patch() { if (stat_host) (void)hipHostFree(stat_host); if (stat_ev) (void)hipEventDestroy(stat_ev); }
};

// device scratch of the OASIS kernels (deconv.hip): pool / task tables, grown on demand and kept with the context
struct DeconvScratch { DevBuf pv, pw, pt, pl, tkp, tko, tkl, tkv, pnum, list, ybuf, obuf, tbuf;
    void swap(DeconvScratch &o) { DevBuf *a[] = {&pv, &pw, &pt, &pl, &tkp, &tko, &tkl, &tkv, &pnum, &list, &ybuf, &obuf, &tbuf}, *b[] = {&o.pv, &o.pw, &o.pt, &o.pl, &o.tkp, &o.tko, &o.tkl, &o.tkv, &o.pnum, &o.list, &o.ybuf, &o.obuf, &o.tbuf};
        for (int i = 0; i < 13; ++i) a[i]->swap(*b[i]); } };

// One patch's temporal update set up but not swept (cnmfe_hals_temporal_job): its projections, A'A lists and traces in buffers of its own, its
// Gauss-Seidel level schedule on the host.  cnmfe_temporal_jobs_sweep then runs level l of EVERY job of the context in one launch -- the patches of a
// rank are independent (update_temporal_parallel.m:112-186 is a parfor over them), and a level is a handful of workgroups whose duration is one
// trace's chain of work: sixteen patches' levels one after the other leave the chip empty sixteen times as long.
struct TemporalJob {
    Patch *P = nullptr; int32_t K = 0; int64_t ldc = 0, T = 0; int maxIter = 0; bool deconv = false; cnmfe_deconv_opts dopts{};
    DevBuf dC, dColptr, dErow, dAval, dU, dCraw, dNk, dNidx, dNval, dNptr, dAa, dOvf, dS, dPars, dSn, dB;
    std::vector<std::vector<int>> levels;
    bool swept = false;
    bool dag = false;                                      // `levels` are the levels of the dependency graph over the items of all maxIter sweeps (factor.hip, dag_schedule)
    // A'A of the job comes back into pinned memory WITHOUT a wait in cnmfe_hals_temporal_job (the host goes on to the next patch); the sweep call waits once
    // for all jobs and finishes each: aa = diag(A'A) against the host-side column test, the projection again if a footprint term overflowed its list
    std::vector<int> diag; std::vector<char> upd; bool term_applied = false, finished = false; int nn = 0;
    float *pin = nullptr; size_t pin_cap = 0;             // nn floats of A'A values + one int (the term-projection overflow word)
    ~TemporalJob() { if (pin) (void)hipHostFree(pin); }
};

}  // namespace cnmfe

namespace cnmfe {
// Pinned staging for the small host -> device uploads of a call (index lists, CSR arrays, tables).  to_dev() copies the caller's data into
// the arena and enqueues the transfer from there, so (a) the source may go away as soon as to_dev returns and (b) the call does not have to
// drain the stream for that -- the host prepares the next patch while the GPU still works on this one.  Two halves: before allocating in a
// half again, the host waits for the event that marks the last transfer enqueued out of it.
struct PinArena {
    char *p = nullptr; size_t cap = 0, off = 0; int half = 0;
    hipEvent_t ev[2] = {nullptr, nullptr}; bool ev_set[2] = {false, false};
    int init(size_t bytes) {
        if (p) return 0;
        if (hipHostMalloc((void **)&p, bytes, hipHostMallocDefault) != hipSuccess) { p = nullptr; return -1; }
        cap = bytes;
        for (int i = 0; i < 2; ++i) if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return -1;
        return 0;
    }
    // nullptr: the request does not fit half the arena (the caller copies straight from its buffer and waits)
    template <class F> void *take(size_t n, F &&stream_of) {      // stream_of(): asked for only when the halves switch (it sends the held-back uploads first)
        n = (n + 255) & ~size_t(255);
        if (!p || n > cap / 2) return nullptr;
        const size_t end = (size_t)(half + 1) * (cap / 2);
        if (off + n > end) {
            (void)hipEventRecord(ev[half], stream_of()); ev_set[half] = true;
            half ^= 1; off = (size_t)half * (cap / 2);
            if (ev_set[half]) (void)hipEventSynchronize(ev[half]);
        }
        void *r = p + off; off += n; return r;
    }
    void swap(PinArena &o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(off, o.off); std::swap(half, o.half); for (int i = 0; i < 2; ++i) { std::swap(ev[i], o.ev[i]); std::swap(ev_set[i], o.ev_set[i]); } }
    ~PinArena() { if (p) (void)hipHostFree(p); for (int i = 0; i < 2; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]); }
};
}  // namespace cnmfe

namespace cnmfe {
// Small uploads out of the pinned arena are not enqueued one by one: to_dev() notes (destination, arena slot, length) here, and whatever next asks for
// the stream -- a launch, a memset, a copy, a wait: cnmfe_ctx::st() -- first sends all of them as ONE copy kernel.  A call's ten or twenty index
// lists and tables then cost one dispatch instead of one each (44 per patch and iteration in the 4 x 4-patch configuration: 700 dependent
// 4-microsecond dispatches, 6 ms of GPU time line per iteration, profiles/r03/gap_analysis_c4.txt).
constexpr int PIN_NSEG = 24;
// int8 matrix pipe, sums over FRAMES (gram_i8.hpp, win_proj_i8.hpp): 4 digit pairs x 64 frames x 2^14 per accumulation step keep an int32 sum exact for this many frames
constexpr int I8_SEG_FRAMES = 24576;
struct PinSegs { const uint4 *src[PIN_NSEG]; uint4 *dst[PIN_NSEG]; unsigned n16[PIN_NSEG]; };
}  // namespace cnmfe

// Execution lanes (option "lanes", round 6).  The patches of a rank are independent inside each of the three updates (the reference's parfor), and most of a small
// patch's kernels are a few workgroups with 8-10 us of dispatch latency between two dependent ones: on ONE stream sixteen 128 x 128 patches leave the chip idle for a
// sixth of the iteration.  A lane is a stream + a full set of the context's per-call scratch (+ its own pinned upload arena and held-back uploads); a patch belongs to
// lane (creation order) mod lanes, every call with a patch id ACTIVATES the patch's lane first (get_patch), and the members below are swapped in and out of the
// context, so that no code path knows about lanes.  Calls without a patch id that touch what all patches share (the bound traces, the stitch accumulator, the
// temporal jobs' sweep, post-processing) run on lane 0 after lane 0 has been made to wait for the other lanes (join), and leave an event the other lanes wait for
// at their next activation (fork) -- cnmfe::GlobalScope.  lanes = 1 (the default): none of this does anything.
struct cnmfe_lane {
    hipStream_t stream_ = nullptr; cnmfe::PinSegs pseg{}; int npseg = 0; cnmfe::PinArena pin; int64_t spatial_nnz = -1;
    cnmfe::DevBuf vp[32]; size_t hw_vp[32] = {}; int64_t last_ldc = 0;
    cnmfe::DevBuf ysig_low, up_tmp, bgs_r, bgs_b, bgs_upr, bgs_upc; int bgs_patch = -1, bgs_d1s = 0; int64_t bgs_dF = 0;
    cnmfe::DevBuf bf, dig_smax, dig_rspart, dig_scale, tdig, tscale, gk, win_items, bf2, outl_cnt, outl_sel, cov, rowsum, tmp[16];
    size_t hw_cc = 0, hw_cm = 0, hw_wa[3] = {0, 0, 0};
    cnmfe::DevBuf inc[7], stg[4], wcodes, solve_fill, stage, scr[24]; cnmfe::DeconvScratch dscr;
    hipEvent_t ev = nullptr; int64_t fork_seen = 0; bool dirty = false;      // (these three describe the lane itself and are never swapped)
    ~cnmfe_lane() { if (stream_) (void)hipStreamDestroy(stream_); if (ev) (void)hipEventDestroy(ev); }
};

struct cnmfe_ctx {
    int device = 0;
    hipStream_t stream_ = nullptr;
    cnmfe::PinSegs pseg; int npseg = 0;
    void flush_copies();                                   // api.hip
    hipStream_t st() { if (npseg) flush_copies(); return stream_; }   // the compute stream, every held-back upload enqueued first (npseg: written by this context's own thread only, pin_flush_range)
    cnmfe::PinArena pin;
    int64_t spatial_nnz = -1;                              // values of the last cnmfe_update_spatial still in scr[6] (deferred fetch)
    hipStream_t copy_stream = nullptr;                     // device -> pinned host downloads that should not hold up the compute stream
    hipEvent_t ev_bound_ready = nullptr, ev_copy_done = nullptr; bool copy_pending = false;
    // every batch of downloads on the copy stream gets a generation number and an event of its own: a host buffer waits for ITS batch (cnmfe_copy_wait),
    // not for whatever was queued on the copy stream since (cnmfe_stitch_wait) -- releasing last iteration's traces must not wait for this iteration's
    int64_t copy_gen = 0; std::vector<std::pair<int64_t, hipEvent_t>> copy_gens; std::vector<hipEvent_t> copy_ev_pool;
    int copy_batch_mark();                                 // api.hip: after the last enqueue of a batch on copy_stream
    std::vector<hipEvent_t> tickets; std::vector<char> ticket_busy;   // cnmfe_update_spatial_fetch_async / cnmfe_ticket_wait
    int *ticket_flags = nullptr;                           // pinned, TICKET_FLAGS words: what the kernels in front of ticket t raised since the previous take (k_flag_take: read and cleared at the ticket's place in the stream)
    cnmfe::DevBuf ticket_dev;                              // the device side of those words
    static constexpr size_t TICKET_FLAGS = 1024;
    cnmfe::Profiler prof;
    std::map<int, cnmfe::Patch *> patches;
    // scratch shared by all patches of this context (sized for the largest)
    cnmfe::DevBuf bound;      // trace matrix bound with cnmfe_traces_bind (K x ldc fp32), passed as c_order = CNMFE_BOUND
    int32_t bound_K = 0; int64_t bound_T = 0; int bound_order = 1; bool bound_valid = false;
    int64_t bound_gen = 0;    // counts the changes of the bound matrix' CONTENT (cnmfe_traces_bind, the stitch, deconvTemporal on it): what a table derived from its rows is valid for
    cnmfe::DevBuf vp[32];     // scratch of the sweep-free projections (vproj.hip; [16..]: the bg_ssub > 1 forms)
    size_t hw_vp[32] = {};
    int64_t last_ldc = 0;     // row stride of the centred traces the last residual_run left in tmp[1]
    cnmfe::DevBuf ysig_low;   // bg_ssub > 1: residual sweep of the low-resolution patch
    cnmfe::DevBuf up_tmp;     // bg_ssub > 1: column-upsampled W*(...) (low rows x block columns)
    cnmfe::DevBuf bgs_r, bgs_b, bgs_upr, bgs_upc;   // bg_ssub > 1, reconstruct_background / compute_RSS: R_low, W*R_low, replication maps
    int bgs_patch = -1, bgs_d1s = 0; int64_t bgs_dF = 0;   // the patch bgs_b belongs to (cnmfe_background_ssub)
    cnmfe::DevBuf bf;         // tiled centred background residual  [blk][t'][256] fp32
    cnmfe::DevBuf dig_smax;   // gram_i8.hpp: per block-region pixel max |Bf| (float bits) of the build in flight
    cnmfe::DevBuf dig_rspart; // gram_i8.hpp: per frame chunk the row sums of the build in flight (reduced in a fixed order by k_rs_reduce)
    cnmfe::DevBuf dig_scale;  // gram_i8.hpp: per block-region pixel the scale of its 32-bit fixed-point trace (the int8-digit table build)
    cnmfe::DevBuf tdig, tscale, gk, win_items;   // win_proj_i8.hpp: digit planes / scales of the centred traces, their K x K Gram matrix, the (block, trace group) work items
    cnmfe::DevBuf bf2, outl_cnt, outl_sel;   // outlier branch of the ring fit: clipped copy of bf, outliers per frame, kept frames
    cnmfe::DevBuf cov;        // block-pair covariances [pair][256][256] (fp64)
    cnmfe::DevBuf rowsum;     // [blk][256] double
    cnmfe::DevBuf tmp[16];    // small scratch
    size_t hw_cc = 0, hw_cm = 0, hw_wa[3] = {0, 0, 0};     // high-water sizes of the buffers that rotate through tmp[1], tmp[2], tmp[8..10] (DevBuf::ensure_hw)
    cnmfe::DevBuf inc[7];     // incremental ring regression: block footprint lists, U~, trace sums
    cnmfe::DevBuf stg[4];     // ring_solve_staged.hpp: per-neuron window metadata, per-list-position metadata, the windows of U~ and of A
    cnmfe::DevBuf wcodes, solve_fill;   // ring solve: the block-pair codes per window origin (k_win_codes); the fill values {0, 1} in global memory (ring_solve.hpp)
    cnmfe::DevBuf stage;      // upload staging
    // per-call device scratch of the factor updates (factor.hip, deconv.hip): grown on demand, NEVER freed between calls -- a hipMalloc /
    // hipFree pair per buffer and call cost more than the small kernels they serve, and hipFree drains the device
    cnmfe::DevBuf scr[24];
    cnmfe::DeconvScratch dscr;
    // the traces the last cnmfe_hals_temporal[_deconv] / cnmfe_fast_temporal left on the device (C_raw rows, row stride last_t_ldc, and aa): what
    // cnmfe_stitch_add folds into the stitch accumulator without a host round trip
    cnmfe::DevBuf last_craw, last_aa;
    std::vector<cnmfe::TemporalJob *> tjobs; int tjobs_used = 0;          // cnmfe_hals_temporal_job: kept (with their buffers) from update to update, counted from cnmfe_stitch_begin
    cnmfe::DevBuf dcv_c, dcv_s, dcv_pars, dcv_sn;          // cnmfe_deconv_temporal_bound: C_raw - b (after the swap with `bound`), S, kernel_pars, sn -- read by the copy stream, so not shared scratch
    int32_t last_t_K = 0; int64_t last_t_ldc = 0, last_t_T = 0; bool last_t_valid = false;
    // overlap-region stitch of update_temporal_parallel.m:264-280: acc[k][0..T) = sum_m aa_m(k) C_raw_m(k,:), acc[k][ld-1..] ... see cnmfe_stitch_begin
    cnmfe::DevBuf stitch;     // K rows of stitch_ld floats: [0, T) the weighted sum, column stitch_ld - 4 the sum of the weights
    int32_t stitch_K = 0; int64_t stitch_T = 0, stitch_ld = 0; bool stitch_open = false;
    void *rccl_comm = nullptr; int rccl_rank = 0, rccl_n = 0;          // single-process multi-GPU stitch (cnmfe_stitch_temporal)
    cnmfe::DevBuf errflag;    // one int, set by kernels that meet a state the host-side set-up should have excluded (checked at the next sync)
    std::map<std::string, int64_t> opts;
    int trace_level = 0;                                   // opts["host_trace"], read by every LAUNCH
    int64_t opt(const char *n, int64_t dflt) const { auto it = opts.find(n); return it == opts.end() ? dflt : it->second; }
    // lanes (see cnmfe_lane): lanes[k] holds lane k's members while another lane is active; empty = one lane
    std::vector<cnmfe_lane *> lanes; int cur_lane = 0, patches_created = 0, spatial_lane = 0;
    hipEvent_t ev_fork = nullptr; int64_t fork_gen = 0;
    void swap_lane(cnmfe_lane &L);                         // api.hip
    int activate(int lane);                                // api.hip: make `lane` the active one (its stream behind st(), its scratch behind the members)
    int join_lanes();                                      // api.hip: lane 0 active and waiting for everything the other lanes have been given
    int fork_mark();                                       // api.hip: the other lanes wait for what lane 0 has been given up to here, at their next activation
    ~cnmfe_ctx();
};

namespace cnmfe {
// option "host_trace" = 1: wall-clock marks of the host-side phases of a call on stderr (scripts/host_timeline.py)
struct HostTrace {
    bool on; const char *name; std::chrono::steady_clock::time_point t0, last;
    HostTrace(cnmfe_ctx *ctx, const char *n) : on(ctx->opt("host_trace", 0) != 0), name(n) { if (on) t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[host_trace] %s: %-28s %8.3f ms (at %8.3f)\n", name, what, std::chrono::duration<double, std::milli>(t - last).count(),
                std::chrono::duration<double, std::milli>(t - t0).count());
        last = t;
    }
};
// (a call with a patch id runs on the patch's lane: its stream and its scratch are the context's from here on)
inline Patch *get_patch(cnmfe_ctx *ctx, int id) {
    auto it = ctx->patches.find(id);
    if (it == ctx->patches.end()) return nullptr;
    if (!ctx->lanes.empty() && ctx->activate(it->second->lane) != 0) return nullptr;
    return it->second;
}
// a call WITHOUT a patch id that reads or writes what the patches share: lane 0, behind everything the other lanes were given; what it queues is waited for by
// the other lanes' next calls
struct GlobalScope {
    cnmfe_ctx *ctx; int rc = 0;
    explicit GlobalScope(cnmfe_ctx *c) : ctx(c) { if (ctx && !ctx->lanes.empty()) rc = ctx->join_lanes(); }
    ~GlobalScope() { if (ctx && !ctx->lanes.empty()) (void)ctx->fork_mark(); }
};
int pinned_to_dev(cnmfe_ctx *ctx, void *dst, const void *src_pinned, size_t bytes);   // api.hip
void pin_register(cnmfe_ctx *ctx, bool live);
// upload a host vector to a DevBuf on the context stream
template <class T> inline int to_dev(cnmfe_ctx *ctx, DevBuf &b, const T *h, size_t n) {
    RET(b.ensure(std::max<size_t>(n, 1) * sizeof(T)));
    if (!n) return 0;
    if (void *st = ctx->pin.take(n * sizeof(T), [ctx] { return ctx->st(); })) {             // staged: `h` is free again when this returns
        memcpy(st, h, n * sizeof(T));
        // small uploads go by a copy KERNEL that reads the pinned arena over PCIe, not by hipMemcpyAsync: the copy engine also serves the big asynchronous
        // downloads of the traces (20-60 MB behind every temporal update), and an upload of a few KB queued behind one of those held the next call's first
        // kernel back by 0.5-1.5 ms (profiles/r03/gap_analysis_*.txt)
        return pinned_to_dev(ctx, b.p, st, n * sizeof(T));
    }
    CK(hipMemcpyAsync(b.p, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->st()));
    CK(hipStreamSynchronize(ctx->st()));                                   // too large for the arena: straight from the caller's buffer
    return 0;
}
// [k][t] row-major fp32 copy of a K x T matrix given in `order`; device result has row stride ldc (multiple of 4)
int upload_traces(cnmfe_ctx *ctx, DevBuf &dst, const float *C, int32_t K, int64_t T, int order, int64_t *ldc);
int download_traces(cnmfe_ctx *ctx, const float *dC, int64_t ldc, float *C, int32_t K, int64_t T, int order);

// implemented in the kernel translation units
int bg_fit_ring(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                const float *C, int c_order, int with_projection, float *b0_out, int64_t info[4], int b0_only = 0, double thresh_outlier = NAN);
int tu_warm_resid(); int tu_warm_bg();
int win_i8_table(cnmfe_ctx *ctx, Patch *P, const char *name_dig, const char *name_proj, int K, const float *dCc, int64_t ldc, const std::vector<int> &lst_ptr, const std::vector<int> &blall,
                 const int *dLp, const int *dLk, int nsg, double *dUt, int64_t ut_stride);   // bg.hip: the int8 window projection of all frames (the spatial update's table, vproj.hip)
int tu_warm_factor(); int tu_warm_deconv(); int tu_warm_ssub(); int tu_warm_vproj();   // one per translation unit: load its code object (cnmfe_create)
int bg_reserve(cnmfe_ctx *ctx, Patch *P);                 // bg.hip: the ring fit's large buffers, sized by the geometry, allocated ahead of the first fit
int residual_run(cnmfe_ctx *ctx, Patch *P, int pid, int32_t Ksel, const int64_t *A_colptr, const int32_t *A_rowidx,
                 const float *A_val, const float *C, int c_order, float *Ysig_out, int out_memspace, DevBuf *outbuf = nullptr, int tables_only = 0);
int ysig_export(cnmfe_ctx *ctx, Patch *P, DevBuf &ysig, float *Ysig_out, int out_memspace);
int rss_run(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val, const float *C, int c_order,
            const float *b0_block, const float *b0_new, double *rss_out);
int bg_reconstruct_run(cnmfe_ctx *ctx, Patch *P, const float *b0_block, const float *b0_new, int64_t frame0, int64_t nframes, float *out, int out_memspace);
int residual_materialize(cnmfe_ctx *ctx, Patch *P);       // fold a pending footprint term into the resident Ysig (no-op without one); realizes a virtual residual first
int residual_realize(cnmfe_ctx *ctx, Patch *P);           // a virtual residual (Patch::ysig_virtual) becomes a resident one: the ring sweep runs now (no-op otherwise)
// rows of the bound trace matrix a (C, c_order) argument names: false when it is some other matrix
bool bound_rows_of(const cnmfe_ctx *ctx, const float *C, int c_order, int32_t K, std::vector<int32_t> &rows);
// vproj.hip -- the projections of a virtual residual.  Return 1 when they cannot serve the request (the caller realizes the residual and projects Ysig).
// (dQ != nullptr: instead of U, the ring sums sum_i W(m,i) P(m + o_i, k) of the mask's entries in fp64 -- what the bg_ssub form upsamples)
int vproj_spatial(cnmfe_ctx *ctx, Patch *P, int32_t K, const float *C, int c_order, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                  const int *dErow, const int *dEcol, const float *dCc, int64_t ldc, float *dU, double *dQ = nullptr);
int vproj_temporal(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                   const int64_t *dColptr, const int *dErow, const float *dAval, float *dU, int64_t ldu);
// the same for a residual of cnmfe_residual_ssub (res_kind 2): through the resampling maps, DESIGN.md section 3 bg_ssub
int vproj_spatial_ssub(cnmfe_ctx *ctx, Patch *M, int32_t K, const float *C, int c_order, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                       const int *dErow, const int *dEcol, const float *dCc, int64_t ldc, float *dU);
int vproj_temporal_ssub(cnmfe_ctx *ctx, Patch *M, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                        const int64_t *dColptr, const int *dErow, const float *dAval, float *dU, int64_t ldu);
int ssub_taps(cnmfe_ctx *ctx, Patch *M, const Patch *R);  // ssub.hip: the upsampling taps of M's block region from R's grid, host + device (once per patch)
int ssub_realize(cnmfe_ctx *ctx, Patch *M);               // ssub.hip: the low-resolution sweep + upsample behind a virtual residual of res_kind 2
int residual_term_project(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *dColptr, const int *dErow, const float *dAval, float *dU, int64_t ldu, int *dOverflow);   // 1: cannot be applied, materialise instead; *dOverflow set on the device if a footprint meets > 512 traces
bool residual_term_foldable_spatial(const Patch *P, int32_t K, int64_t ldc);   // the pending term can enter the spatial update through its projection (no pass over Ysig)
int residual_term_fold_spatial(cnmfe_ctx *ctx, Patch *P, int32_t K, int64_t nnz, const int *dErow, const int *dEcol, const float *dCc, int64_t ldc, float *dU, DevBuf &dG);
int check_csc_pub(const char *what, int32_t ncol, int64_t nrow, const int64_t *colptr, const int32_t *rowidx);
int spatial_run(cnmfe_ctx *ctx, Patch *P, int algorithm, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                const float *A_val, const float *C, int c_order, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                const float *sn, int32_t param, float *A_out);
int temporal_run(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                 const float *C_in, int c_order, int32_t maxIter, float *C_out, float *C_raw_out, float *aa_out,
                 const cnmfe_deconv_opts *dopts, float *kernel_pars, float *S_out, float *sn_out, TemporalJob *job = nullptr);
int temporal_sweep_jobs(cnmfe_ctx *ctx);                  // factor.hip: the level sweeps of every job set up since cnmfe_stitch_begin, level by level across the jobs
#ifdef __HIPCC__
// LDS-DMA: 64 lanes x 16 B, global (uniform base + per-lane 32-bit byte offset) -> LDS [lds_dst + lane*16].  Invisible to
// hipcc's s_waitcnt bookkeeping: count completion by hand (vmcnt), then barrier, then read (cdna_hip_programming 5.7).
__device__ __forceinline__ void glds16(const void *base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
#endif

int sn_pixels_run(cnmfe_ctx *ctx, Patch *P, float *sn_out);
int fast_temporal_run(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                      int c_order, float *C_raw_out, float *aa_out);
int deconv_bound_run(cnmfe_ctx *ctx, const cnmfe_deconv_opts *opts, float *C_out, float *C_raw_out, float *S_out, float *pars_out, float *sn_out);
int deconv_all_run(cnmfe_ctx *ctx, int32_t K, int64_t T, float *C_raw, int c_order, const cnmfe_deconv_opts *opts,
                   float *C_out, float *S_out, float *pars_out, float *sn_out);
int postproc_run(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                 const float *A_val, uint8_t *keep);
int ensure_ymean(cnmfe_ctx *ctx, Patch *P);
int sn_video_run(cnmfe_ctx *ctx, Patch *P, int64_t nframes, float *sn_out);
int spatial_fetch(cnmfe_ctx *ctx, float *A_out, int64_t nnz);
int spatial_fetch_connected(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *IND_colptr, const int32_t *IND_rowidx, float *A_out, uint8_t *keep, bool wait = true);
int ring_first_run(cnmfe_ctx *ctx, Patch *P, bool *first);
int ring_stats_enqueue(cnmfe_ctx *ctx, Patch *P);                    // after W changed on the stream: count + row 1 to pinned memory, event
int ring_stats_get(cnmfe_ctx *ctx, Patch *P, int *pmax, bool *first); // waits for that event only (falls back to a fresh evaluation)
int center_traces(cnmfe_ctx *ctx, const float *C, int64_t ldc, int32_t K, int64_t T, DevBuf &Cc, DevBuf &Cmean);
int upload_centered(cnmfe_ctx *ctx, DevBuf &stage, const float *C, int32_t K, int64_t T, int order, DevBuf &Cc, DevBuf &Cmean, int64_t *ldc_out);
int ctx_errflag(cnmfe_ctx *ctx, int **dflag);            // the device error word (allocated and cleared on first use)
int ctx_check_errflag(cnmfe_ctx *ctx);                   // after a stream sync: CNMFE_ESTATE if a kernel raised it (and clears it)
}  // namespace cnmfe
