// cnmfe_mex.cpp -- thin MEX gateway over the C ABI of include/cnmfe.h (one command string per entry point).
//
// Build where MATLAB is installed (not built in this repository: the image has no MATLAB; tests/test_host_logic.py only syntax-checks this
// file against the stub headers in tests/mex_stub/):
//     mex -O -largeArrayDims cnmfe_mex.cpp -I../../../include -L../.. -lcnmfe_hip
// House conventions follow the reference's own MEX (ca_source_extraction/utilities/graph_conn_comp_mex.cpp:38-64, 88, 115): argument checks
// first, mexErrMsgIdAndTxt on failure, outputs allocated with mxCreate*, inputs never written, scratch from mxMalloc.
//
// mexErrMsgIdAndTxt does not return (it long-jumps out of the MEX function), so NO C++ object with a destructor is alive at any point that can
// fail: every temporary is mxMalloc memory, which MATLAB releases itself when a MEX function exits through an error.
//
//   h = cnmfe_mex('create', device)                                        cnmfe_mex('destroy', h)
//   cnmfe_mex('patch', h, pid, patch_pos, block_pos, d1, d2, T)            distribute_data.m:165-171 rectangles
//   cnmfe_mex('upload', h, pid, Yblock, t0)                                Yblock: d_b x nt of class single / double / uint16 / uint8
//   cnmfe_mex('ring_init', h, pid, radius, num_neighbors)                  get_nhood.m + initComponents_parallel.m:213-236
//   [Wt, b0] = cnmfe_mex('get_ring', h, pid, d, d_b)                       W{m}.' (d_b x d sparse) and b0{m}
//   cnmfe_mex('set_ring', h, pid, Wt, b0)                                  put a W{m}, b0{m} of an earlier session back on the device
//   f = cnmfe_mex('first_run', h, pid)                                     update_background_parallel.m:143
//   cnmfe_mex('bind_traces', h, C)                                         obj.C once per iteration; later C arguments may be int32 ROW INDICES into it
//   info = cnmfe_mex('fit_ring', h, pid, A_block, C_or_rows, with_projection)             fit_ring_model.m:1
//   info = cnmfe_mex('fit_ring_ssub', h, pid, fit_pid, res_pid, bg_ssub, A_block, C_or_rows, with_projection)
//   cnmfe_mex('derive', h, pid, new_pid, bg_ssub, 'nearest' | 'bicubic')
//   cnmfe_mex('residual', h, pid, A_prev_block, C_or_rows)                 update_spatial_parallel.m:162-166 (result stays on the device)
//   cnmfe_mex('residual_ssub', h, pid, res_pid, bg_ssub, A_prev_block, C_or_rows)
//   sn = cnmfe_mex('get_sn', h, pid, d)                                    update_sn = true
//   cnmfe_mex('set_noise', h, pid, sn_block);  sn_block = cnmfe_mex('estimate_noise', h, pid, d_b, nframes)
//   cnmfe_mex('background_ssub', ...), cnmfe_mex('compute_rss_ssub', ...), cnmfe_mex('reconstruct_background_ssub', ...)   bg_ssub > 1
//   A = cnmfe_mex('spatial', h, pid, alg, A_patch, C_or_rows, IND_patch, sn, param)       HALS_spatial*.m / nnls_spatial.m
//   [C, C_raw, aa] = cnmfe_mex('temporal', h, pid, A_patch, C_or_rows, maxIter)           HALS_temporal.m:1 (fewer outputs: fewer downloads)
//   [C, C_raw, S, pars, sn, aa] = cnmfe_mex('temporal_deconv', h, pid, A, C, maxIter, smin, max_tau, pars)
//   [C_raw, aa] = cnmfe_mex('fast_temporal', h, pid, A, T)
//   cnmfe_mex('stitch_begin', h, K, T);  cnmfe_mex('stitch_add', h, ind)   update_temporal_parallel.m:269-278, on the device
//   C_raw = cnmfe_mex('stitch_temporal', hs, subtract_min, K, T)           :279-286 over the contexts hs (one per GPU; RCCL all-reduce)
//   [C, C_raw, S, pars, sn] = cnmfe_mex('deconv_temporal', h, C_raw, smin, max_tau)       deconvTemporal.m:29-105
//   rss = cnmfe_mex('compute_rss', h, pid, A_patch, C_or_rows, b0_block, b0_new_patch)
//   Ybg = cnmfe_mex('reconstruct_background', h, pid, b0_block, b0_new_patch, frame0, nframes)
//   keep = cnmfe_mex('postprocess', h, A, d1, d2)                          post_process_spatial.m:19-32 (connected)
// Round 6 -- the calls the measured host (cnmf_e_amd/sources2d.py, bench.py) makes and this gateway lacked; the .m twins use them:
//   cnmfe_mex('set_option', h, name, value)                                cnmfe_set_option (the tunables of include/cnmfe.h; the defaults ARE the fast path)
//   cnmfe_mex('synchronize', h)                                            wait for the context's stream; reports what its kernels raised
//   q = cnmfe_mex('spatial_queue', h, pid, alg, A_patch, C_or_rows, IND_patch, sn, param)   the update is QUEUED (sweeps + the copy of the result into pinned
//   A = cnmfe_mex('spatial_collect', h, q)                                 memory) and collected later: patch m + 1 is set up and queued while patch m runs
//   job = cnmfe_mex('temporal_job', h, pid, A_patch, C_or_rows, maxIter)   everything of 'temporal' up to the Gauss-Seidel sweeps; 'temporal_jobs_sweep' then runs
//   cnmfe_mex('temporal_jobs_sweep', h)                                    level l of EVERY job in one launch; cnmfe_mex('stitch_add_job', h, job, ind) adds a job
//   cnmfe_mex('stitch_finish_async', h, subtract_min, K, T)                :279-286 + bind on one context without waiting; C_raw = cnmfe_mex('stitch_collect', h)
#include "mex.h"
#include "matrix.h"
#include <string.h>
#include <stdint.h>
#include "cnmfe.h"

#define MAX_CTX 64
static cnmfe_ctx *g_ctx[MAX_CTX];
static int g_nctx = 0;
// queued spatial updates ('spatial_queue' ... 'spatial_collect'): the ticket of the copy into pinned memory and a persistent copy of the mask whose pattern the result has
#define MAX_PEND 1024
typedef struct { int used; cnmfe_ctx *c; int64_t ticket; float *pinned; int64_t nnz; mxArray *ind; } Pending;
static Pending g_pend[MAX_PEND];
// the asynchronous stitch of a context ('stitch_finish_async' ... 'stitch_collect'): K x T floats, row-major, pinned
typedef struct { float *pinned; size_t K, T; } StitchOut;
static StitchOut g_stitch[MAX_CTX];
static void at_exit(void) {
    for (int i = 0; i < MAX_PEND; ++i) if (g_pend[i].used) { if (g_pend[i].pinned) cnmfe_host_free(g_pend[i].pinned); if (g_pend[i].ind) mxDestroyArray(g_pend[i].ind); g_pend[i].used = 0; }
    for (int i = 0; i < MAX_CTX; ++i) if (g_stitch[i].pinned) { cnmfe_host_free(g_stitch[i].pinned); g_stitch[i].pinned = NULL; }
    for (int i = 0; i < g_nctx; ++i) if (g_ctx[i]) { cnmfe_destroy(g_ctx[i]); g_ctx[i] = NULL; }
    g_nctx = 0;
}
#define FAIL(...) mexErrMsgIdAndTxt("cnmfe_mex:error", __VA_ARGS__)
#define CHECK(rc) do { if ((rc) != 0) FAIL("%s", cnmfe_last_error()); } while (0)

static cnmfe_ctx *ctx_of(const mxArray *h) {
    const double v = mxGetScalar(h);
    const int i = (int)v;
    if (i < 1 || i > g_nctx || !g_ctx[i - 1]) FAIL("invalid context handle");
    return g_ctx[i - 1];
}
// ---- plain-data views of MATLAB arrays (all storage mxMalloc'ed) ----
typedef struct { int32_t K; int64_t nnz; int64_t *cp; int32_t *ri; float *v; } Csc;
static Csc csc_of(const mxArray *A) {                       // sparse / full, double / logical -> int64 colptr, int32 rowidx, float values
    Csc o; o.K = 0; o.nnz = 0; o.ri = NULL; o.v = NULL;
    if (mxIsEmpty(A)) { o.cp = (int64_t *)mxCalloc(1, sizeof(int64_t)); return o; }
    mxArray *S = NULL; const mxArray *src = A;
    if (!mxIsSparse(A)) { mxArray *in = (mxArray *)A; if (mexCallMATLAB(1, &S, 1, &in, "sparse")) FAIL("sparse() failed"); src = S; }
    if (!mxIsDouble(src) && !mxIsLogical(src)) FAIL("sparse matrices must be double or logical");
    o.K = (int32_t)mxGetN(src);
    const mwIndex *jc = mxGetJc(src), *ir = mxGetIr(src);
    o.nnz = (int64_t)jc[o.K];
    o.cp = (int64_t *)mxMalloc(((size_t)o.K + 1) * sizeof(int64_t));
    o.ri = (int32_t *)mxMalloc(((size_t)o.nnz + 1) * sizeof(int32_t));
    o.v = (float *)mxMalloc(((size_t)o.nnz + 1) * sizeof(float));
    for (int32_t k = 0; k <= o.K; ++k) o.cp[k] = (int64_t)jc[k];
    if (mxIsLogical(src)) { for (int64_t e = 0; e < o.nnz; ++e) { o.ri[e] = (int32_t)ir[e]; o.v[e] = 1.f; } }
    else { const double *pr = mxGetPr(src); for (int64_t e = 0; e < o.nnz; ++e) { o.ri[e] = (int32_t)ir[e]; o.v[e] = (float)pr[e]; } }
    if (S) mxDestroyArray(S);
    return o;
}
static float *f32_of(const mxArray *M, size_t *n_out) {     // single / double -> float copy
    const size_t n = mxGetNumberOfElements(M);
    float *o = (float *)mxMalloc((n + 1) * sizeof(float));
    if (mxIsSingle(M)) memcpy(o, mxGetData(M), n * sizeof(float));
    else if (mxIsDouble(M)) { const double *p = mxGetPr(M); for (size_t i = 0; i < n; ++i) o[i] = (float)p[i]; }
    else FAIL("expected a single or double array");
    if (n_out) *n_out = n;
    return o;
}
// a trace argument: a K x T single / double matrix (column-major, uploaded) or an int32 vector of 1-based rows of the bound matrix
typedef struct { const float *ptr; int order; } Traces;
static Traces traces_of(const mxArray *C, int32_t K) {
    Traces t; t.ptr = NULL; t.order = CNMFE_COLMAJOR;
    if (K == 0) return t;
    if (mxIsInt32(C)) {
        if ((int32_t)mxGetNumberOfElements(C) != K) FAIL("row list has %d entries for %d footprints", (int)mxGetNumberOfElements(C), (int)K);
        int32_t *rows = (int32_t *)mxMalloc((size_t)K * sizeof(int32_t));
        const int32_t *src = (const int32_t *)mxGetData(C);
        for (int32_t k = 0; k < K; ++k) rows[k] = src[k] - 1;
        t.ptr = (const float *)rows; t.order = CNMFE_BOUND_ROWS;
        return t;
    }
    if ((int32_t)mxGetM(C) != K) FAIL("trace matrix has %d rows for %d footprints", (int)mxGetM(C), (int)K);
    t.ptr = f32_of(C, NULL);
    return t;
}
static mxArray *to_double(const float *v, size_t r, size_t c) {
    mxArray *o = mxCreateDoubleMatrix(r, c, mxREAL);
    double *p = mxGetPr(o);
    for (size_t i = 0; i < r * c; ++i) p[i] = v[i];
    return o;
}
static mxArray *info_of(const int64_t info[4]) {
    mxArray *o = mxCreateDoubleMatrix(1, 4, mxREAL);
    for (int i = 0; i < 4; ++i) mxGetPr(o)[i] = (double)info[i];
    return o;
}
static void deconv_opts_of(cnmfe_deconv_opts *o, double smin, double max_tau) {
    memset(o, 0, sizeof(*o));
    o->type = 1; o->method = 1; o->smin = smin; o->max_tau = max_tau; o->optimize_b = 1; o->optimize_pars = 1; o->maxIter = 10;
}

void mexFunction(int nout, mxArray *pout[], int nin, const mxArray *pin[]) {
    if (nin < 1 || !mxIsChar(pin[0])) FAIL("first argument must be a command string");
    char cmd[40];
    if (mxGetString(pin[0], cmd, sizeof(cmd))) FAIL("command string too long");
    mexAtExit(at_exit);
    if (!strcmp(cmd, "create")) {
        if (g_nctx >= MAX_CTX) FAIL("too many contexts");
        cnmfe_ctx *c = cnmfe_create(nin > 1 ? (int)mxGetScalar(pin[1]) : 0);
        if (!c) FAIL("%s", cnmfe_last_error());
        g_ctx[g_nctx++] = c;
        pout[0] = mxCreateDoubleScalar((double)g_nctx);
        return;
    }
    if (nin < 2) FAIL("missing context handle");
    if (!strcmp(cmd, "stitch_temporal")) {                      // C_raw = cnmfe_mex('stitch_temporal', hs, subtract_min, K, T)   (single, K x T)
        if (nin != 5) FAIL("stitch_temporal: 5 inputs required (hs, subtract_min, K, T)");
        const int n = (int)mxGetNumberOfElements(pin[1]);
        if (n < 1 || n > MAX_CTX) FAIL("stitch_temporal: 1..%d context handles", MAX_CTX);
        cnmfe_ctx *cs[MAX_CTX];
        const double *hv = mxGetPr(pin[1]);
        for (int i = 0; i < n; ++i) { const int k = (int)hv[i]; if (k < 1 || k > g_nctx || !g_ctx[k - 1]) FAIL("invalid context handle"); cs[i] = g_ctx[k - 1]; }
        const size_t K = (size_t)mxGetScalar(pin[3]), T = (size_t)mxGetScalar(pin[4]);
        for (int i = 0; i < n; ++i) {                            // the engine writes stitch_K x stitch_T floats: the output is sized from ITS record, the caller's K, T are only checked
            int32_t Ks = 0; int64_t Ts = 0;
            CHECK(cnmfe_stitch_dims(cs[i], &Ks, &Ts));
            if ((size_t)Ks != K || (size_t)Ts != T) FAIL("stitch_temporal: K x T = %d x %d, but context %d accumulates %d x %d (stitch_begin)", (int)K, (int)T, i + 1, (int)Ks, (int)Ts);
        }
        pout[0] = mxCreateNumericMatrix(K, T, mxSINGLE_CLASS, mxREAL);
        CHECK(cnmfe_stitch_temporal(cs, n, mxGetScalar(pin[2]) != 0, nout > 0 ? (float *)mxGetData(pout[0]) : NULL, CNMFE_COLMAJOR));
        return;
    }
    cnmfe_ctx *c = ctx_of(pin[1]);
    if (!strcmp(cmd, "destroy")) { const int i = (int)mxGetScalar(pin[1]); cnmfe_destroy(c); g_ctx[i - 1] = NULL; return; }
    if (!strcmp(cmd, "bind_traces")) {
        if (nin != 3) FAIL("bind_traces: 3 inputs required");
        if (mxIsEmpty(pin[2])) { CHECK(cnmfe_traces_bind(c, 0, 0, NULL, CNMFE_COLMAJOR)); return; }
        float *Cm = f32_of(pin[2], NULL);
        CHECK(cnmfe_traces_bind(c, (int32_t)mxGetM(pin[2]), (int64_t)mxGetN(pin[2]), Cm, CNMFE_COLMAJOR));
        return;
    }
    if (!strcmp(cmd, "stitch_begin")) {
        if (nin != 4) FAIL("stitch_begin: 4 inputs required (h, K, T)");
        CHECK(cnmfe_stitch_begin(c, (int32_t)mxGetScalar(pin[2]), (int64_t)mxGetScalar(pin[3])));
        return;
    }
    if (!strcmp(cmd, "stitch_add")) {                           // ind: 1-based rows (double or int32)
        if (nin != 3) FAIL("stitch_add: 3 inputs required (h, ind)");
        const size_t n = mxGetNumberOfElements(pin[2]);
        int32_t *ind = (int32_t *)mxMalloc((n + 1) * sizeof(int32_t));
        if (mxIsInt32(pin[2])) { const int32_t *s = (const int32_t *)mxGetData(pin[2]); for (size_t i = 0; i < n; ++i) ind[i] = s[i] - 1; }
        else if (mxIsDouble(pin[2])) { const double *s = mxGetPr(pin[2]); for (size_t i = 0; i < n; ++i) ind[i] = (int32_t)s[i] - 1; }
        else FAIL("stitch_add: ind must be double or int32");
        CHECK(cnmfe_stitch_add(c, (int32_t)n, ind));
        return;
    }
    if (!strcmp(cmd, "deconv_temporal")) {                      // [C, C_raw, S, pars, sn] = cnmfe_mex('deconv_temporal', h, C_raw, smin, max_tau)
        if (nin != 5) FAIL("deconv_temporal: 5 inputs required");
        const size_t K = mxGetM(pin[2]), T = mxGetN(pin[2]);
        float *Cr = f32_of(pin[2], NULL);
        cnmfe_deconv_opts o; deconv_opts_of(&o, mxGetScalar(pin[3]), mxGetScalar(pin[4]));
        float *Co = (float *)mxMalloc((K * T + 1) * sizeof(float)), *S = (float *)mxMalloc((K * T + 1) * sizeof(float));
        float *kp = (float *)mxMalloc((K + 1) * sizeof(float)), *sn = (float *)mxMalloc((K + 1) * sizeof(float));
        CHECK(cnmfe_deconv_temporal(c, (int32_t)K, (int64_t)T, Cr, CNMFE_COLMAJOR, &o, Co, S, kp, sn));
        pout[0] = to_double(Co, K, T);
        if (nout > 1) pout[1] = to_double(Cr, K, T);
        if (nout > 2) pout[2] = to_double(S, K, T);
        if (nout > 3) pout[3] = to_double(kp, K, 1);
        if (nout > 4) pout[4] = to_double(sn, K, 1);
        return;
    }
    if (!strcmp(cmd, "postprocess")) {
        if (nin != 5) FAIL("postprocess: 5 inputs required");
        Csc A = csc_of(pin[2]);
        uint8_t *keep = (uint8_t *)mxCalloc((size_t)A.nnz + 1, 1);
        CHECK(cnmfe_post_process_spatial(c, (int32_t)mxGetScalar(pin[3]), (int32_t)mxGetScalar(pin[4]), A.K, A.cp, A.ri, A.v, keep));
        pout[0] = mxCreateLogicalMatrix((size_t)A.nnz, 1);
        mxLogical *k = mxGetLogicals(pout[0]);
        for (int64_t i = 0; i < A.nnz; ++i) k[i] = keep[i] != 0;
        return;
    }
    if (!strcmp(cmd, "set_option")) {                           // cnmfe_mex('set_option', h, name, value)
        if (nin != 4) FAIL("set_option: 4 inputs required (h, name, value)");
        char name[48];
        if (!mxIsChar(pin[2]) || mxGetString(pin[2], name, sizeof(name))) FAIL("set_option: bad option name");
        CHECK(cnmfe_set_option(c, name, (int64_t)mxGetScalar(pin[3])));
        return;
    }
    if (!strcmp(cmd, "synchronize")) {
        if (nin != 2) FAIL("synchronize: 2 inputs required");
        CHECK(cnmfe_synchronize(c));
        return;
    }
    if (!strcmp(cmd, "temporal_jobs_sweep")) {                  // the Gauss-Seidel levels of every queued job, one launch per level (update_temporal_parallel.m:112-186 is a parfor)
        if (nin != 2) FAIL("temporal_jobs_sweep: 2 inputs required");
        CHECK(cnmfe_temporal_jobs_sweep(c));
        return;
    }
    if (!strcmp(cmd, "stitch_add_job")) {                       // cnmfe_mex('stitch_add_job', h, job, ind): ind 1-based rows (double or int32)
        if (nin != 4) FAIL("stitch_add_job: 4 inputs required (h, job, ind)");
        const int32_t n = (int32_t)mxGetNumberOfElements(pin[3]);
        int32_t *ind = (int32_t *)mxMalloc(((size_t)n + 1) * sizeof(int32_t));
        if (mxIsInt32(pin[3])) { const int32_t *src = (const int32_t *)mxGetData(pin[3]); for (int32_t i = 0; i < n; ++i) ind[i] = src[i] - 1; }
        else if (mxIsDouble(pin[3])) { const double *src = mxGetPr(pin[3]); for (int32_t i = 0; i < n; ++i) ind[i] = (int32_t)src[i] - 1; }
        else FAIL("stitch_add_job: ind must be double or int32");
        CHECK(cnmfe_stitch_add_job(c, (int32_t)mxGetScalar(pin[2]), n, ind));
        return;
    }
    if (!strcmp(cmd, "stitch_finish_async")) {                  // cnmfe_mex('stitch_finish_async', h, subtract_min, K, T)
        if (nin != 5) FAIL("stitch_finish_async: 5 inputs required (h, subtract_min, K, T)");
        const int hi = (int)mxGetScalar(pin[1]) - 1;
        const size_t K = (size_t)mxGetScalar(pin[3]), T = (size_t)mxGetScalar(pin[4]);
        if (g_stitch[hi].pinned) { cnmfe_host_free(g_stitch[hi].pinned); g_stitch[hi].pinned = NULL; }
        g_stitch[hi].pinned = (float *)cnmfe_host_alloc((K * T + 1) * sizeof(float));
        if (!g_stitch[hi].pinned) FAIL("%s", cnmfe_last_error());
        g_stitch[hi].K = K; g_stitch[hi].T = T;
        CHECK(cnmfe_stitch_finish_async(c, mxGetScalar(pin[2]) != 0, g_stitch[hi].pinned));
        return;
    }
    if (!strcmp(cmd, "stitch_collect")) {                       // C_raw = cnmfe_mex('stitch_collect', h)   (single, K x T)
        if (nin != 2) FAIL("stitch_collect: 2 inputs required");
        const int hi = (int)mxGetScalar(pin[1]) - 1;
        if (!g_stitch[hi].pinned) FAIL("stitch_collect: no stitch_finish_async outstanding on this context");
        CHECK(cnmfe_stitch_wait(c));
        const size_t K = g_stitch[hi].K, T = g_stitch[hi].T;
        pout[0] = mxCreateNumericMatrix(K, T, mxSINGLE_CLASS, mxREAL);
        float *dst = (float *)mxGetData(pout[0]);
        const float *src = g_stitch[hi].pinned;                 // row-major K x T -> MATLAB's column-major
        for (size_t k = 0; k < K; ++k) for (size_t t = 0; t < T; ++t) dst[t * K + k] = src[k * T + t];
        cnmfe_host_free(g_stitch[hi].pinned); g_stitch[hi].pinned = NULL;
        return;
    }
    if (!strcmp(cmd, "spatial_collect")) {                      // A = cnmfe_mex('spatial_collect', h, q)
        if (nin != 3) FAIL("spatial_collect: 3 inputs required (h, q)");
        const int q = (int)mxGetScalar(pin[2]) - 1;
        if (q < 0 || q >= MAX_PEND || !g_pend[q].used || g_pend[q].c != c) FAIL("spatial_collect: no such queued update on this context");
        Pending *P = &g_pend[q];
        const int rc = cnmfe_ticket_wait(c, P->ticket);
        const mxArray *IND = P->ind;
        const size_t K = mxGetN(IND);
        if (rc == 0) {
            pout[0] = mxCreateSparse(mxGetM(IND), K, P->nnz > 0 ? (size_t)P->nnz : 1, mxREAL);           // same pattern as IND
            memcpy(mxGetJc(pout[0]), mxGetJc(IND), (K + 1) * sizeof(mwIndex));
            memcpy(mxGetIr(pout[0]), mxGetIr(IND), (size_t)P->nnz * sizeof(mwIndex));
            for (int64_t i = 0; i < P->nnz; ++i) mxGetPr(pout[0])[i] = P->pinned[i];
        }
        cnmfe_host_free(P->pinned); mxDestroyArray(P->ind); P->pinned = NULL; P->ind = NULL; P->used = 0;
        CHECK(rc);
        return;
    }
    if (nin < 3) FAIL("missing patch id");
    const int pid = (int)mxGetScalar(pin[2]);
    if (!strcmp(cmd, "patch")) {
        if (nin != 8) FAIL("patch: 8 inputs required");
        if (mxGetNumberOfElements(pin[3]) != 4 || mxGetNumberOfElements(pin[4]) != 4) FAIL("patch: positions are [r0 r1 c0 c1]");
        int32_t pr[4], br[4];
        for (int i = 0; i < 4; ++i) { pr[i] = (int32_t)mxGetPr(pin[3])[i]; br[i] = (int32_t)mxGetPr(pin[4])[i]; }
        CHECK(cnmfe_patch_create(c, pid, pr, br, (int32_t)mxGetScalar(pin[5]), (int32_t)mxGetScalar(pin[6]), (int64_t)mxGetScalar(pin[7])));
    } else if (!strcmp(cmd, "upload")) {
        if (nin != 5) FAIL("upload: 5 inputs required");
        const mxArray *Y = pin[3];
        const int dt = mxIsSingle(Y) ? CNMFE_F32 : mxIsDouble(Y) ? CNMFE_F64 : mxIsUint16(Y) ? CNMFE_U16 : mxIsUint8(Y) ? CNMFE_U8 : -1;
        if (dt < 0) FAIL("upload: unsupported class %s", mxGetClassName(Y));
        CHECK(cnmfe_upload_block(c, pid, mxGetData(Y), dt, CNMFE_HOST, (int64_t)mxGetScalar(pin[4]), (int64_t)mxGetN(Y)));
    } else if (!strcmp(cmd, "ring_init")) {
        if (nin < 4) FAIL("ring_init: radius required");
        CHECK(cnmfe_ring_init(c, pid, (int32_t)mxGetScalar(pin[3]), nin > 4 && !mxIsEmpty(pin[4]) ? (int32_t)mxGetScalar(pin[4]) : 0));
    } else if (!strcmp(cmd, "fit_reserve")) {
        CHECK(cnmfe_fit_reserve(c, pid));
    } else if (!strcmp(cmd, "first_run")) {
        int f = 0;
        CHECK(cnmfe_ring_first_run(c, pid, &f));
        pout[0] = mxCreateLogicalScalar(f != 0);
    } else if (!strcmp(cmd, "fit_ring")) {
        if (nin != 6 && nin != 7) FAIL("fit_ring: 6 or 7 inputs required (h, pid, A, C, with_projection[, thresh_outlier])");
        Csc A = csc_of(pin[3]);
        Traces Cm = traces_of(pin[4], A.K);
        int64_t info[4];
        CHECK(cnmfe_fit_ring_model(c, pid, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, nin > 6 ? mxGetScalar(pin[6]) : mxGetNaN(), mxGetScalar(pin[5]) != 0, NULL, info));
        pout[0] = info_of(info);
    } else if (!strcmp(cmd, "residual")) {
        if (nin != 5) FAIL("residual: 5 inputs required");
        Csc A = csc_of(pin[3]);
        Traces Cm = traces_of(pin[4], A.K);
        CHECK(cnmfe_residual(c, pid, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, NULL, CNMFE_HOST));
    } else if (!strcmp(cmd, "spatial")) {
        if (nin != 9) FAIL("spatial: 9 inputs required");
        char alg[16];
        if (mxGetString(pin[3], alg, sizeof(alg))) FAIL("spatial: bad algorithm name");
        const int a = !strcmp(alg, "hals") ? CNMFE_SPATIAL_HALS : !strcmp(alg, "hals_thresh") ? CNMFE_SPATIAL_HALS_THRESH : !strcmp(alg, "nnls") ? CNMFE_SPATIAL_NNLS : -1;
        if (a < 0) FAIL("spatial: unknown algorithm '%s'", alg);
        if (!mxIsSparse(pin[6])) FAIL("spatial: IND must be sparse");
        Csc A = csc_of(pin[4]), IND = csc_of(pin[6]);
        Traces Cm = traces_of(pin[5], A.K);
        float *sn = mxIsEmpty(pin[7]) ? NULL : f32_of(pin[7], NULL);
        float *out = (float *)mxCalloc((size_t)IND.nnz + 1, sizeof(float));
        CHECK(cnmfe_update_spatial(c, pid, a, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, IND.cp, IND.ri, sn, (int32_t)mxGetScalar(pin[8]), out));
        pout[0] = mxCreateSparse(mxGetM(pin[6]), (size_t)IND.K, (size_t)IND.nnz > 0 ? (size_t)IND.nnz : 1, mxREAL);       // same pattern as IND
        memcpy(mxGetJc(pout[0]), mxGetJc(pin[6]), ((size_t)IND.K + 1) * sizeof(mwIndex));
        memcpy(mxGetIr(pout[0]), mxGetIr(pin[6]), (size_t)IND.nnz * sizeof(mwIndex));
        for (int64_t i = 0; i < IND.nnz; ++i) mxGetPr(pout[0])[i] = out[i];
    } else if (!strcmp(cmd, "temporal")) {
        if (nin != 6) FAIL("temporal: 6 inputs required");
        Csc A = csc_of(pin[3]);
        Traces Cm = traces_of(pin[4], A.K);
        const size_t K = (size_t)A.K, T = mxIsInt32(pin[4]) ? 0 : mxGetN(pin[4]);
        if (nout > 0 && T == 0) FAIL("temporal: outputs need the trace matrix itself (row lists carry no T); use the stitch commands instead");
        float *Co = nout > 0 ? (float *)mxMalloc((K * T + 1) * sizeof(float)) : NULL, *Cr = nout > 1 ? (float *)mxMalloc((K * T + 1) * sizeof(float)) : NULL;
        float *aa = (float *)mxMalloc((K + 1) * sizeof(float));
        CHECK(cnmfe_hals_temporal(c, pid, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, (int32_t)mxGetScalar(pin[5]), Co, Cr, aa));
        if (nout > 0) pout[0] = to_double(Co, K, T);
        if (nout > 1) pout[1] = to_double(Cr, K, T);
        if (nout > 2) pout[2] = to_double(aa, K, 1);
    } else if (!strcmp(cmd, "temporal_job")) {                 // job = cnmfe_mex('temporal_job', h, pid, A_patch, C_or_rows, maxIter)
        if (nin != 6) FAIL("temporal_job: 6 inputs required");
        Csc A = csc_of(pin[3]);
        Traces Cm = traces_of(pin[4], A.K);
        int32_t job = -1;
        CHECK(cnmfe_hals_temporal_job(c, pid, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, (int32_t)mxGetScalar(pin[5]), NULL, NULL, &job));
        pout[0] = mxCreateDoubleScalar((double)job);
    } else if (!strcmp(cmd, "spatial_queue")) {                // q = cnmfe_mex('spatial_queue', h, pid, alg, A_patch, C_or_rows, IND_patch, sn, param)
        if (nin != 9) FAIL("spatial_queue: 9 inputs required");
        char alg[16];
        if (mxGetString(pin[3], alg, sizeof(alg))) FAIL("spatial_queue: bad algorithm name");
        const int a = !strcmp(alg, "hals") ? CNMFE_SPATIAL_HALS : !strcmp(alg, "hals_thresh") ? CNMFE_SPATIAL_HALS_THRESH : !strcmp(alg, "nnls") ? CNMFE_SPATIAL_NNLS : -1;
        if (a < 0) FAIL("spatial_queue: unknown algorithm '%s'", alg);
        if (!mxIsSparse(pin[6])) FAIL("spatial_queue: IND must be sparse");
        int q = 0;
        while (q < MAX_PEND && g_pend[q].used) ++q;
        if (q == MAX_PEND) FAIL("spatial_queue: too many updates queued and not collected");
        Csc A = csc_of(pin[4]), IND = csc_of(pin[6]);
        Traces Cm = traces_of(pin[5], A.K);
        float *sn = mxIsEmpty(pin[7]) ? NULL : f32_of(pin[7], NULL);
        CHECK(cnmfe_update_spatial(c, pid, a, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, IND.cp, IND.ri, sn, (int32_t)mxGetScalar(pin[8]), NULL));   // A_out == NULL: deferred
        Pending *P = &g_pend[q];
        P->pinned = (float *)cnmfe_host_alloc(((size_t)IND.nnz + 1) * sizeof(float));
        if (!P->pinned) FAIL("%s", cnmfe_last_error());
        const int rc = cnmfe_update_spatial_fetch_async(c, P->pinned, IND.nnz, &P->ticket);
        if (rc != 0) { cnmfe_host_free(P->pinned); P->pinned = NULL; CHECK(rc); }
        P->ind = mxDuplicateArray(pin[6]);
        mexMakeArrayPersistent(P->ind);
        P->c = c; P->nnz = IND.nnz; P->used = 1;
        pout[0] = mxCreateDoubleScalar((double)(q + 1));
    } else if (!strcmp(cmd, "temporal_deconv")) {
        if (nin != 9) FAIL("temporal_deconv: 9 inputs required (h, pid, A, C, maxIter, smin, max_tau, pars)");
        Csc A = csc_of(pin[3]);
        const size_t K = (size_t)A.K, T = mxGetN(pin[4]);
        if (mxIsInt32(pin[4])) FAIL("temporal_deconv: pass the trace matrix itself");
        Traces Cm = traces_of(pin[4], A.K);
        cnmfe_deconv_opts o; deconv_opts_of(&o, mxGetScalar(pin[6]), mxGetScalar(pin[7]));
        float *kp = (float *)mxCalloc(K + 1, sizeof(float));
        if (!mxIsEmpty(pin[8])) { size_t n; float *p0 = f32_of(pin[8], &n); if (n != K) FAIL("temporal_deconv: pars must have K entries"); memcpy(kp, p0, K * sizeof(float)); }
        float *Co = nout > 0 ? (float *)mxMalloc((K * T + 1) * sizeof(float)) : NULL, *Cr = nout > 1 ? (float *)mxMalloc((K * T + 1) * sizeof(float)) : NULL;
        float *S = nout > 2 ? (float *)mxMalloc((K * T + 1) * sizeof(float)) : NULL;
        float *sn = (float *)mxMalloc((K + 1) * sizeof(float)), *aa = (float *)mxMalloc((K + 1) * sizeof(float));
        CHECK(cnmfe_hals_temporal_deconv(c, pid, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, (int32_t)mxGetScalar(pin[5]), &o, kp, Co, Cr, S, sn, aa));
        if (nout > 0) pout[0] = to_double(Co, K, T);
        if (nout > 1) pout[1] = to_double(Cr, K, T);
        if (nout > 2) pout[2] = to_double(S, K, T);
        if (nout > 3) pout[3] = to_double(kp, K, 1);
        if (nout > 4) pout[4] = to_double(sn, K, 1);
        if (nout > 5) pout[5] = to_double(aa, K, 1);
    } else if (!strcmp(cmd, "fast_temporal")) {
        if (nin != 5) FAIL("fast_temporal: 5 inputs required (h, pid, A, T)");
        Csc A = csc_of(pin[3]);
        const size_t K = (size_t)A.K, T = (size_t)mxGetScalar(pin[4]);
        float *Cr = nout > 0 ? (float *)mxMalloc((K * T + 1) * sizeof(float)) : NULL, *aa = (float *)mxMalloc((K + 1) * sizeof(float));
        CHECK(cnmfe_fast_temporal(c, pid, A.K, A.cp, A.ri, A.v, CNMFE_COLMAJOR, Cr, aa));
        if (nout > 0) pout[0] = to_double(Cr, K, T);
        if (nout > 1) pout[1] = to_double(aa, K, 1);
    } else if (!strcmp(cmd, "compute_rss")) {
        if (nin != 7) FAIL("compute_rss: 7 inputs required (h, pid, A, C, b0_block, b0_new)");
        Csc A = csc_of(pin[3]);
        Traces Cm = traces_of(pin[4], A.K);
        float *bb = f32_of(pin[5], NULL), *bn = f32_of(pin[6], NULL);
        double rss = 0.0;
        CHECK(cnmfe_compute_rss(c, pid, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, bb, bn, &rss));
        pout[0] = mxCreateDoubleScalar(rss);
    } else if (!strcmp(cmd, "reconstruct_background")) {
        if (nin != 7) FAIL("reconstruct_background: 7 inputs required (h, pid, b0_block, b0_new, frame0, nframes)");
        size_t nb = 0, nn = 0;
        float *bb = f32_of(pin[3], &nb), *bn = f32_of(pin[4], &nn);
        const int64_t f0 = (int64_t)mxGetScalar(pin[5]), nf = (int64_t)mxGetScalar(pin[6]);
        if (nf <= 0) FAIL("reconstruct_background: nframes must be positive");
        pout[0] = mxCreateNumericMatrix(nn, (size_t)nf, mxSINGLE_CLASS, mxREAL);
        CHECK(cnmfe_reconstruct_background(c, pid, bb, bn, f0, nf, (float *)mxGetData(pout[0]), CNMFE_HOST));
    } else if (!strcmp(cmd, "set_noise")) {                    // cnmfe_mex('set_noise', h, pid, sn_block): read by the outlier branch of the ring fit
        if (nin != 4) FAIL("set_noise: 4 inputs required");
        CHECK(cnmfe_set_noise(c, pid, f32_of(pin[3], NULL)));
    } else if (!strcmp(cmd, "estimate_noise")) {               // sn_block = cnmfe_mex('estimate_noise', h, pid, d_b, nframes)   Sources2D.m:328-379 per pixel
        if (nin != 5) FAIL("estimate_noise: 5 inputs required");
        const size_t db = (size_t)mxGetScalar(pin[3]);
        float *sn = (float *)mxMalloc((db + 1) * sizeof(float));
        CHECK(cnmfe_estimate_noise(c, pid, (int64_t)mxGetScalar(pin[4]), sn));
        pout[0] = to_double(sn, db, 1);
    } else if (!strcmp(cmd, "background_ssub")) {              // cnmfe_mex('background_ssub', h, pid, fit_pid, bg_ssub, A_prev_block, C_or_rows, b0_block)
        if (nin != 8) FAIL("background_ssub: 8 inputs required");
        Csc A = csc_of(pin[5]);
        Traces Cm = traces_of(pin[6], A.K);
        CHECK(cnmfe_background_ssub(c, pid, (int)mxGetScalar(pin[3]), (int32_t)mxGetScalar(pin[4]), A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, f32_of(pin[7], NULL)));
    } else if (!strcmp(cmd, "compute_rss_ssub")) {             // rss = cnmfe_mex('compute_rss_ssub', h, pid, A_patch, C_or_rows, b0_new_patch)
        if (nin != 6) FAIL("compute_rss_ssub: 6 inputs required");
        Csc A = csc_of(pin[3]);
        Traces Cm = traces_of(pin[4], A.K);
        double rss = 0.0;
        CHECK(cnmfe_compute_rss_ssub(c, pid, A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, f32_of(pin[5], NULL), &rss));
        pout[0] = mxCreateDoubleScalar(rss);
    } else if (!strcmp(cmd, "reconstruct_background_ssub")) {  // Ybg = cnmfe_mex('reconstruct_background_ssub', h, pid, b0_new_patch, frame0, nframes)
        if (nin != 6) FAIL("reconstruct_background_ssub: 6 inputs required");
        size_t nn = 0;
        float *bn = f32_of(pin[3], &nn);
        const int64_t f0 = (int64_t)mxGetScalar(pin[4]), nf = (int64_t)mxGetScalar(pin[5]);
        if (nf <= 0) FAIL("reconstruct_background_ssub: nframes must be positive");
        pout[0] = mxCreateNumericMatrix(nn, (size_t)nf, mxSINGLE_CLASS, mxREAL);
        CHECK(cnmfe_reconstruct_background_ssub(c, pid, bn, f0, nf, (float *)mxGetData(pout[0]), CNMFE_HOST));
    } else if (!strcmp(cmd, "get_sn")) {
        if (nin != 4) FAIL("get_sn: 4 inputs required");
        const size_t d = (size_t)mxGetScalar(pin[3]);
        float *sn = (float *)mxMalloc((d + 1) * sizeof(float));
        CHECK(cnmfe_get_sn(c, pid, sn));
        pout[0] = to_double(sn, d, 1);
    } else if (!strcmp(cmd, "derive")) {
        if (nin != 6) FAIL("derive: 6 inputs required");
        char md[16];
        if (mxGetString(pin[5], md, sizeof(md))) FAIL("derive: bad mode");
        CHECK(cnmfe_patch_derive(c, pid, (int)mxGetScalar(pin[3]), (int32_t)mxGetScalar(pin[4]), !strcmp(md, "nearest") ? CNMFE_DERIVE_NEAREST : CNMFE_DERIVE_BICUBIC));
    } else if (!strcmp(cmd, "fit_ring_ssub")) {
        if (nin != 9 && nin != 10) FAIL("fit_ring_ssub: 9 or 10 inputs required");
        Csc A = csc_of(pin[6]);
        Traces Cm = traces_of(pin[7], A.K);
        int64_t info[4];
        CHECK(cnmfe_fit_ring_model_ssub(c, pid, (int)mxGetScalar(pin[3]), (int)mxGetScalar(pin[4]), (int32_t)mxGetScalar(pin[5]), A.K, A.cp, A.ri, A.v,
                                        Cm.ptr, Cm.order, nin > 9 ? mxGetScalar(pin[9]) : mxGetNaN(), mxGetScalar(pin[8]) != 0, info));
        pout[0] = info_of(info);
    } else if (!strcmp(cmd, "residual_ssub")) {
        if (nin != 7) FAIL("residual_ssub: 7 inputs required");
        Csc A = csc_of(pin[5]);
        Traces Cm = traces_of(pin[6], A.K);
        CHECK(cnmfe_residual_ssub(c, pid, (int)mxGetScalar(pin[3]), (int32_t)mxGetScalar(pin[4]), A.K, A.cp, A.ri, A.v, Cm.ptr, Cm.order, NULL, CNMFE_HOST));
    } else if (!strcmp(cmd, "get_ring")) {
        if (nin != 5) FAIL("get_ring: 5 inputs required (h, pid, d, d_b)");
        int64_t nnz; int32_t p;
        CHECK(cnmfe_ring_nnz(c, pid, &nnz, &p));
        const size_t d = (size_t)mxGetScalar(pin[3]), d_b = (size_t)mxGetScalar(pin[4]);
        int64_t *rp = (int64_t *)mxMalloc((d + 1) * sizeof(int64_t));
        int32_t *col = (int32_t *)mxMalloc(((size_t)nnz + 1) * sizeof(int32_t));
        float *val = (float *)mxMalloc(((size_t)nnz + 1) * sizeof(float)), *b0 = (float *)mxMalloc((d + 1) * sizeof(float));
        CHECK(cnmfe_ring_get_csr(c, pid, rp, col, val));
        CHECK(cnmfe_b0_get(c, pid, b0));
        // CSR (d x d_b) == CSC of the transpose: W' (d_b x d) directly, the caller transposes
        pout[0] = mxCreateSparse(d_b, d, (size_t)nnz > 0 ? (size_t)nnz : 1, mxREAL);
        for (size_t i = 0; i <= d; ++i) mxGetJc(pout[0])[i] = (mwIndex)rp[i];
        for (int64_t e = 0; e < nnz; ++e) { mxGetIr(pout[0])[e] = (mwIndex)col[e]; mxGetPr(pout[0])[e] = val[e]; }
        if (nout > 1) pout[1] = to_double(b0, d, 1);
    } else if (!strcmp(cmd, "set_ring")) {                      // Wt = W{m}.' with the ring's own pattern (what get_ring returned)
        if (nin != 5) FAIL("set_ring: 5 inputs required (h, pid, Wt, b0)");
        int64_t nnz; int32_t p;
        CHECK(cnmfe_ring_nnz(c, pid, &nnz, &p));
        if (!mxIsSparse(pin[3]) || (int64_t)mxGetJc(pin[3])[mxGetN(pin[3])] != nnz) FAIL("set_ring: Wt must be sparse with the ring's %lld entries", (long long)nnz);
        float *val = (float *)mxMalloc(((size_t)nnz + 1) * sizeof(float));
        const double *pr = mxGetPr(pin[3]);
        for (int64_t e = 0; e < nnz; ++e) val[e] = (float)pr[e];
        CHECK(cnmfe_ring_set_values(c, pid, val));
        float *b0 = f32_of(pin[4], NULL);
        CHECK(cnmfe_b0_set(c, pid, b0));
    } else FAIL("unknown command '%s'", cmd);
}
