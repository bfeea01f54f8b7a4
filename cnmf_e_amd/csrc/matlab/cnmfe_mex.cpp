// cnmfe_mex.cpp -- thin MEX gateway over the C ABI of include/cnmfe.h.
//
// NOT compiled in this repository's CI: it needs MATLAB's mex.h / matrix.h.  Build where MATLAB is installed:
//     mex -O -largeArrayDims cnmfe_mex.cpp -I../include -L../cnmf_e_amd -lcnmfe_hip
// House conventions follow the reference's own MEX (ca_source_extraction/utilities/graph_conn_comp_mex.cpp:38-64):
// argument checks first, mexErrMsgIdAndTxt on failure, outputs allocated with mxCreate*, inputs never written.
//
//   h = cnmfe_mex('create', device)
//   cnmfe_mex('patch', h, pid, patch_pos, block_pos, d1, d2, T)          % distribute_data.m:165-171 rectangles
//   cnmfe_mex('upload', h, pid, Yblock, t0)                               % Yblock: d_b x nt, any numeric class
//   cnmfe_mex('ring_init', h, pid, radius, num_neighbors)
//   [b0, info] = cnmfe_mex('fit_ring', h, pid, A_block, C_block, with_projection)   % fit_ring_model.m:1
//   Ysig = cnmfe_mex('residual', h, pid, A_prev_block, C_prev)            % update_spatial_parallel.m:162-166
//   A = cnmfe_mex('spatial', h, pid, alg, A_patch, C_patch, IND_patch, sn, param)   % HALS_spatial*.m / nnls_spatial.m
//   [C, C_raw, aa] = cnmfe_mex('temporal', h, pid, A_patch, C_patch, maxIter)       % HALS_temporal.m:1
//   keep = cnmfe_mex('postprocess', h, A, d1, d2)                         % post_process_spatial.m:19-32
//   [W, b0] = cnmfe_mex('get_ring', h, pid)
//   cnmfe_mex('destroy', h)
#include "mex.h"
#include "matrix.h"
#include <string.h>
#include <stdint.h>
#include <vector>
#include "cnmfe.h"

static std::vector<cnmfe_ctx *> g_ctx;
static void at_exit() { for (auto c : g_ctx) if (c) cnmfe_destroy(c); g_ctx.clear(); }
#define FAIL(...) mexErrMsgIdAndTxt("cnmfe_mex:error", __VA_ARGS__)
#define CHECK(rc) do { if ((rc) != 0) FAIL("%s", cnmfe_last_error()); } while (0)

static cnmfe_ctx *ctx_of(const mxArray *h) {
    size_t i = (size_t)mxGetScalar(h);
    if (i < 1 || i > g_ctx.size() || !g_ctx[i - 1]) FAIL("invalid context handle");
    return g_ctx[i - 1];
}
// sparse double (CSC) -> ABI arrays (int64 colptr, int32 rowidx, float val)
struct Csc { std::vector<int64_t> cp; std::vector<int32_t> ri; std::vector<float> v; int32_t K; };
static Csc csc_of(const mxArray *A) {
    Csc o;
    if (mxIsEmpty(A)) { o.K = 0; o.cp.assign(1, 0); return o; }
    mxArray *S = nullptr; const mxArray *src = A;
    if (!mxIsSparse(A)) { mxArray *in = const_cast<mxArray *>(A); mexCallMATLAB(1, &S, 1, &in, "sparse"); src = S; }
    o.K = (int32_t)mxGetN(src);
    const mwIndex *jc = mxGetJc(src), *ir = mxGetIr(src);
    o.cp.assign(jc, jc + o.K + 1);
    mwIndex nnz = jc[o.K];
    o.ri.resize(nnz); o.v.resize(nnz);
    if (mxIsLogical(src)) { for (mwIndex e = 0; e < nnz; ++e) { o.ri[e] = (int32_t)ir[e]; o.v[e] = 1.f; } }
    else { const double *pr = mxGetPr(src); for (mwIndex e = 0; e < nnz; ++e) { o.ri[e] = (int32_t)ir[e]; o.v[e] = (float)pr[e]; } }
    if (S) mxDestroyArray(S);
    return o;
}
static std::vector<float> f32_of(const mxArray *M) {
    size_t n = mxGetNumberOfElements(M);
    std::vector<float> o(n);
    if (mxIsSingle(M)) memcpy(o.data(), mxGetData(M), n * sizeof(float));
    else if (mxIsDouble(M)) { const double *p = mxGetPr(M); for (size_t i = 0; i < n; ++i) o[i] = (float)p[i]; }
    else FAIL("expected a single or double array");
    return o;
}
static mxArray *to_double(const std::vector<float> &v, size_t r, size_t c) {
    mxArray *o = mxCreateDoubleMatrix(r, c, mxREAL);
    double *p = mxGetPr(o);
    for (size_t i = 0; i < v.size(); ++i) p[i] = v[i];
    return o;
}

void mexFunction(int nout, mxArray *pout[], int nin, const mxArray *pin[]) {
    if (nin < 1 || !mxIsChar(pin[0])) FAIL("first argument must be a command string");
    char cmd[32]; mxGetString(pin[0], cmd, sizeof(cmd));
    mexAtExit(at_exit);
    if (!strcmp(cmd, "create")) {
        cnmfe_ctx *c = cnmfe_create(nin > 1 ? (int)mxGetScalar(pin[1]) : 0);
        if (!c) FAIL("%s", cnmfe_last_error());
        g_ctx.push_back(c);
        pout[0] = mxCreateDoubleScalar((double)g_ctx.size());
        return;
    }
    if (nin < 2) FAIL("missing context handle");
    cnmfe_ctx *c = ctx_of(pin[1]);
    if (!strcmp(cmd, "destroy")) { size_t i = (size_t)mxGetScalar(pin[1]); cnmfe_destroy(c); g_ctx[i - 1] = nullptr; return; }
    if (!strcmp(cmd, "postprocess")) {
        if (nin != 5) FAIL("postprocess: 5 inputs required");
        Csc A = csc_of(pin[2]);
        std::vector<uint8_t> keep(A.v.size());
        CHECK(cnmfe_post_process_spatial(c, (int32_t)mxGetScalar(pin[3]), (int32_t)mxGetScalar(pin[4]), A.K, A.cp.data(), A.ri.data(), A.v.data(), keep.data()));
        pout[0] = mxCreateLogicalMatrix(keep.size(), 1);
        mxLogical *k = mxGetLogicals(pout[0]);
        for (size_t i = 0; i < keep.size(); ++i) k[i] = keep[i] != 0;
        return;
    }
    if (nin < 3) FAIL("missing patch id");
    const int pid = (int)mxGetScalar(pin[2]);
    if (!strcmp(cmd, "patch")) {
        if (nin != 8) FAIL("patch: 8 inputs required");
        int32_t pr[4], br[4];
        for (int i = 0; i < 4; ++i) { pr[i] = (int32_t)mxGetPr(pin[3])[i]; br[i] = (int32_t)mxGetPr(pin[4])[i]; }
        CHECK(cnmfe_patch_create(c, pid, pr, br, (int32_t)mxGetScalar(pin[5]), (int32_t)mxGetScalar(pin[6]), (int64_t)mxGetScalar(pin[7])));
    } else if (!strcmp(cmd, "upload")) {
        if (nin != 5) FAIL("upload: 5 inputs required");
        const mxArray *Y = pin[3];
        int dt = mxIsSingle(Y) ? CNMFE_F32 : mxIsDouble(Y) ? CNMFE_F64 : mxIsUint16(Y) ? CNMFE_U16 : mxIsUint8(Y) ? CNMFE_U8 : -1;
        if (dt < 0) FAIL("upload: unsupported class %s", mxGetClassName(Y));
        CHECK(cnmfe_upload_block(c, pid, mxGetData(Y), dt, CNMFE_HOST, (int64_t)mxGetScalar(pin[4]), (int64_t)mxGetN(Y)));
    } else if (!strcmp(cmd, "ring_init")) {
        CHECK(cnmfe_ring_init(c, pid, (int32_t)mxGetScalar(pin[3]), nin > 4 && !mxIsEmpty(pin[4]) ? (int32_t)mxGetScalar(pin[4]) : 0));
    } else if (!strcmp(cmd, "fit_ring")) {
        if (nin != 6) FAIL("fit_ring: 6 inputs required");
        Csc A = csc_of(pin[3]);
        std::vector<float> C = A.K ? f32_of(pin[4]) : std::vector<float>();
        int64_t nnz; int32_t p; CHECK(cnmfe_ring_nnz(c, pid, &nnz, &p));
        int64_t info[4];
        // d is not known here without a query; b0 is fetched through get_ring
        CHECK(cnmfe_fit_ring_model(c, pid, A.K, A.cp.data(), A.ri.data(), A.v.data(), C.data(), CNMFE_COLMAJOR,
                                   mxGetNaN(), mxGetScalar(pin[5]) != 0, nullptr, info));
        pout[0] = mxCreateDoubleMatrix(1, 4, mxREAL);
        for (int i = 0; i < 4; ++i) mxGetPr(pout[0])[i] = (double)info[i];
    } else if (!strcmp(cmd, "residual")) {
        if (nin != 5) FAIL("residual: 5 inputs required");
        Csc A = csc_of(pin[3]);
        std::vector<float> C = A.K ? f32_of(pin[4]) : std::vector<float>();
        CHECK(cnmfe_residual(c, pid, A.K, A.cp.data(), A.ri.data(), A.v.data(), C.data(), CNMFE_COLMAJOR, nullptr, CNMFE_HOST));
    } else if (!strcmp(cmd, "spatial")) {
        if (nin != 9) FAIL("spatial: 9 inputs required");
        char alg[16]; mxGetString(pin[3], alg, sizeof(alg));
        int a = !strcmp(alg, "hals") ? CNMFE_SPATIAL_HALS : !strcmp(alg, "hals_thresh") ? CNMFE_SPATIAL_HALS_THRESH : CNMFE_SPATIAL_NNLS;
        Csc A = csc_of(pin[4]), IND = csc_of(pin[6]);
        std::vector<float> C = f32_of(pin[5]), sn = mxIsEmpty(pin[7]) ? std::vector<float>() : f32_of(pin[7]);
        std::vector<float> out(IND.v.size());
        CHECK(cnmfe_update_spatial(c, pid, a, A.K, A.cp.data(), A.ri.data(), A.v.data(), C.data(), CNMFE_COLMAJOR, IND.cp.data(), IND.ri.data(),
                                   sn.empty() ? nullptr : sn.data(), (int32_t)mxGetScalar(pin[8]), out.data()));
        pout[0] = mxCreateSparse(mxGetM(pin[6]), IND.K, out.size(), mxREAL);       // same pattern as IND
        memcpy(mxGetJc(pout[0]), mxGetJc(pin[6]), (IND.K + 1) * sizeof(mwIndex));
        memcpy(mxGetIr(pout[0]), mxGetIr(pin[6]), out.size() * sizeof(mwIndex));
        for (size_t i = 0; i < out.size(); ++i) mxGetPr(pout[0])[i] = out[i];
    } else if (!strcmp(cmd, "temporal")) {
        if (nin != 6) FAIL("temporal: 6 inputs required");
        Csc A = csc_of(pin[3]);
        std::vector<float> C = f32_of(pin[4]);
        size_t K = mxGetM(pin[4]), T = mxGetN(pin[4]);
        std::vector<float> Co(K * T), Cr(K * T), aa(K);
        CHECK(cnmfe_hals_temporal(c, pid, (int32_t)K, A.cp.data(), A.ri.data(), A.v.data(), C.data(), CNMFE_COLMAJOR,
                                  (int32_t)mxGetScalar(pin[5]), Co.data(), Cr.data(), aa.data()));
        pout[0] = to_double(Co, K, T);
        if (nout > 1) pout[1] = to_double(Cr, K, T);
        if (nout > 2) pout[2] = to_double(aa, K, 1);
    } else if (!strcmp(cmd, "temporal_deconv")) {            // [C, C_raw, S, pars, sn, aa] = cnmfe_mex('temporal_deconv', h, pid, A, C, maxIter, smin, max_tau)
        if (nin != 8) FAIL("temporal_deconv: 8 inputs required");
        Csc A = csc_of(pin[3]);
        std::vector<float> C = f32_of(pin[4]);
        size_t K = mxGetM(pin[4]), T = mxGetN(pin[4]);
        cnmfe_deconv_opts o; memset(&o, 0, sizeof(o));
        o.type = 1; o.method = 1; o.smin = mxGetScalar(pin[6]); o.max_tau = mxGetScalar(pin[7]); o.optimize_b = 1; o.optimize_pars = 1; o.maxIter = 10;
        std::vector<float> Co(K * T), Cr(K * T), S(K * T), kp(K), sn(K), aa(K);
        CHECK(cnmfe_hals_temporal_deconv(c, pid, (int32_t)K, A.cp.data(), A.ri.data(), A.v.data(), C.data(), CNMFE_COLMAJOR,
                                         (int32_t)mxGetScalar(pin[5]), &o, Co.data(), Cr.data(), S.data(), kp.data(), sn.data(), aa.data()));
        pout[0] = to_double(Co, K, T);
        if (nout > 1) pout[1] = to_double(Cr, K, T);
        if (nout > 2) pout[2] = to_double(S, K, T);
        if (nout > 3) pout[3] = to_double(kp, K, 1);
        if (nout > 4) pout[4] = to_double(sn, K, 1);
        if (nout > 5) pout[5] = to_double(aa, K, 1);
    } else if (!strcmp(cmd, "fast_temporal")) {              // [C_raw, aa] = cnmfe_mex('fast_temporal', h, pid, A, T)   (use_c_hat = false)
        if (nin != 5) FAIL("fast_temporal: 5 inputs required (h, pid, A, T)");
        Csc A = csc_of(pin[3]);
        int64_t info_T = (int64_t)mxGetScalar(pin[4]);
        std::vector<float> Cr((size_t)A.K * info_T), aa(A.K);
        CHECK(cnmfe_fast_temporal(c, pid, A.K, A.cp.data(), A.ri.data(), A.v.data(), CNMFE_COLMAJOR, Cr.data(), aa.data()));
        pout[0] = to_double(Cr, A.K, (size_t)info_T);
        if (nout > 1) pout[1] = to_double(aa, A.K, 1);
    } else if (!strcmp(cmd, "compute_rss")) {                // rss = cnmfe_mex('compute_rss', h, pid, A_patch, C, b0_block, b0_new_patch)   (Sources2D.m:1358-1510, one patch)
        if (nin != 7) FAIL("compute_rss: 7 inputs required (h, pid, A, C, b0_block, b0_new)");
        Csc A = csc_of(pin[3]);
        std::vector<float> C = A.K ? f32_of(pin[4]) : std::vector<float>(), bb = f32_of(pin[5]), bn = f32_of(pin[6]);
        double rss = 0.0;
        CHECK(cnmfe_compute_rss(c, pid, A.K, A.cp.data(), A.ri.data(), A.v.data(), C.data(), CNMFE_COLMAJOR, bb.data(), bn.data(), &rss));
        pout[0] = mxCreateDoubleScalar(rss);
    } else if (!strcmp(cmd, "reconstruct_background")) {     // Ybg = cnmfe_mex('reconstruct_background', h, pid, b0_block, b0_new_patch, frame0, nframes)   (Sources2D.m:1247-1355, one patch; d x nframes)
        if (nin != 7) FAIL("reconstruct_background: 7 inputs required (h, pid, b0_block, b0_new, frame0, nframes)");
        std::vector<float> bb = f32_of(pin[3]), bn = f32_of(pin[4]);
        const int64_t f0 = (int64_t)mxGetScalar(pin[5]), nf = (int64_t)mxGetScalar(pin[6]);
        if (nf <= 0) FAIL("reconstruct_background: nframes must be positive");
        std::vector<float> out(bn.size() * (size_t)nf);
        CHECK(cnmfe_reconstruct_background(c, pid, bb.data(), bn.data(), f0, nf, out.data(), CNMFE_HOST));
        pout[0] = to_double(out, bn.size(), (size_t)nf);
    } else if (!strcmp(cmd, "get_sn")) {                     // sn = cnmfe_mex('get_sn', h, pid, d)   (update_sn = true)
        if (nin != 4) FAIL("get_sn: 4 inputs required");
        size_t d = (size_t)mxGetScalar(pin[3]);
        std::vector<float> sn(d);
        CHECK(cnmfe_get_sn(c, pid, sn.data()));
        pout[0] = to_double(sn, d, 1);
    } else if (!strcmp(cmd, "derive")) {                     // cnmfe_mex('derive', h, pid, new_pid, bg_ssub, 'nearest'|'bicubic')
        if (nin != 6) FAIL("derive: 6 inputs required");
        char md[16]; mxGetString(pin[5], md, sizeof(md));
        CHECK(cnmfe_patch_derive(c, pid, (int)mxGetScalar(pin[3]), (int32_t)mxGetScalar(pin[4]), !strcmp(md, "nearest") ? CNMFE_DERIVE_NEAREST : CNMFE_DERIVE_BICUBIC));
    } else if (!strcmp(cmd, "fit_ring_ssub")) {              // info = cnmfe_mex('fit_ring_ssub', h, pid, fit_pid, res_pid, bg_ssub, A, C, with_projection)
        if (nin != 9) FAIL("fit_ring_ssub: 9 inputs required");
        Csc A = csc_of(pin[6]);
        std::vector<float> C = A.K ? f32_of(pin[7]) : std::vector<float>();
        int64_t info[4];
        CHECK(cnmfe_fit_ring_model_ssub(c, pid, (int)mxGetScalar(pin[3]), (int)mxGetScalar(pin[4]), (int32_t)mxGetScalar(pin[5]), A.K, A.cp.data(),
                                        A.ri.data(), A.v.data(), C.data(), CNMFE_COLMAJOR, mxGetNaN(), mxGetScalar(pin[8]) != 0, info));
        pout[0] = mxCreateDoubleMatrix(1, 4, mxREAL);
        for (int i = 0; i < 4; ++i) mxGetPr(pout[0])[i] = (double)info[i];
    } else if (!strcmp(cmd, "residual_ssub")) {              // cnmfe_mex('residual_ssub', h, pid, res_pid, bg_ssub, A_prev, C_prev)
        if (nin != 7) FAIL("residual_ssub: 7 inputs required");
        Csc A = csc_of(pin[5]);
        std::vector<float> C = A.K ? f32_of(pin[6]) : std::vector<float>();
        CHECK(cnmfe_residual_ssub(c, pid, (int)mxGetScalar(pin[3]), (int32_t)mxGetScalar(pin[4]), A.K, A.cp.data(), A.ri.data(), A.v.data(), C.data(),
                                  CNMFE_COLMAJOR, nullptr, CNMFE_HOST));
    } else if (!strcmp(cmd, "get_ring")) {
        int64_t nnz; int32_t p; CHECK(cnmfe_ring_nnz(c, pid, &nnz, &p));
        // rows = patch pixels: recovered from the CSR row pointer length the caller passes as pin[3] = d, pin[4] = d_b
        size_t d = (size_t)mxGetScalar(pin[3]), d_b = (size_t)mxGetScalar(pin[4]);
        std::vector<int64_t> rp(d + 1); std::vector<int32_t> col(nnz); std::vector<float> val(nnz), b0(d);
        CHECK(cnmfe_ring_get_csr(c, pid, rp.data(), col.data(), val.data()));
        CHECK(cnmfe_b0_get(c, pid, b0.data()));
        // CSR (d x d_b) == CSC of the transpose: build W' (d_b x d) directly, the caller transposes
        pout[0] = mxCreateSparse(d_b, d, nnz, mxREAL);
        for (size_t i = 0; i <= d; ++i) mxGetJc(pout[0])[i] = (mwIndex)rp[i];
        for (int64_t e = 0; e < nnz; ++e) { mxGetIr(pout[0])[e] = (mwIndex)col[e]; mxGetPr(pout[0])[e] = val[e]; }
        if (nout > 1) pout[1] = to_double(b0, d, 1);
    } else FAIL("unknown command '%s'", cmd);
}
