/*
 * cnmfe.h -- C ABI of the MI355X-native CNMF-E factor-update engine.
 *
 * This is the drop-in boundary for ONE path of zhoupc/CNMF_E: the alternating
 * ring-background -> spatial -> temporal update that
 * demos/demo_large_data_1p.m:199-201 runs through
 *   @Sources2D/update_background_parallel.m, update_spatial_parallel.m,
 *   update_temporal_parallel.m
 * and, per patch, through the plain MATLAB functions listed beside each entry
 * point below (file:line are into the reference tree).  A MEX gateway
 * (matlab/cnmfe_mex.cpp) or any FFI (ctypes: cnmf_e_amd/_lib.py) binds exactly
 * these symbols.  No C++/torch types cross the boundary.
 *
 * Conventions
 *   - every call returns 0 on success or a negative CNMFE_E* code;
 *     cnmfe_last_error() returns a thread-local message.  Nothing throws.
 *   - arrays are MATLAB-shaped: the video block is d_b x T column-major, i.e.
 *     frame after frame, each frame an nr_b x nc_b image with the ROW index
 *     fastest.  Pixel linear indices are 0-based inside the ABI.
 *   - sparse matrices are CSC (MATLAB's native layout: Jc = colptr, Ir = rowidx),
 *     64-bit column pointers, 32-bit 0-based row indices, rows sorted per column.
 *   - bulk values are float32 (the engine computes in fp32 with fp64 where it
 *     matters: means, the ring regression Gram/solve, NNLS); index arrays int32/int64.
 *   - the caller owns every pointer it passes; the context owns all device memory.
 *     "out" buffers are caller-allocated; a NULL out pointer means "keep the
 *     result resident only".
 *   - a context is single-threaded and bound to one GPU (one context per GPU,
 *     one process per GPU).
 */
#ifndef CNMFE_H
#define CNMFE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cnmfe_ctx cnmfe_ctx;

enum { CNMFE_OK = 0, CNMFE_EINVAL = -1, CNMFE_ENOMEM = -2, CNMFE_EHIP = -3,
       CNMFE_ESTATE = -4, CNMFE_EUNSUPPORTED = -5 };

/* element type of the video handed to cnmfe_upload_block */
enum { CNMFE_F32 = 0, CNMFE_F64 = 1, CNMFE_U16 = 2, CNMFE_U8 = 3, CNMFE_F16 = 4 };
/* where a bulk pointer lives */
enum { CNMFE_HOST = 0, CNMFE_DEVICE = 1 };
/* memory order of a K x T trace matrix */
enum { CNMFE_COLMAJOR = 0 /* MATLAB: element (k,t) at k + t*K */,
       CNMFE_ROWMAJOR = 1 /* numpy:  element (k,t) at k*T + t */,
       CNMFE_BOUND = 2,   /* the pointer is ignored: use the matrix bound with cnmfe_traces_bind */
       CNMFE_BOUND_ROWS = 3 /* the pointer is (const int32_t *) K ascending-or-not 0-based row indices into the bound matrix */ };

/* Bind a K x T trace matrix to the context: one host-to-device transfer, after which every entry point that takes
 * (C, c_order) accepts (NULL, CNMFE_BOUND) for that same matrix (K and T must match) and copies it on the device.
 * obj.C is the same K x T matrix for the background fit, both residual sweeps and the temporal HALS of one iteration
 * (update_background_parallel.m:130, update_temporal_parallel.m:86,91): bind it once, upload it once.  Trace OUTPUTS of a call made
 * with CNMFE_BOUND use the layout the matrix was bound with.  K = 0 unbinds.  A call that works on a SUBSET of the neurons (a patch's
 * C(ind,:), update_*_parallel.m) passes ((const float *)ind, CNMFE_BOUND_ROWS): the rows are gathered on the device, nothing crosses
 * PCIe.  C may be a host or a device pointer (row-major device matrices bind with a device-to-device copy). */
int cnmfe_traces_bind(cnmfe_ctx *ctx, int32_t K, int64_t T, const float *C, int c_order);
/* spatial algorithm (options.spatial_algorithm, CNMFSetParms.m:117) */
enum { CNMFE_SPATIAL_HALS = 0, CNMFE_SPATIAL_HALS_THRESH = 1, CNMFE_SPATIAL_NNLS = 2 };

const char *cnmfe_last_error(void);
const char *cnmfe_version(void);

/* ---- context ------------------------------------------------------------ */
cnmfe_ctx *cnmfe_create(int device);          /* NULL on failure (see cnmfe_last_error) */
void       cnmfe_destroy(cnmfe_ctx *ctx);

/* ---- data plane: one resident block per patch ----------------------------
 * Replaces get_patch_data(mat_data, tmp_patch, frame_range, true)
 * (endoscope/get_patch_data.m:50-93) at update_background_parallel.m:208,
 * update_spatial_parallel.m:147, update_temporal_parallel.m:137: the block is
 * uploaded once and stays in HBM.
 * patch_rect/block_rect = [r0 r1 c0 c1], 1-based inclusive, exactly
 * mat_data.patch_pos{m} / block_pos{m} (distribute_data.m:165-171). */
int cnmfe_patch_create(cnmfe_ctx *ctx, int patch_id, const int32_t patch_rect[4],
                       const int32_t block_rect[4], int32_t d1, int32_t d2, int64_t T);
/* frames [t0, t0+nt) of the block, nt x d_b elements of `dtype`, frame-major */
int cnmfe_upload_block(cnmfe_ctx *ctx, int patch_id, const void *Y, int dtype, int memspace,
                       int64_t t0, int64_t nt);
/* P.Ymean{m} (initComponents_parallel.m:338-339): temporal mean of the BLOCK, d_b doubles */
int cnmfe_get_ymean(cnmfe_ctx *ctx, int patch_id, double *ymean_block);

/* ---- B0: ring pattern -----------------------------------------------------
 * get_nhood(radius, num_neighbors) (endoscope/get_nhood.m:1-25) + the W builder
 * at initComponents_parallel.m:213-236: fixed ring sparsity pattern, W = 1/count,
 * b0 = 0.  num_neighbors <= 0 means [] (all ring pixels). */
int cnmfe_ring_init(cnmfe_ctx *ctx, int patch_id, int32_t radius, int32_t num_neighbors);
int cnmfe_ring_nnz(cnmfe_ctx *ctx, int patch_id, int64_t *nnz, int32_t *p);
/* W{m} as CSR (d x d_b): rowptr[d+1], col[nnz] (block pixel, ascending), val[nnz] */
int cnmfe_ring_get_csr(cnmfe_ctx *ctx, int patch_id, int64_t *rowptr, int32_t *col, float *val);
int cnmfe_ring_set_values(cnmfe_ctx *ctx, int patch_id, const float *val /* nnz, CSR order */);
/* the reference's first-run test, by VALUE inspection of row 1 of W{m}:
 * length(unique(W_old(1,:)))==2   (fit_ring_model.m:25, update_background_parallel.m:143) */
int cnmfe_ring_first_run(cnmfe_ctx *ctx, int patch_id, int *first_run);
int cnmfe_b0_get(cnmfe_ctx *ctx, int patch_id, float *b0 /* d */);
int cnmfe_b0_set(cnmfe_ctx *ctx, int patch_id, const float *b0 /* d */);

/* ---- B1/B2: [W, b0] = fit_ring_model(Y, A, C, W_old, thresh_outlier, sn, ind_patch, with_projection)
 * endoscope/fit_ring_model.m:1-127.  Y = the resident block, W_old = the resident W{m}
 * (first-run detection by value inspection of row 1, :25; ind_active, :28; frame stride k,
 * :60,84-87).  A is d_b x K CSC (block rows), C is K x T.  thresh_outlier = NaN is what every demo runs.  A finite value takes the
 * outlier branch (:50-56: entries of the patch rows above W_old*Bf + thresh_outlier*sn are replaced by W_old*Bf; :62-67: only the frames
 * with at most the nmax/T quantile of outliers are regressed on) -- it needs the noise levels of the block (cnmfe_set_noise, else
 * CNMFE_ESTATE) and computes the Gram of the clipped residual directly (the kept video table does not apply); info[1] is 1 then.
 * info[0]=first_run, info[1]=frame stride k, info[2]=#active pixels (-1 when the call returns without waiting for the device -- b0_out ==
 * NULL on a later run), info[3]=pmax.  Nothing about W_old is asked of the device at call time: pmax and row 1 were copied to pinned memory
 * behind the call that wrote W, so a fit does not drain the stream before it queues its kernels.
 * With b0_out == NULL the call returns while the Gram / solve kernels are still running on the context's stream; every later
 * call on this context is ordered behind them and reports their errors.
 * The first fit of a patch (and the first one after its frame stride k changes) also builds the block-pair covariance table of the
 * resident VIDEO and keeps it with the patch (fp64, 58 bytes per block pixel and needed neighbour pair: 3.8 GB for 512 x 512,
 * radius 15) until the next cnmfe_upload_block; later fits only add the footprint corrections (option "gram_incremental", default 1;
 * 0 = the Gram of Y - A*C from scratch every call).  Results are the same regression either way. */
int cnmfe_fit_ring_model(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr,
                         const int32_t *A_rowidx, const float *A_val, const float *C, int c_order,
                         double thresh_outlier, int with_projection,
                         float *b0_out /* d or NULL */, int64_t info[4]);

/* Diagnostics of the last fit's ring solve when it ran out of the cached inverses (option solve_inv, ring_solve_inv.hpp; replaces nothing in the reference --
 * fit_ring_model.m:106 solves every pixel from scratch): out[0] pixels the fast path left to the factorising kernel (more than 8 neurons around the ring, a
 * ridge series that did not converge), out[3] those among them whose inverse was rebuilt; with solve_probe bit 512 also out[1] ridge-series terms taken over
 * all pixels, out[2] pixels that took at least one.  All -1 when the patch has no inverses.  Synchronises the context's stream. */
int cnmfe_ring_solve_stats(cnmfe_ctx *ctx, int patch_id, int64_t out[4]);

/* Optional, once per patch that will be FITTED (after cnmfe_ring_init; with bg_ssub > 1 that is the low-resolution fit patch): allocate the ring fit's large
 * device buffers now -- the block-pair covariance tables, the tiled residual, the window projection's partial sums (18 GB for 512 x 512 x 10000, radius 15) --
 * so that the first cnmfe_fit_ring_model queues its kernels without a hipMalloc in between.  Sizes follow the geometry only; nothing is computed.  The host
 * mirror calls it while it sets the patches up (initComponents_parallel.m:213-236 is where the reference allocates W, b0); a host that does not simply gets
 * the allocations inside its first fit. */
int cnmfe_fit_reserve(cnmfe_ctx *ctx, int patch_id);

/* P.sn: sn = estimate_noise(obj, frame_range, 'psd')  (@Sources2D/Sources2D.m:328-379 -> OASIS_matlab/functions/GetSn.m:19-46) for the block
 * pixels of a patch: the Welch estimate of the first `nframes` frames (the reference's default is min(T, 3000)) of the resident RAW video
 * (pixel mean included, as pwelch sees it).  The per-storage-block bookkeeping of :361-375 (row / column end-1 of every block but the
 * last is dropped) is index arithmetic on the assembled image and stays with the host (sources2d.estimate_noise_image). */
int cnmfe_estimate_noise(cnmfe_ctx *ctx, int patch_id, int64_t nframes, float *sn_block_out /* d_b */);

/* sn of the BLOCK pixels of a patch (obj.P.sn(logical(mask)), update_background_parallel.m:131; for a low-resolution fit patch of
 * bg_ssub > 1 the resized values of :137).  Only the outlier branch of the ring fit reads them. */
int cnmfe_set_noise(cnmfe_ctx *ctx, int patch_id, const float *sn_block /* d_b */);

/* ---- bg_ssub > 1: the ring model on a spatially downsampled block
 * W lives on the ceil(nr_b/s) x ceil(nc_b/s) grid with ring radius ceil(r/s)  (@Sources2D/initComponents_parallel.m:214,237-251).
 * cnmfe_patch_derive creates a patch whose FOV is that grid and whose resident video is computed on the device from the
 * source patch: CNMFE_DERIVE_NEAREST = imresize(., 1/s, 'nearest') (what fit_ring_model sees, update_background_parallel.m:224),
 * CNMFE_DERIVE_BICUBIC = imresize(., 1/s) of the centred video (what W multiplies, update_spatial_parallel.m:171-172).
 * Call cnmfe_ring_init(new_patch, ceil(r/s), num_neighbors) on both afterwards. */
enum { CNMFE_DERIVE_NEAREST = 0, CNMFE_DERIVE_BICUBIC = 1 };
int cnmfe_patch_derive(cnmfe_ctx *ctx, int src_patch, int new_patch, int32_t ssub, int mode);

/* update_background_parallel.m:219-230:  b0 = mean(Y - A*C, 2)(patch);  W = fit_ring_model(imresize(Y - A*C, 1/s, 'nearest'), [], [], W_old, ...).
 * A is d_b x K CSC over BLOCK rows of patch_id; W is fitted on fit_patch and copied to res_patch; b0 is set on patch_id.  info as in
 * cnmfe_fit_ring_model (of the low-resolution fit). */
int cnmfe_fit_ring_model_ssub(cnmfe_ctx *ctx, int patch_id, int fit_patch, int res_patch, int32_t ssub, int32_t K,
                              const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val, const float *C, int c_order,
                              double thresh_outlier, int with_projection, int64_t info[4]);

/* update_spatial_parallel.m:167-178 == update_temporal_parallel.m:153-165:
 *   Ysig = Y(ind_patch,:) - imresize(W * imresize(R - mean(R,2), 1/s), [nr_block nc_block])(ind_patch,:) - b0,  R = Y - A_prev*C_prev.
 * The result stays resident as the Ysig of patch_id (for cnmfe_update_spatial / cnmfe_hals_temporal / cnmfe_get_sn).
 * Round 5: with Ysig_out == NULL the request is only RECORDED (as cnmfe_residual does since round 4; option ssub_virtual): cnmfe_update_spatial and
 * cnmfe_hals_temporal take Ysig*C' and A'*Ysig through the two resampling maps from the video rows under the masks / footprints and the low-resolution video of
 * res_patch (fp64 sums); every other consumer runs the low-resolution sweep and its upsample first.  res_patch must stay alive until the next fit / upload. */
int cnmfe_residual_ssub(cnmfe_ctx *ctx, int patch_id, int res_patch, int32_t ssub, int32_t Ksel, const int64_t *A_colptr,
                        const int32_t *A_rowidx, const float *A_val, const float *C, int c_order,
                        float *Ysig_out /* d x T or NULL */, int out_memspace);

/* ---- R1: the residual / background-subtraction expression
 *   Ysig = Y(ind_patch,:) - W*(Y - A_prev*C_prev) - (b0 - W*mean(Y - A_prev*C_prev, 2))
 * update_spatial_parallel.m:162-166 == update_temporal_parallel.m:149-152.
 * A_prev is d_b x Ksel CSC (block rows), C_prev Ksel x T.  The result (d x T fp32,
 * frame-major) stays resident for the HALS/NNLS calls below; Ysig_out may be NULL, and then
 * the call returns with the sweep still running on the context's stream (all inputs have been
 * consumed; later calls on this context are stream-ordered behind it and report its errors).
 * Ysig stays resident PER PATCH.  While the video, W and b0 of the patch are unchanged, a further call only changes the footprint
 * term (W*A_prev)*(C_prev - mean): the difference is folded into the resident Ysig in one streaming pass (option "r1_delta",
 * default 1), or merely recorded (option "r1_lazy", default 1) -- cnmfe_hals_temporal[_deconv] then adds A'*(W*A_prev)*(C_prev - mean)
 * to A'*Ysig without another pass over the video, every other consumer folds it in first.  The values any caller sees are those of
 * the full expression above (up to fp32 rounding of the re-association).
 * Option "r1_defer" (default 1; ring radius 15, no Ysig_out): the FIRST residual after a fit also runs its sweep without the footprint term and leaves
 * the term pending -- cnmfe_update_spatial adds (W*A_prev)*((C_prev - mean)*(C - mean)') to its projection Ysig*C' on the search mask, so patches with
 * halo neurons are swept by the same (fastest) kernel as a patch without. */
int cnmfe_residual(cnmfe_ctx *ctx, int patch_id, int32_t Ksel, const int64_t *A_colptr,
                   const int32_t *A_rowidx, const float *A_val, const float *C, int c_order,
                   float *Ysig_out, int out_memspace);

/* ---- S5: sn = GetSn(Ysig) of every patch pixel (update_sn = true)
 * @Sources2D/update_spatial_parallel.m:191-194 -> OASIS_matlab/functions/GetSn.m:33-47
 * (Welch PSD with pwelch's defaults, sn = sqrt(exp(mean(log(psd/2)))) over 0.25 <= f <= 0.5).
 * Ysig = the resident residual of this patch (cnmfe_residual must have been called).  sn_out: d floats. */
int cnmfe_get_sn(cnmfe_ctx *ctx, int patch_id, float *sn_out);

/* ---- S1-S4: A = HALS_spatial(Y,A,C,active_pixel,maxIter)            utilities/HALS_spatial.m:1-45
 *             A = HALS_spatial_thresh(Y,A,C,active_pixel,maxIter,sn)  utilities/HALS_spatial_thresh.m:1-53
 *             A = nnls_spatial(Y,A,C,active_pixel,maxN)               endoscope/nnls_spatial.m:1-109
 * Y = the resident Ysig of this patch (cnmfe_residual must have been called).
 * A is d x K CSC over PATCH rows, IND (active_pixel) d x K CSC pattern, sn d floats
 * (HALS_THRESH only).  param = maxIter (HALS*) or maxN (NNLS).  The result has exactly
 * the IND pattern: A_out[nnz(IND)] in IND's CSC order (entries may be 0).
 * A_out == NULL defers the download: the call returns with the sweeps queued and cnmfe_update_spatial_fetch(ctx, A_out, nnz(IND)) collects the
 * values later (valid until the next spatial update of this context) -- the host mirror sets up the temporal update's residual in between. */
int cnmfe_update_spatial(cnmfe_ctx *ctx, int patch_id, int algorithm, int32_t K,
                         const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                         const float *C, int c_order,
                         const int64_t *IND_colptr, const int32_t *IND_rowidx,
                         const float *sn, int32_t param, float *A_out);
int cnmfe_update_spatial_fetch(cnmfe_ctx *ctx, float *A_out, int64_t nnz);
/* the same fetch without the wait: the copy into PINNED host memory (cnmfe_host_alloc; a kernel writes it through the mapping, in 16-byte units: room for nnz floats
 * rounded up to 16 bytes) is queued behind the sweeps and *ticket names the point of the
 * stream where it is complete; cnmfe_ticket_wait(ctx, ticket) waits for exactly that point (not for what was queued after it) and releases the ticket.
 * With several patches per context the host mirror queues patch m + 1 before it collects patch m, so the device never waits for the host's assembly
 * of A (update_spatial_parallel.m:324-334).  A kernel-raised error is reported by the next waiting call of the context, not by the ticket. */
int cnmfe_update_spatial_fetch_async(cnmfe_ctx *ctx, float *A_out_pinned, int64_t nnz, int64_t *ticket);
int cnmfe_ticket_wait(cnmfe_ctx *ctx, int64_t ticket);
/* the same fetch with post_process_spatial's connectivity constraint (cnmfe_post_process_spatial below) applied to the result where it lies on the
 * device: for a patch that IS the d1 x d2 field of view (patch rows = FOV pixels), IND = the mask the deferred update ran on.  A_out = the raw update
 * (obj.A before post-processing), keep_out[e] = 1 where entry e survives. */
int cnmfe_update_spatial_fetch_connected(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                                         float *A_out, uint8_t *keep_out);

/* the connected fetch without the wait: the connectivity kernel and the two copies into PINNED memory (cnmfe_host_alloc) are queued, *ticket names the point of
 * the stream where they are complete (cnmfe_ticket_wait).  The host mirror queues the temporal update's residual request behind it -- whatever that request starts
 * on the device (the deferred half of the ring solve, the W*A_prev tables) then runs while the host turns the result into the next call's A. */
int cnmfe_update_spatial_fetch_connected_async(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                                               float *A_out_pinned, uint8_t *keep_out_pinned, int64_t *ticket);

/* ---- fast_temporal (use_c_hat = false)                @Sources2D/update_temporal_parallel.m:174-175,314-337
 *   tmp_A = A .* (A ./ max(A,[],1) >= 0.5);  aa = sum(tmp_A.^2,1);  C_raw = (tmp_A' * Ysig) ./ aa'
 * (rows with aa == 0 are 0 and report aa = 0).  A is d x K CSC over PATCH rows; Ysig = the resident
 * residual of this patch.  C_raw_out K x T (NULL: keep the result on the device for cnmfe_stitch_add), aa_out K floats (may be NULL). */
int cnmfe_fast_temporal(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr,
                        const int32_t *A_rowidx, const float *A_val, int c_order,
                        float *C_raw_out, float *aa_out);

/* ---- T1-T3: [C, C_raw, ~, ~] = HALS_temporal(Y, A, C, maxIter, [])   utilities/HALS_temporal.m:1-119
 * (no-deconvolution branch :64-68).  Y = resident Ysig.  A d x K CSC over PATCH rows.
 * Outputs in c_order (each may be NULL: C_raw and aa also stay on the device for cnmfe_stitch_add); aa_out[k] = sum(A(:,k).^2)
 * (update_temporal_parallel.m:181). */
int cnmfe_hals_temporal(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr,
                        const int32_t *A_rowidx, const float *A_val, const float *C_in, int c_order,
                        int32_t maxIter, float *C_out, float *C_raw_out, float *aa_out);

/* ---- T4/T6: deconvolution (options.deconv_flag = true) ---------------------------------------
 * deconv_options of demos/demo_large_data_1p.m:38-43; only type 'ar1' + method 'foopsi' is built. */
typedef struct cnmfe_deconv_opts {
    int32_t type;            /* 1 = 'ar1' */
    int32_t method;          /* 1 = 'foopsi' */
    double  smin;            /* negative: |smin| * noise level (deconvolveCa.m:116-118) */
    double  lambda;          /* must be 0 */
    double  max_tau;         /* gmax = exp(-1/max_tau) (deconvolveCa.m:119) */
    int32_t optimize_b;
    int32_t optimize_pars;
    int32_t maxIter;         /* deconvTemporal only (default 10); HALS_temporal forces 20 (HALS_temporal.m:92) */
} cnmfe_deconv_opts;

/* [C, C_raw, results_deconv] = HALS_temporal(Y, A, C, maxIter, deconv_options)   utilities/HALS_temporal.m:70-104
 * (the deconvolution branch: per row GetSn + deconvolveCa inside the Gauss-Seidel sweep).
 * kernel_pars[K] in/out (0 = not yet estimated), S_out / sn_out receive results_deconv.S / .sn.
 * With C_out, C_raw_out, S_out and sn_out all NULL nothing is copied back -- kernel_pars is then input only -- and the call returns with the sweeps in
 * flight (C_raw and aa stay on the device for cnmfe_stitch_add). */
int cnmfe_hals_temporal_deconv(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr,
                               const int32_t *A_rowidx, const float *A_val, const float *C_in, int c_order,
                               int32_t maxIter, const cnmfe_deconv_opts *opts, float *kernel_pars,
                               float *C_out, float *C_raw_out, float *S_out, float *sn_out, float *aa_out);
/* obj.deconvTemporal()  (@Sources2D/deconvTemporal.m:29-105): every row of C_raw (K x T, in/out: the fitted
 * baseline is subtracted) is deconvolved with a fresh time-constant estimate. */
int cnmfe_deconv_temporal(cnmfe_ctx *ctx, int32_t K, int64_t T, float *C_raw, int c_order,
                          const cnmfe_deconv_opts *opts, float *C_out, float *S_out,
                          float *kernel_pars_out, float *sn_out);
/* The same on the BOUND trace matrix (cnmfe_traces_bind, cnmfe_stitch_finish*): the stitched C_raw is deconvolved where it lies, the denoised C becomes
 * the bound matrix (what the next background / spatial / temporal update reads with (NULL, CNMFE_BOUND)), C_raw - b and S stay in the context -- no
 * K x T array crosses PCIe on the critical path.  The host outputs (row-major K x T / K floats, PINNED memory from cnmfe_host_alloc, any may be NULL) are
 * written by a second stream behind the kernels: cnmfe_stitch_wait waits for them.  CNMFE_ESTATE without a bound matrix. */
int cnmfe_deconv_temporal_bound(cnmfe_ctx *ctx, const cnmfe_deconv_opts *opts, float *C_out, float *C_raw_out, float *S_out,
                                float *kernel_pars_out, float *sn_out);

/* ---- T5: the overlap-region stitch of the temporal update, on the device
 * @Sources2D/update_temporal_parallel.m:264-280:
 *     C_new(ind_m, :) += aa_m .* C_raw_m;  aa(ind_m) += aa_m   over the patches m;   C_raw = C_new ./ aa  (aa == 0 -> 1);
 *     without deconvolution  C_raw = C_raw - min(C_raw, [], 2),  C = C_raw   (:285-286)
 * cnmfe_hals_temporal[_deconv] / cnmfe_fast_temporal leave their C_raw rows and aa on the device (their host outputs may be NULL); the
 * stitch accumulates them there, so the K_m x T pieces never cross PCIe:
 *   cnmfe_stitch_begin(ctx, K, T)             zero the accumulator: K rows of (T + weight) floats
 *   cnmfe_stitch_add(ctx, K_m, ind_m)         after each patch's temporal call: rows ind_m (0-based, distinct) += aa_m .* C_raw_m
 *   cnmfe_stitch_temporal(ctxs, n, ...)       the exchange + :279-286.  n contexts of ONE process (one per GPU, each holding the sum over its own
 *                                             patches): RCCL all-reduce (sum) of the accumulators over xGMI, in place; every context then
 *                                             divides, subtracts the row minima if asked, and BINDS the result as its trace matrix
 *                                             (cnmfe_traces_bind semantics, row-major): the next background / spatial update reads it with
 *                                             (NULL, CNMFE_BOUND).  C_raw_out (K x T, c_order; may be NULL) receives a host copy from ctxs[0].
 *   one process PER GPU (torch.distributed, MPI): cnmfe_stitch_buffer hands out the accumulator's device address for the caller's own
 *   all-reduce (count = K * ld floats), cnmfe_stitch_finish does :279-286 + bind on this context. */
int cnmfe_stitch_begin(cnmfe_ctx *ctx, int32_t K, int64_t T);
int cnmfe_stitch_add(cnmfe_ctx *ctx, int32_t K_m, const int32_t *ind_m);
/* Several patches per context: cnmfe_hals_temporal_job does everything cnmfe_hals_temporal[_deconv] does up to the Gauss-Seidel sweeps (opts == NULL: the
 * no-deconvolution branch; kernel_pars: K time constants in, with opts) and keeps the patch's buffers as job *job_out (numbered from cnmfe_stitch_begin);
 * cnmfe_temporal_jobs_sweep runs level l of EVERY job in one launch -- the patches are independent (update_temporal_parallel.m:112-186 is a parfor), and one
 * patch's level is a handful of workgroups as long as one trace's work; cnmfe_stitch_add_job(job, ...) then adds that job's aa .* C_raw to the accumulator.
 * Same values as the per-patch calls (the same kernels on the same operands). */
int cnmfe_hals_temporal_job(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                            const float *C_in, int c_order, int32_t maxIter, const cnmfe_deconv_opts *opts, const float *kernel_pars, int32_t *job_out);
int cnmfe_temporal_jobs_sweep(cnmfe_ctx *ctx);
int cnmfe_stitch_add_job(cnmfe_ctx *ctx, int32_t job, int32_t K_m, const int32_t *ind_m);
int cnmfe_stitch_buffer(cnmfe_ctx *ctx, float **dev_acc, int64_t *ld);
/* cnmfe_stitch_buffer without the drain of the stream: also hands out the hipStream_t the additions are queued on.  A collective enqueued ON that stream (torch:
 * `with torch.cuda.stream(torch.cuda.ExternalStream(ptr)): all_reduce(...)` -- RCCL orders itself behind and in front of the current stream with events) needs
 * no host wait on either side; cnmfe_stitch_finish* follows in stream order. */
int cnmfe_stitch_buffer_stream(cnmfe_ctx *ctx, float **dev_acc, int64_t *ld, void **hip_stream);
int cnmfe_stitch_dims(cnmfe_ctx *ctx, int32_t *K, int64_t *T);   /* the K x T recorded by cnmfe_stitch_begin (CNMFE_ESTATE if no stitch is open): a gateway sizes C_raw_out from these, not from its caller */
int cnmfe_stitch_finish(cnmfe_ctx *ctx, int subtract_min, float *C_raw_out, int c_order);
int cnmfe_stitch_temporal(cnmfe_ctx *const *ctxs, int n, int subtract_min, float *C_raw_out, int c_order);
/* cnmfe_stitch_finish without the wait: the host copy (row-major K x T) is written by a second stream into PINNED memory (cnmfe_host_alloc) and is
 * complete after cnmfe_synchronize(ctx); the call itself returns at once, so the caller can set up the next background fit -- which takes the
 * traces from the device binding this call leaves -- while the temporal kernels and the download are still running. */
int cnmfe_stitch_finish_async(cnmfe_ctx *ctx, int subtract_min, float *C_raw_pinned);
int cnmfe_stitch_wait(cnmfe_ctx *ctx);         /* waits for the downloads of cnmfe_stitch_finish_async only (not for the compute stream) */
/* every asynchronous download batch (cnmfe_stitch_finish_async, cnmfe_deconv_temporal_bound) has a generation number: cnmfe_copy_generation after the call
 * that queued it, cnmfe_copy_wait(ctx, gen) waits for THAT batch only -- a host that releases last iteration's buffers must not wait for this iteration's. */
int cnmfe_copy_generation(cnmfe_ctx *ctx, int64_t *gen);
int cnmfe_copy_wait(cnmfe_ctx *ctx, int64_t gen);
void *cnmfe_host_alloc(size_t bytes);          /* page-locked host memory (NULL + cnmfe_last_error on failure) */
void cnmfe_host_free(void *p);

/* ---- host helper: the reference's sparse row selections `A(mask, ind)` with `ind = find(sum(A(mask, :), 1) > 0)` (@Sources2D/update_spatial_parallel.m:87-91,96-97,
 * update_temporal_parallel.m:136-141, update_background_parallel.m:129-131), which MATLAB's sparse indexing does natively and a host in another language does not.
 * CSC in (colptr / rowidx / val of the whole-FOV matrix), `lut[pixel]` = row inside the selection or -1, `cand` = ascending candidate columns.
 * keep_all = 0: a candidate is kept when its selected values sum to > 0 (double accumulation in storage order); 1: every candidate is kept.
 * Out: out_ind[0..*nkept) the kept columns, out_colptr[0..*nkept], their entries (row = lut value, in storage order) in out_rowidx / out_val (capacity `cap`
 * entries: the candidates' total nnz always suffices).  No device involved; CNMFE_EINVAL on a null argument, unsorted candidates or too small a capacity. */
int cnmfe_csc_select_rows(const int64_t *colptr, const int32_t *rowidx, const float *val, const int32_t *lut, int64_t ncand, const int64_t *cand,
                          int keep_all, int64_t cap, int64_t *out_ind, int64_t *out_colptr, int32_t *out_rowidx, float *out_val, int64_t *nkept);
/* host helper: the CSC matrix (ncol columns) without its stored zeros -- and, with keep != NULL, without the entries whose flag is 0 (the connectivity
 * flags of cnmfe_update_spatial_fetch_connected).  A spatial update returns values on the search mask's pattern, most of them zero; MATLAB's sparse
 * matrices drop them on assignment (update_spatial_parallel.m:324-334).  Outputs sized for nnz(in) always suffice; *nnz_out = entries written. */
/* sources2d.py rows_of twice in one pass (update_temporal_parallel.m:83-91: ind = find(sum(A(block,:),1) > 0), A(block, ind), A(patch, ind)): the candidates whose
 * BLOCK entries sum to > 0 as out_ind, their block entries (local rows lut_block) and their patch entries (lut_patch) as two CSC matrices over the same columns */
int cnmfe_csc_select_block_patch(const int64_t *colptr, const int32_t *rowidx, const float *val, const int32_t *lut_block, const int32_t *lut_patch, int64_t ncand,
                                 const int64_t *cand, int64_t cap, int64_t *out_ind, int64_t *blk_colptr, int32_t *blk_rowidx, float *blk_val,
                                 int64_t *pat_colptr, int32_t *pat_rowidx, float *pat_val, int64_t *nkept);
/* the non-empty columns of a sorted CSC footprint matrix over a d1-row image and the bounding boxes of their entries (0-based image rows / columns): the prefilter
 * of the mask selections of update_*_parallel.m (a neuron whose box misses a patch's block cannot be selected there) */
int cnmfe_csc_bbox(int32_t K, int32_t d1, const int64_t *colptr, const int32_t *rowidx, int64_t *nz, int32_t *rmin, int32_t *rmax, int32_t *cmin, int32_t *cmax, int64_t *nnz_cols);
int cnmfe_csc_drop_zeros(int32_t ncol, const int64_t *colptr, const int32_t *rowidx, const float *val, const uint8_t *keep,
                         int64_t *out_colptr, int32_t *out_rowidx, float *out_val, int64_t *nnz_out);

/* host helper: sum, centre of mass (utilities/com.m:20-28, clamped as :26-28) and second central moments (determine_search_location.m:73) of every footprint =
 * column of the d x K CSC matrix A (pixels column-major in a d1 x d2 image); empty[k] = 1 where the values sum to 0 (:52).  K doubles per output.  The caller
 * takes the 2 x 2 eigendecompositions (:74) and hands the result to cnmfe_search_ellipse.  No device involved. */
int cnmfe_footprint_moments(int32_t K, int32_t d1, int32_t d2, const int64_t *colptr, const int32_t *rowidx, const float *val, double *s_out, uint8_t *empty,
                            double *cmx, double *cmy, double *vxx, double *vxy, double *vyy);
/* host helper: the 'ellipse' search masks IND = determine_search_location(A, 'ellipse', params) (utilities/determine_search_location.m:76-100, call site
 * update_spatial_parallel.m:66) from the per-neuron quantities of :52-82 the caller has computed -- centre of mass (utilities/com.m:20-28), eigenvectors
 * vk = (V11, V21, V12, V22) and eigenvalues clamped to [min_size^2, max_size^2] of the footprint's second moments: pixel (r, c) is in mask k iff
 * sqrt(((r - cmx) V11 + (c - cmy) V21)^2 / d11 + ((r - cmx) V12 + (c - cmy) V22)^2 / d22) <= dist (:84), evaluated in exactly that order in double precision,
 * within R pixels of (floor(cmx), floor(cmy)) and inside the d1 x d2 image; empty[k] != 0: no mask (:102-104).  Outputs: out_colptr[K + 1] (always),
 * out_rowidx (0-based global pixels, ascending per neuron; written only when cap >= the total), *nnz_out = the total: call once with out_rowidx = NULL to size
 * the buffer.  No device involved. */
int cnmfe_search_ellipse(int32_t K, int32_t d1, int32_t d2, const double *cmx, const double *cmy, const double *vk, const double *d11, const double *d22,
                         const uint8_t *empty, double dist, int32_t R, int64_t cap, int64_t *out_colptr, int32_t *out_rowidx, int64_t *nnz_out);

/* host helper: the CSC matrix (nrow x ncol, rows ascending per column) of n (row, column, value) triplets in any order -- the gathered rows of A after the spatial
 * update (update_spatial_parallel.m:324-334: every patch writes its own disjoint rows of A_), assembled in one counting pass + a short sort per column instead of a
 * sort of the whole list.  Outputs: out_colptr[ncol + 1], out_rowidx[n], out_val[n].  CNMFE_EINVAL on an index out of range or a (row, column) given twice. */
int cnmfe_csc_from_triplets(int64_t n, const int32_t *rows, const int32_t *cols, const float *vals, int32_t ncol, int64_t nrow,
                            int64_t *out_colptr, int32_t *out_rowidx, float *out_val);

/* ---- objective: [RSS_total, RSS] = compute_RSS(obj)  (@Sources2D/Sources2D.m:1358-1510), ring model, bg_ssub = 1, one patch, all frames:
 *   RSS = sum((Y(patch,:) - A(patch,:)*C - (W*(Y_block - b0_block - A_prev*C_prev) + b0_new(patch))).^2)
 * The resident residual of this patch must be the one of (A_prev, C_prev) = the block's neurons (cnmfe_residual; :1427-1429, :1475);
 * with it RSS is one read of Ysig.  A: d x K CSC on the PATCH rows, C: K x T; b0_block: reconstruct_b0() on the block
 * (d_b floats, column-major), b0_new: obj.b0_new on the patch (d floats). */
int cnmfe_compute_rss(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                      const float *A_val, const float *C, int c_order, const float *b0_block, const float *b0_new,
                      double *rss_out);

/* Ybg = reconstruct_background(obj, frame_range)  (@Sources2D/Sources2D.m:1247-1355), ring model, bg_ssub = 1, one patch:
 *   Ybg(patch, t) = W*(Y_block - b0_block - A_prev*C_prev)(:, t) + b0_new(patch),  t in [frame0, frame0 + nframes)
 * from the resident residual of (A_prev, C_prev) as for cnmfe_compute_rss.  Ybg_out: d x nframes, frame-major, host or device. */
int cnmfe_reconstruct_background(cnmfe_ctx *ctx, int patch_id, const float *b0_block, const float *b0_new,
                                  int64_t frame0, int64_t nframes, float *Ybg_out, int out_memspace);

/* The same two with bg_ssub > 1 (Sources2D.m:1325-1334 and :1479-1486), where both resizes are 'nearest':
 *   Bf = imresize( W * imresize(Y_block - b0_block - A_prev*C_prev, 1/s, 'nearest'), [nr_block nc_block], 'nearest' )(patch)
 * cnmfe_background_ssub forms W * imresize(...) on the fit patch (the low-resolution patch that holds W; A_prev: d_b x K CSC on the BLOCK rows
 * of patch_id, C_prev: K x T) and keeps it with the context until the next call of it, the next cnmfe_fit_ring_model_ssub or the next upload;
 * the other two then read it:  Ybg(patch, t) = Bf(:, t) + b0_new  and  RSS = sum((Y(patch,:) - A*C - Ybg).^2)  (A: d x K CSC on the patch rows). */
int cnmfe_background_ssub(cnmfe_ctx *ctx, int patch_id, int fit_patch, int32_t ssub, int32_t K, const int64_t *A_colptr,
                          const int32_t *A_rowidx, const float *A_val, const float *C, int c_order, const float *b0_block /* d_b */);
int cnmfe_reconstruct_background_ssub(cnmfe_ctx *ctx, int patch_id, const float *b0_new /* d */, int64_t frame0, int64_t nframes,
                                       float *Ybg_out /* d x nframes, frame-major */, int out_memspace);
int cnmfe_compute_rss_ssub(cnmfe_ctx *ctx, int patch_id, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx,
                           const float *A_val, const float *C, int c_order, const float *b0_new /* d */, double *rss_out);

/* ---- S6: post_process_spatial (connected = true, circular = false)
 * @Sources2D/post_process_spatial.m:19-32 -> endoscope/connectivity_constraint.m:1-18.
 * A is the whole-FOV d1*d2 x K CSC; keep[nnz] receives 1 for entries that survive. */
int cnmfe_post_process_spatial(cnmfe_ctx *ctx, int32_t d1, int32_t d2, int32_t K,
                               const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                               uint8_t *keep);

/* ---- measurement: per-kernel HIP-event timing on the engine's stream ---------
 * on = 1: every launch is bracketed by a pair of events; on = 2: only the kernels a roofline is quoted for (residual sweep, ring solve, window
 * projection / table correction, the two residual projections) -- the pairs cost host and device time per launch, which a run that wants its
 * wall time undisturbed avoids; on = 0: off. */
int cnmfe_profile_enable(cnmfe_ctx *ctx, int on);
int cnmfe_profile_reset(cnmfe_ctx *ctx);
/* number of distinct kernel names seen; name/total_ms/calls of the i-th */
int cnmfe_profile_count(cnmfe_ctx *ctx);
int cnmfe_profile_get(cnmfe_ctx *ctx, int i, char *name, int name_cap, double *total_ms, int64_t *calls);
int cnmfe_synchronize(cnmfe_ctx *ctx);
/* Tunables; unknown names -> CNMFE_EINVAL.  Thirteen behaviour switches, every default = the measured path, every other value parity-tested against it:
 *   r1_variant        R1 sweep kernel (14: duo-role LDS-DMA kernel, the default where it applies; -1: the generic kernel)
 *   r1_delta, r1_lazy, r1_defer    incremental residual (see cnmfe_residual): fold a footprint-term difference into the resident Ysig / only record a request nobody
 *                     reads / sweep without the term and keep it pending.  Default 1 each.
 *   r1_virtual        default 1: a cnmfe_residual without an output buffer records its request and the two updates project the centred video instead of a swept
 *                     Ysig; 0 restores the sweep
 *   ssub_virtual      the same for cnmfe_residual_ssub (through the resampling maps): 2 always, 1 (default) on patches of at least 5e8 samples -- below that the
 *                     low-resolution sweep is the faster form, profiles/r05/ssub_virtual_check.txt --, 0: the low-resolution sweep + upsample
 *   gram_incremental  default 1: the covariance table of the VIDEO is kept and corrected per fit (see cnmfe_fit_ring_model); 0: the direct Gram of Bf every fit
 *   proj_i8_planes    default 3: the temporal projection on the int8 pipe reads the upper three of the video's four digit planes (24-bit samples: 3/4 of the bytes;
 *                     A, C move by < 1e-7); 4: all of them
 *   solve_staged      default 1: the ring solve samples the footprints' U~ and A out of per-neuron windows (ring_solve_staged.hpp; bit-identical weights); 0: it walks
 *                     the CSR rows and slot tables
 *   solve_inv         default 0; 1: fits of a patch from its second one with footprints on solve their pixels out of explicit inverses of the VIDEO's normal equations
 *                     (built once in front of that fit, the bytes of solve_packed's systems once more; the footprints enter by the Woodbury identity, the ridge's
 *                     drift by a short series: ring_solve_inv.hpp); 2: built in front of the first such fit; 0: every fit factors every pixel's system
 *   solve_inv_terms   default 5: terms of the ridge series before a pixel is left to the factorising kernel and its inverse rebuilt
 *   solve_packed      default 1: the ring solve reads per-pixel packed copies of the video's normal equations (43 KB per patch pixel at 96 ring offsets, allocated
 *                     when that much + 8 GB is free) and applies the footprints' corrections in registers; 0: the block-pair table is swept and gathered from
 *   gram_i8           default 1: the table (and the direct Gram of the fallback) on the int8 matrix pipe from 32-bit fixed-point digit planes, exact int32
 *                     accumulation, up to 24576 used frames; 0: the fp64 matrix pipe
 *   win_i8            default 1: the digit planes stay resident (one more video's worth of memory, taken only when that leaves 8 GB free) and every fit's window
 *                     projection runs on the int8 pipe; 0: the fp64 kernel on the centred video
 *   win_i8_planes     default 0 = 3 when the sums run over at least 2048 frames, else 4 (3 / 4 force it): the fit's window projection (and the all-frames table of a strided recording) reads the upper three digit planes of the video -- 24-bit
 *                     samples, 3/4 of the bytes (1.93 -> 1.48 ms at 512 x 512 x 10000); W within 2e-6 of the float64 oracle's (observed 5e-7 .. 1.8e-6 of the largest
 *                     weight; with all four planes, win_i8_planes = 4: 4e-8 .. 2e-7), A and C unchanged at ~1e-7
 *   proj_tiled / proj_i8   default 1 / 1: the temporal projection on the int8 pipe out of a pixel-major copy of the digit planes (needs win_i8), else out of a copy of
 *                     the centred video in its own read order (one video's worth either, same rule); 0 / 0: the frame-major video on the fp64 pipe.  The tables over ALL frames
 *                     (spatial update, temporal projection) sum inside frame segments of at most 24576 frames, recordings up to 16 x 24576 frames
 *   lanes             default 1; 2 .. 4 (set BEFORE the first patch is created): the patches alternate between that many execution lanes -- a HIP stream + a set of the
 *                     context's scratch each -- so that the small kernels of independent patches overlap (update_*_parallel.m: parfor); calls on what the patches share
 *                     (bound traces, stitch, the temporal jobs' sweep, post-processing) join the lanes.  Results are bit-equal to lanes = 1 (tests/test_gpu_lanes.py)
 *   sweep_dag         default 1: the maxIter Gauss-Seidel sweeps of a HALS update (HALS_spatial.m:36-44, HALS_temporal.m:59-68, with or without the in-sweep
 *                     deconvolution) are launched level by level of ONE dependency graph over (sweep, neuron) items -- the same reads and writes per item, equal results,
 *                     fewer and fuller launches (512 x 512, K = 500: 21 instead of 25); 0: sweep after sweep, level by level (rounds 1-5)
 *   prealloc          default 1: cnmfe_fit_reserve may allocate the fit's large buffers ahead of the first fit
 * A deployment short of HBM sets win_i8 = proj_tiled = 0 (and solve_packed = 0) or leaves it to the engine, which falls back by itself.
 * Diagnostics (scripts/): solve_probe, r1_probe (phase probes: results are NOT the product's), deconv_trace, host_trace (1: host-side phase times of every call on
 * stderr, 2: + slow launch calls), debug (1: NaN-poison never-computed table entries).
 * Round 6 removed the retired names rounds 2-5 still accepted and ignored (gram_mode, gram_flush, solve_defer, solve_mode, solve_gfill, gram_kernel, r1_arc_d,
 * r1_arc_bias, r1_duo_ord) and the experiment switches tile_order, gram_probe, r1_nseg.
 * Every option can be preset for a process with CNMFE_OPTS="name=value,..." (logged once on stderr). */
int cnmfe_set_option(cnmfe_ctx *ctx, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* CNMFE_H */
