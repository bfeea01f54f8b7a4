function update_temporal_parallel(obj, use_parallel, use_c_hat)
% UPDATE_TEMPORAL_PARALLEL  temporal update of Sources2D on the MI355X engine.
%
% Drop-in for ca_source_extraction/@Sources2D/update_temporal_parallel.m (same name / arguments; sets obj.C_raw, obj.C and, with
% deconvolution, obj.S, obj.P.kernel_pars, obj.P.neuron_sn; obj.b0_new).  Written from cnmf_e_amd/sources2d.py (update_temporal_parallel).
% Per patch: the residual sweep with the block's neurons of (A_prev, C_prev), then HALS on the traces of the neurons on the block; the
% pieces never leave the device -- every patch adds aa .* C_raw to its context's stitch accumulator, 'stitch_temporal' sums the accumulators
% of all contexts (RCCL all-reduce over xGMI when there are several GPUs), divides by the summed weights and subtracts the row minima.
    if ~isfield(obj.P, 'mat_data') || isempty(obj.P.mat_data)
        error('No data file selected');
    end
    if nargin < 2, use_parallel = true; end  %#ok<NASGU>
    if nargin < 3 || isempty(use_c_hat), use_c_hat = true; end
    eng = cnmfe_handle(obj);
    d1 = eng.dims(1);  T = eng.dims(3);
    np = numel(eng.pid);
    opt = obj.options;
    s = opt.bg_ssub;
    K = size(obj.C, 1);
    deconv = opt.deconv_flag;
    if deconv
        dopt = opt.deconv_options;
        if ~strcmpi(dopt.type, 'ar1') || ~strcmpi(dopt.method, 'foopsi')
            error('cnmfe:deconv', 'the engine deconvolves with type ''ar1'', method ''foopsi''');
        end
        if ~isfield(obj.P, 'kernel_pars') || numel(obj.P.kernel_pars) ~= K, pars_all = zeros(K, 1); else, pars_all = obj.P.kernel_pars(:); end
    end

    same_C = isequal(size(obj.C), size(obj.C_prev)) && isequal(obj.C, obj.C_prev);
    for g = 1:numel(eng.h)
        cnmfe_mex('bind_traces', eng.h(g), obj.C_prev);
        cnmfe_mex('stitch_begin', eng.h(g), K, T);
    end
    jobs = {};
    for m = 1:np
        h = eng.h(eng.owner(m));
        pix_p = local_pixels(eng.patch_pos{m}, d1);
        pix_b = local_pixels(eng.block_pos{m}, d1);
        ind = find(sum(obj.A(pix_b, :), 1) > 0);
        if isempty(ind), continue; end
        indp = find(sum(obj.A_prev(pix_b, :), 1) > 0);
        if s == 1
            cnmfe_mex('residual', h, eng.pid(m), obj.A_prev(pix_b, indp), int32(indp(:)));
        else
            cnmfe_mex('residual_ssub', h, eng.pid(m), eng.pid_res(m), s, obj.A_prev(pix_b, indp), int32(indp(:)));
        end
        A_pp = obj.A(pix_p, ind);
        if ~use_c_hat
            cnmfe_mex('fast_temporal', h, eng.pid(m), A_pp, T);
            cnmfe_mex('stitch_add', h, ind);
        elseif deconv
            [~, ~, ~, pars] = cnmfe_mex('temporal_deconv', h, eng.pid(m), A_pp, obj.C(ind, :), opt.maxIter, dopt.smin, dopt.max_tau, pars_all(ind));
            pars_all(ind) = pars;
            cnmfe_mex('stitch_add', h, ind);
        else
            % the patches are independent (the reference's loop is a parfor, :112-186): every patch's projections are queued as a JOB and the
            % Gauss-Seidel levels of all jobs of a context run together below -- a level of one patch is a handful of workgroups
            if same_C, Carg = int32(ind(:)); else, Carg = obj.C(ind, :); end
            jobs{end + 1} = {h, cnmfe_mex('temporal_job', h, eng.pid(m), A_pp, Carg, opt.maxIter), ind}; %#ok<AGROW>
        end
    end
    if ~isempty(jobs)
        for g = 1:numel(eng.h), cnmfe_mex('temporal_jobs_sweep', eng.h(g)); end
        for j = 1:numel(jobs), cnmfe_mex('stitch_add_job', jobs{j}{1}, jobs{j}{2}, jobs{j}{3}); end
    end
    C_raw = double(cnmfe_mex('stitch_temporal', eng.h, ~deconv, K, T));
    if deconv
        [C, C_raw, S, kp, sn] = cnmfe_mex('deconv_temporal', eng.h(1), C_raw, dopt.smin, dopt.max_tau);
        obj.C = C;  obj.C_raw = C_raw;  obj.S = S;
        obj.P.kernel_pars = kp;  obj.P.neuron_sn = sn;
    else
        obj.C_raw = C_raw;
        obj.C = C_raw;
    end
    Ymean = cell2mat(obj.P.Ymean);
    obj.b0_new = Ymean - obj.reshape(obj.A * mean(obj.C, 2), 2);
end

function pix = local_pixels(rect, d1)
    [rr, cc] = ndgrid(rect(1):rect(2), rect(3):rect(4));
    pix = (cc(:) - 1) * d1 + rr(:);
end
