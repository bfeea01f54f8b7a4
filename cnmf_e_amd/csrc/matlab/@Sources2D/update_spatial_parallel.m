function update_spatial_parallel(obj, use_parallel, update_sn)
% UPDATE_SPATIAL_PARALLEL  spatial update of Sources2D on the MI355X engine.
%
% Drop-in for ca_source_extraction/@Sources2D/update_spatial_parallel.m (same name / arguments; sets obj.A, obj.P.sn when update_sn,
% obj.b0_new).  Written from cnmf_e_amd/sources2d.py (update_spatial_parallel).  Per patch: the background-subtracted video
% Ysig = Y(patch) - W*(Y - A_prev*C_prev) - (b0 - W*mean(.)) is formed on the device from the resident block ('residual'), the
% footprints are updated on it by HALS / thresholded HALS / NNLS ('spatial'); rows of different patches are disjoint and are written,
% not accumulated.  The search mask is the reference's own determine_search_location (host).
    if ~isfield(obj.P, 'mat_data') || isempty(obj.P.mat_data)
        error('No data file selected');
    end
    if nargin < 2, use_parallel = true; end  %#ok<NASGU>
    if nargin < 3 || isempty(update_sn), update_sn = false; end
    eng = cnmfe_handle(obj);
    d1 = eng.dims(1);  d2 = eng.dims(2);
    np = numel(eng.pid);
    opt = obj.options;
    s = opt.bg_ssub;
    alg = lower(opt.spatial_algorithm);
    if strcmp(alg, 'nnls'), param = 20; else, param = 3; end

    IND = sparse(logical(determine_search_location(obj.A, opt.search_method, opt)));
    for g = 1:numel(eng.h), cnmfe_mex('bind_traces', eng.h(g), obj.C_prev); end
    same_C = isequal(size(obj.C), size(obj.C_prev)) && isequal(obj.C, obj.C_prev);   % after a background update they are the same matrix

    K = size(obj.A, 2);
    ii = cell(np, 1);  jj = cell(np, 1);  vv = cell(np, 1);
    sn_all = obj.P.sn(:);
    pending = {};
    for m = 1:np
        h = eng.h(eng.owner(m));
        pix_p = local_pixels(eng.patch_pos{m}, d1);
        pix_b = local_pixels(eng.block_pos{m}, d1);
        ind = find(any(IND(pix_p, :), 1));
        if isempty(ind) && ~update_sn, continue; end
        % neurons of the previous (A, C) that reach the HALO of the block only: what the sweep subtracts before W is applied
        in_patch = ismember(pix_b, pix_p);
        halo = pix_b(~in_patch);
        indp = find(sum(obj.A_prev(halo, :), 1) > 0);
        Aprev_b = obj.A_prev(pix_b, indp);
        if s == 1
            cnmfe_mex('residual', h, eng.pid(m), Aprev_b, int32(indp(:)));
        else
            cnmfe_mex('residual_ssub', h, eng.pid(m), eng.pid_res(m), s, Aprev_b, int32(indp(:)));
        end
        sn_p = sn_all(pix_p);
        if update_sn
            sn_p = cnmfe_mex('get_sn', h, eng.pid(m), numel(pix_p));
            sn_all(pix_p) = sn_p;
        end
        if isempty(ind), continue; end
        if same_C, Carg = int32(ind(:)); else, Carg = obj.C(ind, :); end
        % the update is QUEUED (sweeps + the copy of the result into pinned memory) and collected two patches late: the host cuts the next patch's
        % slices and assembles the previous patch's triplets while the device works (a blocking 'spatial' per patch drained the stream np times)
        pending{end + 1} = {h, cnmfe_mex('spatial_queue', h, eng.pid(m), alg, obj.A(pix_p, ind), Carg, IND(pix_p, ind), sn_p, param), m, pix_p, ind}; %#ok<AGROW>
        while numel(pending) > 2
            [ii, jj, vv] = collect_one(pending{1}, ii, jj, vv);  pending(1) = [];
        end
    end
    while ~isempty(pending)
        [ii, jj, vv] = collect_one(pending{1}, ii, jj, vv);  pending(1) = [];
    end
    for g = 1:numel(eng.h), cnmfe_mex('synchronize', eng.h(g)); end      % what the kernels had to report is heard before obj.A is replaced
    A_new = sparse(cell2mat(ii), cell2mat(jj), cell2mat(vv), d1 * d2, K);
    if update_sn, obj.P.sn = reshape(sn_all, size(obj.P.sn)); end

    % post-processing: connected component of the peak (spatial_constraints.connected), on the device
    if ~isfield(opt, 'spatial_constraints') || ~isfield(opt.spatial_constraints, 'connected') || opt.spatial_constraints.connected
        keep = cnmfe_mex('postprocess', eng.h(1), A_new, d1, d2);
        [r, c, v] = find(A_new);
        A_new = sparse(r(keep), c(keep), v(keep), d1 * d2, K);
    end
    % spatial_constraints.circular (off in every demo): one small image per neuron, the reference's own circular_constraints on the host,
    % after the connectivity step as in post_process_spatial.m:22-31
    if isfield(opt, 'spatial_constraints') && isfield(opt.spatial_constraints, 'circular') && opt.spatial_constraints.circular
        for k = 1:K
            ai = circular_constraints(reshape(full(A_new(:, k)), d1, d2));
            A_new(:, k) = sparse(ai(:));
        end
    end
    obj.A = A_new;
    Ymean = cell2mat(obj.P.Ymean);
    obj.b0_new = Ymean - obj.reshape(obj.A * mean(obj.C, 2), 2);
end

function [ii, jj, vv] = collect_one(p, ii, jj, vv)
    Anew = cnmfe_mex('spatial_collect', p{1}, p{2});
    [r, c, v] = find(Anew);
    m = p{3};  pix_p = p{4};  ind = p{5};
    ii{m} = pix_p(r);  jj{m} = reshape(ind(c), [], 1);  vv{m} = v;
end

function pix = local_pixels(rect, d1)
    [rr, cc] = ndgrid(rect(1):rect(2), rect(3):rect(4));
    pix = (cc(:) - 1) * d1 + rr(:);
end
