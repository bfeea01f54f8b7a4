function make_fixtures(ref_root, repo_root)
% MAKE_FIXTURES  run the REFERENCE (zhoupc/CNMF_E) on the seeded inputs of tests/golden/matlab_inputs.mat and store what it returns.
%
%   make_fixtures('/path/to/CNMF_E', '/path/to/this/repo')      (MATLAB with the Image Processing, Signal Processing and Statistics toolboxes,
%                                                                 or GNU Octave >= 6 with the packages signal, image, statistics)
%   from a shell:  octave --eval "addpath('oracle/matlab'); make_fixtures('/path/to/CNMF_E', pwd)"
%
% Every group of calls runs in its own try/catch: a group whose functions the interpreter lacks is skipped and named in out.skipped, the others are
% still written.  out.engine records which interpreter produced the numbers: under Octave pwelch / fminbnd / imresize / medfilt2 are DIFFERENT
% implementations (not MathWorks'), so tests/test_matlab_fixtures.py holds those fields to the documented-semantics restatement only when
% out.engine says MATLAB; everything the reference itself computes (get_nhood, fit_ring_model, HALS_*, nnls_spatial, com, ...) is pinned either way.
%
% Output: <repo_root>/tests/golden/matlab_outputs.mat (-v7).  tests/test_matlab_fixtures.py compares oracle/cnmfe_oracle.py and
% oracle/oasis_oracle.py with it -- this is what turns "parity unpinned" (DESIGN.md) into a pinned oracle.  Nothing of the reference is copied:
% the script only CALLS its functions.  TEST INFRASTRUCTURE.
    is_octave = exist('OCTAVE_VERSION', 'builtin') ~= 0;
    if is_octave
        pkg load signal; pkg load image; pkg load statistics;
    end
    addpath(genpath(fullfile(ref_root, 'ca_source_extraction')));
    addpath(genpath(fullfile(ref_root, 'OASIS_matlab')));
    in = load(fullfile(repo_root, 'tests', 'golden', 'matlab_inputs.mat'));
    d1 = in.d1; d2 = in.d2; T = in.T;
    out = struct();
    if is_octave, out.engine = ['octave ' OCTAVE_VERSION]; else, out.engine = ['matlab ' version]; end
    out.skipped = '';

    % ---- ring geometry: get_nhood.m
    try
        for i = 1:numel(in.nhood_radii)
            [rs, cs] = get_nhood(in.nhood_radii(i));
            out.(sprintf('nhood_r%d', in.nhood_radii(i))) = [rs(:), cs(:)];
        end
        [rs, cs] = get_nhood(15, 40);  out.nhood_r15_k40 = [rs(:), cs(:)];
    catch err
        out.skipped = [out.skipped ' ring_geometry(' err.message ');']; fprintf('skipped ring_geometry: %s\n', err.message);
    end

    ind_patch = logical(in.ind_patch);
    mask = false(d1, d2);  mask(in.block(1):in.block(2), in.block(3):in.block(4)) = true;
    Yb = in.Y(mask(:), :);  Ab = in.A(mask(:), :);  snb = in.sn(mask(:));

    % ---- background: fit_ring_model.m (first run, second run, without projection, outlier branch)
    try
        [W1, b01] = fit_ring_model(Yb, Ab, in.C, in.W0, NaN, snb(ind_patch), ind_patch, true);
        [W2, b02] = fit_ring_model(Yb, Ab, in.C, W1, NaN, snb(ind_patch), ind_patch, true);
        [W3, ~] = fit_ring_model(Yb, Ab, in.C, in.W0, NaN, snb(ind_patch), ind_patch, false);
        [W4, b04] = fit_ring_model(Yb, Ab, in.C, in.W0, in.thresh_outlier, snb(ind_patch), ind_patch, true);
        out.fit_W1 = full(W1); out.fit_b01 = b01; out.fit_W2 = full(W2); out.fit_b02 = b02; out.fit_W3 = full(W3); out.fit_W4 = full(W4); out.fit_b04 = b04;
    catch err
        out.skipped = [out.skipped ' fit_ring_model(' err.message ');']; fprintf('skipped fit_ring_model: %s\n', err.message);
    end

    % ---- the residual expression of update_spatial_parallel.m:162-166 with W1, b01
    try
        tmpY = double(Yb) - Ab * in.C;
        Ysig = double(Yb(ind_patch, :)) - W1 * tmpY - (b01 - W1 * mean(tmpY, 2)) * ones(1, T);
        out.Ysig = Ysig;
    catch err
        out.skipped = [out.skipped ' residual(' err.message ');']; fprintf('skipped residual: %s\n', err.message);
    end

    % ---- spatial: HALS_spatial.m, HALS_spatial_thresh.m, nnls_spatial.m on the patch rows
    try
        IND = sparse(logical(determine_search_location(in.A, 'ellipse', struct('d1', d1, 'd2', d2, 'min_size', 3, 'max_size', 8, 'dist', 3))));
        out.IND_ellipse = full(IND);
        Ap = full(Ab(ind_patch, :));  INDp = IND(mask(:), :);  INDp = INDp(ind_patch, :);  snp = snb(ind_patch);
        out.A_hals = HALS_spatial(Ysig, Ap, in.C, INDp, 3);
        out.A_thresh = HALS_spatial_thresh(Ysig, Ap, in.C, INDp, 3, snp);
        out.A_nnls = nnls_spatial(Ysig, Ap, in.C, INDp, in.maxN);
        out.cm = com(in.A, d1, d2);
    catch err
        out.skipped = [out.skipped ' spatial(' err.message ');']; fprintf('skipped spatial: %s\n', err.message);
    end

    % ---- temporal: HALS_temporal.m without deconvolution
    try
        [Ct, Crawt] = HALS_temporal(Ysig, out.A_hals, in.C, 5, []);
        out.C_hals = Ct;  out.Craw_hals = Crawt;
    catch err
        out.skipped = [out.skipped ' temporal(' err.message ');']; fprintf('skipped temporal: %s\n', err.message);
    end

    % ---- post-processing and the optional branches
    try
        K = size(in.imgs, 3);
        out.conn = zeros(size(in.imgs));  out.circ = zeros(size(in.imgs));
        for k = 1:K
            out.conn(:, :, k) = connectivity_constraint(in.imgs(:, :, k));
            out.circ(:, :, k) = circular_constraints(in.imgs(:, :, k));
        end
        Aimg = sparse(reshape(in.imgs, d1 * d2, K));
        opt = struct('d1', d1, 'd2', d2, 'd3', 1, 'nb', 1, 'nrgthr', 0.99, 'clos_op', strel('square', 3), 'medw', [3, 3]);
        out.thr_comp = full(threshold_components(Aimg, opt));
        opt.se = strel('disk', 4, 0);  opt.min_size = 3;  opt.max_size = 8;  opt.dist = 3;  opt.bSiz = 3;
        out.IND_dilate = full(determine_search_location(Aimg, 'dilate', opt));
    catch err
        out.skipped = [out.skipped ' post_processing(' err.message ');']; fprintf('skipped post_processing: %s\n', err.message);
    end

    % ---- toolbox functions the restatement depends on (PARITY UNPINNED until this file exists)
    try
        out.resize_half = imresize(in.resize_img, 1/2);
        out.resize_third = imresize(in.resize_img, 1/3);
        out.resize_half_nearest = imresize(in.resize_img, 1/2, 'nearest');
        out.resize_up = imresize(out.resize_half, [size(in.resize_img, 1), size(in.resize_img, 2)]);
        out.resize_up_nearest = imresize(out.resize_half_nearest, [size(in.resize_img, 1), size(in.resize_img, 2)], 'nearest');
        out.quant = quantile(in.quant_x, in.quant_p);
        out.medfilt = medfilt2(in.imgs(:, :, 1));
    catch err
        out.skipped = [out.skipped ' toolbox(' err.message ');']; fprintf('skipped toolbox: %s\n', err.message);
    end

    % ---- OASIS: GetSn.m, estimate_time_constant.m, deconvolveCa.m (ar1 / foopsi, the demo's options)
    try
        n = size(in.traces, 1);
        out.sn_tr = zeros(n, 1);  out.g_tr = zeros(n, 1);  out.b_tr = zeros(n, 1);
        out.c_tr = zeros(size(in.traces));  out.s_tr = zeros(size(in.traces));  out.g0_tr = zeros(n, 1);
        for i = 1:n
            y = in.traces(i, :);
            out.sn_tr(i) = GetSn(y);
            out.g0_tr(i) = estimate_time_constant(y, 1, out.sn_tr(i));
            [c, s, o] = deconvolveCa(y, 'ar1', 'foopsi', 'smin', -5, 'optimize_pars', true, 'optimize_b', true, 'max_tau', 100, 'sn', out.sn_tr(i));
            out.c_tr(i, :) = c(:)';  out.s_tr(i, :) = s(:)';  out.g_tr(i) = o.pars(1);  out.b_tr(i) = o.b;
        end
    catch err
        out.skipped = [out.skipped ' oasis(' err.message ');']; fprintf('skipped oasis: %s\n', err.message);
    end

    save(fullfile(repo_root, 'tests', 'golden', 'matlab_outputs.mat'), '-struct', 'out', '-v7');
    fprintf('written %s\n', fullfile(repo_root, 'tests', 'golden', 'matlab_outputs.mat'));
end
