function eng = cnmfe_handle(obj, gpus)
% CNMFE_HANDLE  the MI355X engine state of a Sources2D object, created on first use.
%
%   eng = cnmfe_handle(obj)          one context on GPU 0
%   eng = cnmfe_handle(obj, [0 1 2]) one context per listed GPU; patches are dealt round-robin in MATLAB's linear patch order
%                                    (SURVEY 8(e)); the temporal update then stitches over RCCL inside cnmfe_mex('stitch_temporal').
%
% On first use every block of mat_data (patch + ring halo, distribute_data.m:165-171) is read ONCE through the reference's own
% get_patch_data (endoscope/get_patch_data.m:50-93) and uploaded in its file class; it stays in HBM for all later updates, and with it
% W{m}, b0{m}.  The state lives in obj.P.cnmfe (P is the reference's free-form parameter struct, Sources2D.m:33), so it travels with the
% object handle and a `clear mex` simply rebuilds it.  eng fields: h (context per GPU), owner / pid (per patch), pos (geometry), dims.
    if isfield(obj.P, 'cnmfe') && ~isempty(obj.P.cnmfe) && cnmfe_alive(obj.P.cnmfe)
        eng = obj.P.cnmfe;
        return;
    end
    if nargin < 2 || isempty(gpus), gpus = 0; end
    md = obj.P.mat_data;
    dims = md.dims;  d1 = dims(1);  d2 = dims(2);
    fr = obj.frame_range;  T = diff(fr) + 1;
    opt = obj.options;
    eng = struct('h', zeros(1, numel(gpus)), 'gpus', gpus, 'dims', [d1 d2 T], 'ssub', opt.bg_ssub);
    for g = 1:numel(gpus), eng.h(g) = cnmfe_mex('create', gpus(g)); end
    np = numel(md.patch_pos);
    % several patches per context: their calls alternate between execution lanes (a HIP stream + a scratch set each, include/cnmfe.h option 'lanes'; set before the
    % first patch) -- the parfor over patches of the three update methods as concurrent streams, the same values bit for bit
    per_ctx = ceil(np / numel(gpus));
    if per_ctx > 1
        for g = 1:numel(gpus), cnmfe_mex('set_option', eng.h(g), 'lanes', min(3, per_ctx)); end
    end
    eng.patch_pos = md.patch_pos;  eng.block_pos = md.block_pos;
    eng.owner = mod((0:np-1), numel(gpus)) + 1;          % context index of patch m
    eng.pid = 0:np-1;                                    % ids of the full-resolution patches; 2 low-resolution companions per patch behind them
    eng.pid_fit = np + 2 * (0:np-1);  eng.pid_res = eng.pid_fit + 1;
    rr = ceil(opt.ring_radius / opt.bg_ssub);            % initComponents_parallel.m:214
    for m = 1:np
        h = eng.h(eng.owner(m));
        cnmfe_mex('patch', h, eng.pid(m), md.patch_pos{m}, md.block_pos{m}, d1, d2, T);
        Y = get_patch_data(md, md.patch_pos{m}, fr, true);                 % nr_b x nc_b x T in the file's class
        cnmfe_mex('upload', h, eng.pid(m), reshape(Y, [], T), 0);
        cnmfe_mex('ring_init', h, eng.pid(m), opt.ring_radius, opt.num_neighbors);
        if opt.bg_ssub > 1
            cnmfe_mex('derive', h, eng.pid(m), eng.pid_fit(m), opt.bg_ssub, 'nearest');
            cnmfe_mex('derive', h, eng.pid(m), eng.pid_res(m), opt.bg_ssub, 'bicubic');
            cnmfe_mex('ring_init', h, eng.pid_fit(m), rr, opt.num_neighbors);
            cnmfe_mex('ring_init', h, eng.pid_res(m), rr, opt.num_neighbors);
            cnmfe_mex('fit_reserve', h, eng.pid_fit(m));                   % the fit's large device buffers now, not inside the first update_background_parallel
        else
            cnmfe_mex('fit_reserve', h, eng.pid(m));
        end
        % a W{m}, b0{m} fitted in an earlier session goes back on the device (same ring pattern; values in MATLAB's column order of W.')
        if ~isempty(obj.W) && numel(obj.W) >= m && ~isempty(obj.W{m}) && nnz(obj.W{m}) > 0 && opt.bg_ssub == 1
            cnmfe_mex('set_ring', h, eng.pid(m), obj.W{m}.', obj.b0{m});
        end
    end
    eng.stamp = now;
    obj.P.cnmfe = eng;
end

function ok = cnmfe_alive(eng)
% a context handle survives as long as the MEX file stays loaded
    ok = true;
    try
        cnmfe_mex('first_run', eng.h(1), eng.pid(1));
    catch
        ok = false;
    end
end
