function update_background_parallel(obj, use_parallel)
% UPDATE_BACKGROUND_PARALLEL  ring-model background update of Sources2D on the MI355X engine.
%
% Drop-in for ca_source_extraction/@Sources2D/update_background_parallel.m (same name, same arguments, same effect on the object:
% obj.W, obj.b0, obj.b0_new, obj.A_prev, obj.C_prev).  Copy this file over the reference's, put cnmfe_mex and cnmfe_handle.m on the path.
% Written from the Python mirror cnmf_e_amd/sources2d.py (update_background_parallel): the per-patch neuron selection stays host logic,
% the regression [W, b0] = fit_ring_model(...) of every patch runs on the GPU against the RESIDENT block (nothing is re-read from disk,
% endoscope/get_patch_data.m is only used once, by cnmfe_handle).  use_parallel is accepted and ignored: MATLAB pool workers are separate
% processes and would each need their own copy of the video; the parallelism is the GPU's (and, with several GPUs, the contexts').
    if ~isfield(obj.P, 'mat_data') || isempty(obj.P.mat_data)
        error('No data file selected');
    end
    if ~strcmpi(obj.options.background_model, 'ring')
        error('cnmfe:model', 'the MI355X engine implements the ring background model; got ''%s''', obj.options.background_model);
    end
    thr = obj.options.thresh_outlier;                    % NaN in every demo; a finite value takes fit_ring_model's outlier branch
    if nargin < 2, use_parallel = true; end  %#ok<NASGU>
    eng = cnmfe_handle(obj);
    d1 = eng.dims(1);  d2 = eng.dims(2);
    np = numel(eng.pid);
    s = obj.options.bg_ssub;
    accel = obj.options.bg_acceleration;

    % obj.C goes up once per context; the per-patch calls only name the rows they need
    for g = 1:numel(eng.h), cnmfe_mex('bind_traces', eng.h(g), obj.C); end

    % "first run" is decided on patch 1's W for every patch, by value, like the reference does
    first_run = cnmfe_mex('first_run', eng.h(eng.owner(1)), local_fit_pid(eng, 1, s));
    A = obj.A;
    for m = 1:np
        h = eng.h(eng.owner(m));
        pix = local_pixels(eng.block_pos{m}, d1);
        Ablk = A(pix, :);
        ind = find(sum(Ablk, 1) > 0);
        if isempty(ind) && ~first_run
            continue;                                    % nothing changed in this area: W{m}, b0{m} stay
        end
        rows = int32(ind(:));
        if ~isnan(thr)                                   % sn of the block (reference :131-138; resized for bg_ssub > 1)
            blk = eng.block_pos{m};
            sn_blk = obj.P.sn(blk(1):blk(2), blk(3):blk(4));
            if s == 1
                cnmfe_mex('set_noise', h, eng.pid(m), sn_blk(:));
            else
                sn_low = imresize(sn_blk, 1/s, 'nearest') * s;
                cnmfe_mex('set_noise', h, eng.pid_fit(m), sn_low(:));
            end
        end
        if s == 1
            cnmfe_mex('fit_ring', h, eng.pid(m), Ablk(:, ind), rows, accel, thr);
        else
            cnmfe_mex('fit_ring_ssub', h, eng.pid(m), eng.pid_fit(m), eng.pid_res(m), s, Ablk(:, ind), rows, accel, thr);
        end
    end

    % keep the object's copies in step with the device (other methods of the class read obj.W / obj.b0)
    for m = 1:np
        h = eng.h(eng.owner(m));
        blk = eng.block_pos{m};  pat = eng.patch_pos{m};
        d_b = (blk(2) - blk(1) + 1) * (blk(4) - blk(3) + 1);
        d = (pat(2) - pat(1) + 1) * (pat(4) - pat(3) + 1);
        if s == 1
            [Wt, b0m] = cnmfe_mex('get_ring', h, eng.pid(m), d, d_b);
        else
            nlow = ceil((blk(2) - blk(1) + 1) / s) * ceil((blk(4) - blk(3) + 1) / s);
            [Wt, ~] = cnmfe_mex('get_ring', h, eng.pid_fit(m), nlow, nlow);
            [~, b0m] = cnmfe_mex('get_ring', h, eng.pid(m), d, d_b);
        end
        obj.W{m} = Wt.';
        obj.b0{m} = b0m;
    end
    obj.b0_new = obj.reconstruct_b0();
    obj.A_prev = obj.A;
    obj.C_prev = obj.C;
end

function pix = local_pixels(rect, d1)
% linear (column-major) FOV indices of the rectangle [r0 r1 c0 c1], in the rectangle's own column-major order
    [rr, cc] = ndgrid(rect(1):rect(2), rect(3):rect(4));
    pix = (cc(:) - 1) * d1 + rr(:);
end

function pid = local_fit_pid(eng, m, s)
    if s == 1, pid = eng.pid(m); else, pid = eng.pid_fit(m); end
end
