function Ybg = cnmfe_reconstruct_background(obj, frame_range)
% CNMFE_RECONSTRUCT_BACKGROUND  body for Sources2D.reconstruct_background (ca_source_extraction/@Sources2D/Sources2D.m:1247-1355) on the MI355X
% engine, ring model: Ybg(patch) = W*(Y - b0 - A_prev*C_prev) + b0_new for the frames frame_range = [first last] (1-based, inclusive; default
% obj.frame_range), single precision d1 x d2 x T'.  The background of the whole video is formed once per patch on the device; only the frames asked
% for cross PCIe, at most 4096 per call.  Written from cnmf_e_amd/sources2d.py (reconstruct_background).
    eng = cnmfe_handle(obj);
    d1 = eng.dims(1);  d2 = eng.dims(2);  T = eng.dims(3);
    if nargin < 2 || isempty(frame_range), frame_range = obj.frame_range; end          % :1278-1280
    if isempty(frame_range), frame_range = [1 T]; end
    if ~isempty(obj.frame_range), frame_range = frame_range + 1 - obj.frame_range(1); end   % frame_shift, :1281-1285: the resident video starts at obj.frame_range(1)
    f0 = frame_range(1);  f1 = frame_range(2);
    np = numel(eng.pid);
    s = obj.options.bg_ssub;
    b0_ = obj.reconstruct_b0();                                  % :1292
    b0_new_ = obj.reshape(obj.b0_new, 2);                        % :1293
    for g = 1:numel(eng.h), cnmfe_mex('bind_traces', eng.h(g), obj.C_prev); end
    Ybg = zeros(d1 * d2, f1 - f0 + 1, 'single');                 % :1297
    for m = 1:np
        h = eng.h(eng.owner(m));
        pix_p = rect_pixels(eng.patch_pos{m}, d1);
        pix_b = rect_pixels(eng.block_pos{m}, d1);
        indp = find(sum(obj.A_prev(pix_b, :), 1) > 0);           % :1317-1320
        Aprev_b = obj.A_prev(pix_b, indp);
        if s == 1
            cnmfe_mex('residual', h, eng.pid(m), Aprev_b, int32(indp(:)));
        else                                                     % :1325-1334
            cnmfe_mex('background_ssub', h, eng.pid(m), eng.pid_fit(m), s, Aprev_b, int32(indp(:)), b0_(pix_b));
        end
        for t0 = (f0 - 1):4096:(f1 - 1)                          % 0-based first frame of the slab
            n = min(4096, f1 - t0);
            if s == 1
                slab = cnmfe_mex('reconstruct_background', h, eng.pid(m), b0_(pix_b), b0_new_(pix_p), t0, n);
            else
                slab = cnmfe_mex('reconstruct_background_ssub', h, eng.pid(m), b0_new_(pix_p), t0, n);
            end
            Ybg(pix_p, (t0 - f0 + 2):(t0 - f0 + 1 + n)) = slab;
        end
    end
    Ybg = reshape(Ybg, d1, d2, []);
end

function pix = rect_pixels(rect, d1)
    [rr, cc] = ndgrid(rect(1):rect(2), rect(3):rect(4));
    pix = (cc(:) - 1) * d1 + rr(:);
end
