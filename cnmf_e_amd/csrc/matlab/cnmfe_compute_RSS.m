function [RSS_total, RSS] = cnmfe_compute_RSS(obj)
% CNMFE_COMPUTE_RSS  body for Sources2D.compute_RSS (ca_source_extraction/@Sources2D/Sources2D.m:1358-1510) on the MI355X engine, ring model, all
% frames of obj.frame_range.  Per patch the device forms  sum((Y(patch) - A*C - (W*(Y - b0 - A_prev*C_prev) + b0_new)).^2)  from the resident block:
% 'residual' (or 'background_ssub' when bg_ssub > 1: 'nearest' both ways, :1479-1486) leaves the background of (A_prev, C_prev) in HBM,
% 'compute_rss' reads it once against A*C.  Written from cnmf_e_amd/sources2d.py (compute_RSS).
    eng = cnmfe_handle(obj);
    d1 = eng.dims(1);  d2 = eng.dims(2);
    np = numel(eng.pid);
    s = obj.options.bg_ssub;
    b0_ = obj.reconstruct_b0();                                  % :1398
    b0_new_ = obj.reshape(obj.b0_new, 2);                        % :1399
    for g = 1:numel(eng.h), cnmfe_mex('bind_traces', eng.h(g), obj.C_prev); end
    RSS = cell(size(eng.patch_pos));
    for m = 1:np
        h = eng.h(eng.owner(m));
        pix_p = rect_pixels(eng.patch_pos{m}, d1);
        pix_b = rect_pixels(eng.block_pos{m}, d1);
        ind = find(sum(obj.A(pix_b, :), 1) > 0);                 % :1423
        indp = find(sum(obj.A_prev(pix_b, :), 1) > 0);           % :1427
        Aprev_b = obj.A_prev(pix_b, indp);
        A_pp = obj.A(pix_p, ind);                                % A_patch(ind_patch, :)  (:1467)
        if s == 1
            cnmfe_mex('residual', h, eng.pid(m), Aprev_b, int32(indp(:)));
            RSS{m} = cnmfe_mex('compute_rss', h, eng.pid(m), A_pp, obj.C(ind, :), b0_(pix_b), b0_new_(pix_p));
        else
            cnmfe_mex('background_ssub', h, eng.pid(m), eng.pid_fit(m), s, Aprev_b, int32(indp(:)), b0_(pix_b));
            RSS{m} = cnmfe_mex('compute_rss_ssub', h, eng.pid(m), A_pp, obj.C(ind, :), b0_new_(pix_p));
        end
    end
    RSS_total = sum(cell2mat(RSS(:)));                           % :1507-1508
    obj.P.RSS = RSS_total;                                       % :1509
end

function pix = rect_pixels(rect, d1)
    [rr, cc] = ndgrid(rect(1):rect(2), rect(3):rect(4));
    pix = (cc(:) - 1) * d1 + rr(:);
end
