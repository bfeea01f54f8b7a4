function sn = cnmfe_estimate_noise(obj, frame_range)
% CNMFE_ESTIMATE_NOISE  body for Sources2D.estimate_noise(obj, frame_range, 'psd') (ca_source_extraction/@Sources2D/Sources2D.m:328-379) on the
% MI355X engine: GetSn (Welch periodogram, upper half of the spectrum, mean of the log) per pixel on the resident blocks, then the storage-block
% bookkeeping of :361-376 -- the reference evaluates storage block [r0 r1] x [c0 c1] INCLUDING the line it shares with the next block and then
% deletes row / column END-1 of every block but the last, so line b-1 of the image holds the estimate of line b for every interior cut line b;
% kept as it is.  frame_range = [1 n]: the first n frames (default [1 min(T, 3000)], :331-333).  'hist' / 'std' stay with the reference.
    eng = cnmfe_handle(obj);
    md = obj.P.mat_data;
    d1 = eng.dims(1);  d2 = eng.dims(2);  T = eng.dims(3);
    if nargin < 2 || isempty(frame_range), frame_range = [1 min(T, 3000)]; end
    if frame_range(1) ~= 1, error('cnmfe:frame_range', 'the engine reads the frames from the first one on'); end
    n = diff(frame_range) + 1;
    sn = zeros(d1 * d2, 1);
    for m = 1:numel(eng.pid)
        h = eng.h(eng.owner(m));
        pix_p = rect_pixels(eng.patch_pos{m}, d1);
        pix_b = rect_pixels(eng.block_pos{m}, d1);
        sn_b = cnmfe_mex('estimate_noise', h, eng.pid(m), numel(pix_b), n);
        sn(pix_p) = sn_b(ismember(pix_b, pix_p));
    end
    sn = reshape(sn, d1, d2);
    br = md.block_idx_r;  bc = md.block_idx_c;                   % distribute_data.m:81-110
    for b = reshape(br(2:end-1), 1, []), sn(b - 1, :) = sn(b, :); end
    for b = reshape(bc(2:end-1), 1, []), sn(:, b - 1) = sn(:, b); end
end

function pix = rect_pixels(rect, d1)
    [rr, cc] = ndgrid(rect(1):rect(2), rect(3):rect(4));
    pix = (cc(:) - 1) * d1 + rr(:);
end
