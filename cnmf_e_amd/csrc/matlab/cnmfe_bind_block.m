function [h, pid, d, d_b] = cnmfe_bind_block(Y, ind_patch, W_old)
% CNMFE_BIND_BLOCK  one-shot binding for the function-level shadows (INTEGRATION.md 2(a)): a block handed over as a matrix.
%
%   [h, pid, d, d_b] = cnmfe_bind_block(Y, ind_patch, W_old)
%
% Y is the d_b x T block a reference function received (fit_ring_model.m:1, HALS_spatial.m:1, ...), ind_patch the logical nr_b x nc_b mask
% of the patch inside it (all true when empty), W_old the current ring matrix (d x d_b) or [] for a block without a ring.  The block is
% uploaded to a scratch patch (id 0) of a persistent context; the ring geometry is recovered from W_old's pattern: its radius is the largest
% row / column distance between a pixel and a neighbour, and the values of W_old are installed so that the engine's first-run test
% (fit_ring_model.m:25) sees what the reference would see.  This path uploads the block on EVERY call -- it exists so that the engine can be
% tried function by function; the method-level files in @Sources2D/ keep the video resident.
    persistent H
    if isempty(H) || ~cnmfe_ok(H), H = cnmfe_mex('create', 0); end
    h = H;  pid = 0;
    [d_b, T] = size(Y);
    if isempty(ind_patch), ind_patch = true(d_b, 1); end
    [nr_b, nc_b] = size(ind_patch);
    if nc_b == 1 && nr_b == d_b                      % a column mask: the block is treated as one image column
        nr_b = d_b;  nc_b = 1;
    end
    [rr, cc] = find(reshape(ind_patch, nr_b, nc_b));
    prect = [min(rr) max(rr) min(cc) max(cc)];
    d = (prect(2) - prect(1) + 1) * (prect(4) - prect(3) + 1);
    cnmfe_mex('patch', h, pid, prect, [1 nr_b 1 nc_b], nr_b, nc_b, T);
    cnmfe_mex('upload', h, pid, Y, 0);
    if nargin >= 3 && ~isempty(W_old)
        [i, j] = find(W_old(1, :));                  %#ok<ASGLU> neighbours of the first patch pixel
        [r1, c1] = ind2sub([nr_b nc_b], j);
        radius = max(max(abs(r1 - prect(1))), max(abs(c1 - prect(3))));
        cnmfe_mex('ring_init', h, pid, radius, []);
        cnmfe_mex('set_ring', h, pid, W_old.', zeros(d, 1));
    else
        cnmfe_mex('ring_init', h, pid, 1, []);       % a minimal ring: the caller only wants the factor updates on this block
    end
end

function ok = cnmfe_ok(h)
    ok = true;
    try
        cnmfe_mex('bind_traces', h, []);
    catch
        ok = false;
    end
end
