"""Static instruction statistics + DPP-hazard check of kernels in a device assembly file:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -S cnmf_e_amd/csrc/bg.hip -o /tmp/bg_dev.s
    python scripts/isa_stats.py /tmp/bg_dev.s k_ring_solve6
The hazard: a VALU write of a VGPR followed within two wait states by a DPP read of it (inline-asm DPP operations are invisible to the compiler's hazard
recogniser; round 5 found the scheduler sinking a select to right in front of such a read)."""
import re, sys
def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
def hazards(ins):
    out = []
    for k, l in enumerate(ins):
        if '_dpp' in l and l.startswith('v_'):
            src = regs(l.split(None, 1)[1].split(',')[1].split()[0])
            ws, j = 0, k - 1
            while j >= 0 and ws < 2:
                p = ins[j]
                if p.startswith('s_nop'): ws += int(p.split()[1]) + 1
                else:
                    if p.startswith('v_') and regs(p.split(None, 1)[1].split(',')[0]) & src: out.append((p, l))
                    ws += 1
                j -= 1
    return out
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for (i, name), (j, _) in zip(starts, starts[1:] + [(len(lines), '')]):
    if pat not in name: continue
    body = [l.strip() for l in lines[i:j]]
    body = body[:next((k for k, l in enumerate(body) if l.startswith('.Lfunc_end')), len(body))]
    ins = [l for l in body if l and not l.startswith(('.', ';')) and not l.endswith(':')]
    cnt = lambda pre: sum(1 for l in ins if l.startswith(pre))
    hz = hazards(ins)
    print('%-60s total %5d valu %5d mfma %3d ds %4d global %3d flat %3d scratch %3d s_nop %4d waitcnt %4d dpp-hazards %d' % (
        name[:60], len(ins), sum(1 for l in ins if l.startswith('v_') and not l.startswith('v_mfma')), cnt('v_mfma'), cnt('ds_'), cnt('global_'), cnt('flat_'), cnt('scratch_'),
        cnt('s_nop'), cnt('s_waitcnt'), len(hz)))
    for p, l in hz[:4]: print('    HAZARD', p, '->', l)
