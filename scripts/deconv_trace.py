"""The foopsi / fminbnd sequence of ONE trace, engine against oracle (gamma differed by up to 2e-3 from identical inputs):
   python scripts/deconv_trace.py [--seed 3] [--T 3000] [--smin -5]
The engine prints its sequence from the kernel (option deconv_trace = k + 1, thread 0: 'DT ...' lines on stdout); the oracle's restatement is
instrumented here the same way.  The two listings are printed side by side up to the first line that differs by more than --tol."""
import argparse, io, os, sys, contextlib, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
ap = argparse.ArgumentParser(); ap.add_argument("--seed", type=int, default=3); ap.add_argument("--T", type=int, default=3000); ap.add_argument("--smin", type=float, default=-5.0)
ap.add_argument("--tol", type=float, default=1e-9); ap.add_argument("--traces", type=int, default=6)
a = ap.parse_args()
import oasis_oracle as oo

rng = np.random.default_rng(a.seed)
K, T = a.traces, a.T
Ctrue = np.zeros((K, T))
for k in range(K):
    g = 0.9 + 0.08 * rng.random()
    s = (rng.random(T) < 0.01) * (5 + 10 * rng.random(T))
    for t in range(1, T):
        Ctrue[k, t] = g * Ctrue[k, t - 1] + s[t]
Craw = (Ctrue + rng.standard_normal((K, T)) + 3.0).astype(np.float32)


def oracle_listing(y, smin):
    """deconvTemporal.m:45-50 for one trace with the instrumentation of the kernel"""
    out = []
    y = np.asarray(y, dtype=np.float64)
    sn = oo.GetSn(y)
    g = oo.estimate_time_constant_ar1(y, sn)
    sm = abs(smin) * sn if smin < 0 else smin
    gmax = np.exp(-1.0 / 100.0)
    b = oo.matlab_quantile(y, 0.15)
    sol, spks, aset = oo.oasisAR1(y - b, g, 0.0, sm)
    out.append(("first", dict(sn=sn, g=g, b=b, bsub=0.0, smin=sm, pools=len(aset))))
    optimize_g = True
    for it in range(10):
        b = float(np.mean(y - sol))
        out.append(("it", dict(it=it, b=b)))
        if not optimize_g or len(aset) == 0:
            break
        g0 = g
        if g > gmax:
            out.append(("gmax", {})); break
        # _update_g with the objective wrapped
        yy = y - b
        pools = [list(p) for p in aset]
        maxl = int(max(p[3] for p in pools))
        evals = []
        c = np.zeros_like(yy)
        def rss_g(gg):
            h = np.exp(np.log(gg) * np.arange(maxl + 1)); hh = np.cumsum(h * h)
            for (_, _, ti, li) in pools:
                ti = int(ti); li = int(li)
                seg = yy[ti - 1:ti - 1 + li]
                c[ti - 1:ti - 1 + li] = max(seg @ h[:li] / hh[li - 1], 0.0) * h[:li]
            r = yy - c
            f = float(r @ r)
            evals.append((gg, f))
            return f
        oo.fminbnd(rss_g, 0.0, 1.0)
        for i, (x, f) in enumerate(evals):
            out.append(("brent", dict(i=i, x=x, f=f)))
        sol, aset, g, spks = oo._update_g(yy, aset, 0.0, sm)
        out.append(("updated", dict(g=g, glast=evals[-1][0], pools=len(aset))))
        if abs(g - g0) / g0 < 1e-3:
            optimize_g = False
    return out


def engine_listing(k):
    import subprocess
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from cnmf_e_amd.engine import Engine; e = Engine(0); e.set_option('deconv_trace', %d); "
            "C = np.load(%r); e.deconv_temporal(C, dict(smin=%r, optimize_pars=True, optimize_b=True, max_tau=100.0)); e.synchronize(); e.close()"
            % (ROOT, k + 1, "/tmp/deconv_trace_in.npy", a.smin))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = []
    for ln in r.stdout.decode().splitlines():
        if not ln.startswith("DT "):
            continue
        w = ln.split()
        kind = w[1]
        vals = {}
        rest = w[2:]
        if kind in ("it", "brent"):
            vals["it" if kind == "it" else "i"] = int(rest[0]); rest = rest[1:]
        for i in range(0, len(rest) - 1, 2):
            vals[rest[i]] = float(rest[i + 1])
        out.append((kind, vals))
    if not out:
        print(r.stderr.decode()[-2000:])
    return out


np.save("/tmp/deconv_trace_in.npy", Craw)
worst = 0.0
for k in range(K):
    ol = oracle_listing(Craw[k], a.smin)
    el = engine_listing(k)
    print("==== trace %d: %d oracle lines, %d engine lines" % (k, len(ol), len(el)))
    first_bad = None
    for i, (o, e) in enumerate(zip(ol, el)):
        bad = o[0] != e[0]
        rels = []
        for key in o[1]:
            if key in e[1] and key not in ("it", "i", "bsub"):
                ov, ev = float(o[1][key]), float(e[1][key])
                rels.append(abs(ov - ev) / max(1e-300, abs(ov)))
        rel = max(rels) if rels else 0.0
        mark = ""
        if (bad or rel > a.tol) and first_bad is None:
            first_bad = i; mark = "   <-- first difference (rel %.2e)" % rel
        if first_bad is None or i <= first_bad + 3:
            print("  %-8s oracle %s\n  %-8s engine %s%s" % (o[0], {k_: (v if isinstance(v, int) else float("%.12g" % v)) for k_, v in o[1].items()}, "",
                                                             {k_: (v if isinstance(v, int) else float("%.12g" % v)) for k_, v in e[1].items()}, mark))
    go = [o[1]["g"] for o in ol if o[0] == "updated"]; ge = [e[1]["g"] for e in el if e[0] == "updated"]
    if go and ge:
        print("  final gamma: oracle %.10f engine %.10f  |d| = %.2e" % (go[-1], ge[-1], abs(go[-1] - ge[-1])))
        worst = max(worst, abs(go[-1] - ge[-1]))
print("worst |d gamma| = %.3e" % worst)
