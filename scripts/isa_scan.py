"""Static scan of the engine's gfx950 ISA (no GPU needed): per kernel / device function the flat, global and scratch memory operations.
   python scripts/isa_scan.py [api resid bg factor deconv ssub]
Why: a FLAT load (emitted when the compiler cannot tell a pointer's address space: a select between a global and a constant-address-space pointer,
a pointer read out of a device table, a generic parameter of a function that was not inlined) counts on lgkmcnt as well as vmcnt, so every
`s_waitcnt lgkmcnt(0)` behind an LDS read also waits for the flat loads in flight -- kernels that mix LDS reads with such loads run them one at a
time (scripts/probes/solve_gfill/README.md).  Scratch operations are register spills (or indexed private arrays)."""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tus = sys.argv[1:] or ["api", "resid", "bg", "factor", "deconv", "ssub"]
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
tmp = tempfile.mkdtemp()
print("%-7s %5s %6s %7s %5s  %s" % ("TU", "flat", "global", "scratch", "VGPR", "function (flat or scratch > 0)"))
for tu in tus:
    out = os.path.join(tmp, tu + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "--cuda-device-only", "-S", "-o", out,
                    os.path.join(ROOT, "cnmf_e_amd", "csrc", tu + ".hip")], check=True, stderr=subprocess.DEVNULL)
    cur, counts, order = None, {}, []
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1); counts[cur] = dict(flat=0, glob=0, scratch=0, vgpr=None); order.append(cur)
        elif cur:
            if re.search(r"\tflat_(load|store|atomic)", ln): counts[cur]["flat"] += 1
            elif "\tglobal_load" in ln: counts[cur]["glob"] += 1
            elif "\tscratch_" in ln: counts[cur]["scratch"] += 1
            m2 = re.match(r"^; NumVgprs: (\d+)", ln)
            if m2 and counts[cur]["vgpr"] is None: counts[cur]["vgpr"] = int(m2.group(1))
    for k in order:
        v = counts[k]
        if v["flat"] or v["scratch"]:
            name = subprocess.run([filt, k], capture_output=True, text=True).stdout.strip() if filt else k
            print("%-7s %5d %6d %7d %5s  %s" % (tu, v["flat"], v["glob"], v["scratch"], v["vgpr"], name[:110]))
