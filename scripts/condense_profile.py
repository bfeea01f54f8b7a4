"""Condense scripts/profile_round.sh output (gpurun_out/prof_<tag>/) into profiles/<round>/*_<ver>.csv:
   python scripts/condense_profile.py r01v3 r01 v3"""
import csv, re, shutil, sys
tag, rnd, ver = sys.argv[1:4]
src = f"gpurun_out/prof_{tag}"
rows = list(csv.DictReader(open(f"{src}/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out, other = [], [0, 0.0]
for r in rows:
    n = r["Name"]
    if "cnmfe::" in n:
        short = re.sub(r"\(.*", "", n.replace("void ", "")).strip()
        out.append((short, int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"])))
    else:
        other[0] += int(r["Calls"]); other[1] += float(r["TotalDurationNs"])
with open(f"profiles/{rnd}/bench_c3_kernel_stats_{ver}.csv", "w") as g:
    g.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline  (5 iterations traced, the first one includes the one-off table of the video: k_build_bf, k_rowsum, k_gram4<0>; scripts/profile_round.sh)\n")
    g.write("# kernel argument lists stripped; all non-engine kernels (torch synthetic-video generation, rocsparse/torch glue in synth + host logic) summed in the last row\n")
    g.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for s, c, t, a, mn, mx in sorted(out, key=lambda x: -x[2]):
        g.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (s, c, t, a, 100 * t / tot, mn, mx))
    g.write('"(non-engine: torch/rocsparse kernels of synthetic data generation and host glue)",%d,%d,%.1f,%.2f,,\n' % (other[0], other[1], other[1] / max(1, other[0]), 100 * other[1] / tot))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    lines = open(f"{src}/pmc_{c}.csv").read().splitlines()
    keep = [lines[0]] + [l for l in lines[1:] if "cnmfe::" in l]
    keep = [re.sub(r"\(cnmfe::[^\"]*|\((float|HIP|long|int|double|unsigned)[^\"]*", "", l) for l in keep]
    open(f"profiles/{rnd}/bench_c3_pmc_{c}_{ver}.csv", "w").write(
        "# rocprofv3 --pmc %s --kernel-trace (own pass) -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline; per-launch mean; counter unit KiB.  gfx950: FETCH_SIZE of 16-B/lane coalesced reads is reported at 1/2 (MI355X_MICROARCH.md, HBM section)\n" % c + "\n".join(keep) + "\n")
shutil.copy(f"{src}/bench_under_rocprof.json", f"profiles/{rnd}/bench_c3_under_rocprof_{ver}.json")
