"""Per-kernel register / LDS / spill table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or a file).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2>&1 | python scripts/resusage.py [filter]"""
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None; rows = {}
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
keys = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
for k, v in rows.items():
    if flt in k:
        print("%-60s" % k[:60], " ".join("%s=%s" % (a.split(" [")[0].replace(" ", ""), v.get(a, "?")) for a in keys))
