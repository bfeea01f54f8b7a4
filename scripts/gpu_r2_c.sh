#!/bin/bash
# round 2, GPU call C: bench modes -- default (c3), c4 on one GPU, demo sequence, cold start, 2-rank code path on one device (gloo)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1], "value", round(d["value"],3), d.get("unit"), "ms_per_step", round(d.get("ms_per_step",0),2), "kernel_sum", d.get("kernel_sum_ms_per_step"))
ks=d.get("kernels_ms_per_step") or d.get("kernels_ms_total")
print("   ", {k:v for k,v in list(ks.items())[:12]})
if "first_iteration" in d: print("    first", d["first_iteration"]["ms"], d["first_iteration"]["one_off_kernels_ms"])
if "second_fit" in d: print("    second_fit", d["second_fit"])
PY
}
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "c3 rc=$?"; show gpurun_out/bench_c3.json
timeout 600 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c4_n1.json 2> gpurun_out/bench_c4_n1.err; echo "c4 rc=$?"; show gpurun_out/bench_c4_n1.json
timeout 600 python bench.py --demo-sequence > gpurun_out/bench_demo_seq.json 2> gpurun_out/bench_demo_seq.err; echo "demo rc=$?"; show gpurun_out/bench_demo_seq.json
timeout 600 python bench.py --warmup 0 --steps 5 --no-cpu-baseline > gpurun_out/bench_c3_cold.json 2> gpurun_out/bench_c3_cold.err; echo "cold rc=$?"; show gpurun_out/bench_c3_cold.json
CNMFE_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --config c4tiny --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c4tiny_2r.json 2> gpurun_out/bench_c4tiny_2r.err; echo "2rank rc=$?"; tail -n 3 gpurun_out/bench_c4tiny_2r.err; show gpurun_out/bench_c4tiny_2r.json
timeout 300 python bench.py --config c4tiny --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c4tiny_1r.json 2>/dev/null; show gpurun_out/bench_c4tiny_1r.json
