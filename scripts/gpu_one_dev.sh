#!/bin/bash
# bench.py --config c4 with N ranks on ONE device (gloo; CNMFE_BENCH_ONE_DEVICE=1): checks the sharded path end to end and shows the per-rank host cost
mkdir -p gpurun_out/r02
for n in "$@"; do
  CNMFE_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n bench.py --gpus $n --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02/one_dev_$n.json 2> gpurun_out/r02/one_dev_$n.err
  python - gpurun_out/r02/one_dev_$n.json <<'PY'
import sys, json
for l in open(sys.argv[1]).read().strip().splitlines():
    try: d = json.loads(l)
    except Exception: continue
    print(d["n_gpus"], "ranks: %.1f ms/step, kernel sum (rank 0) %s" % (d["ms_per_step"], d["kernel_sum_ms_per_step"]), d["config"]["workload"][:100])
PY
  tail -n 2 gpurun_out/r02/one_dev_$n.err
done
