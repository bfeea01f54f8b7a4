#!/bin/bash
# round 2, GPU call A: the wave-per-pixel ring solve -- correctness (A/B against the LDS solver, GPU test suite) and timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python scripts/solve_ab.py --cfg small --modes 2,5,6 > gpurun_out/ab_small.log 2>&1; echo "ab_small rc=$?"
timeout 300 python scripts/solve_ab.py --cfg small --radius 18 --modes 2,5,6 > gpurun_out/ab_small_r18.log 2>&1; echo "ab_small_r18 rc=$?"
timeout 300 python scripts/solve_ab.py --cfg small --radius 8 --modes 2,5,6 > gpurun_out/ab_small_r8.log 2>&1; echo "ab_small_r8 rc=$?"
timeout 600 python scripts/solve_ab.py --cfg c3 --modes 2,5,6 --probes 0,1,2,4 > gpurun_out/ab_c3.log 2>&1; echo "ab_c3 rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_a.log 2>&1; echo "pytest rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"
for f in gpurun_out/ab_small.log gpurun_out/ab_small_r18.log gpurun_out/ab_small_r8.log; do tail -n 3 $f; done; cat gpurun_out/ab_c3.log; tail -n 15 gpurun_out/pytest_gpu_a.log; cut -c1-3000 gpurun_out/bench_a.json
