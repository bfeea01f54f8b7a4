#!/bin/bash
# round 6, call 42: the whole GPU suite on the final code (after the split Welch transform)
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | grep -a "passed\|failed\|error" | tail -5 > gpurun_out/r06/gputest_v5.txt
cat gpurun_out/r06/gputest_v5.txt
