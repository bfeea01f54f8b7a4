#!/bin/bash
# round 6, call 21: the whole GPU suite on the staged solve + three-plane temporal projection defaults
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06/gputest_v3.txt
cat gpurun_out/r06/gputest_v3.txt
