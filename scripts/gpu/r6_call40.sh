#!/bin/bash
# round 6, call 40: the whole GPU suite on the round's final code
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/gputest_v4.txt
cat gpurun_out/r06/gputest_v4.txt
