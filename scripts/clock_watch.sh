#!/bin/bash
# shader clock / power while the background fit (Gram-dominated) runs in a loop: bash scripts/clock_watch.sh
python scripts/bg_only.py --cfg c3 --mode 3 --kernels 4 --reps 40 > /tmp/cw.log 2>&1 &
pid=$!
sleep 14
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" | tr '\n' ' '; echo
  sleep 0.4
done
wait $pid
tail -2 /tmp/cw.log
echo idle:; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' '; echo
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | head -40
