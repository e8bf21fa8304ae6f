#!/bin/bash
# Regression check for the interpreter-exit crash seen in round 1 ("terminate called without an active exception", rc 134,
# about 1 in 7 runs of a script with a backward pass): N consecutive runs of the stepwise drop-in loop with backward.
N=${1:-50}
fail=0
for i in $(seq 1 $N); do
  python3 tools/bench_stepwise.py > /tmp/exit_check.out 2> /tmp/exit_check.err
  rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "run $i rc=$rc: $(tail -2 /tmp/exit_check.err)"; fi
  if [ $((i % 10)) -eq 0 ]; then echo "done $i runs, failures so far: $fail"; fi
done
echo "exit_check: $N runs, $fail failures; last output: $(tail -1 /tmp/exit_check.out)"
[ $fail -eq 0 ]
