#!/bin/bash
# like ab.sh but without torch in the process (system HIP runtime): rate.py notorch
for v in "$@"; do
  echo -n "$v : "; env $v timeout -k 5 120 python experiments/rate.py notorch 2>&1 | tail -1
done
