#!/bin/bash
# usage: ab.sh "ENV1=.. ENV2=.." "..." : one bench line per variant (2000 steps)
for v in "$@"; do
  env $v python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['whole_update']['kernel_launches_per_update'])"
done
