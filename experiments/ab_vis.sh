#!/bin/bash
# usage: ab_vis.sh "ENV1=.. ENV2=.." "..." : one visual-workload bench line per variant (200 steps)
for v in "$@"; do
  env $v python bench.py --workload visual --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['whole_update']['kernel_launches_per_update'])"
done
