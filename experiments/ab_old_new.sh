#!/bin/bash
b() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras "$@" 2>/tmp/ab_err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || tail -2 /tmp/ab_err | cut -c1-200; }
for r in 1 2 3; do
echo "new f32: $(b)  x3: $(b --precision bf16x3)"
echo "old f32: $(FQL_AMD_LIB=experiments/libfql_old.so b)  x3: $(FQL_AMD_LIB=experiments/libfql_old.so b --precision bf16x3)"
done
