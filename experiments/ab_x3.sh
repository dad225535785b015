#!/bin/bash
b() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras "$@" 2>/tmp/ab_err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || tail -2 /tmp/ab_err | cut -c1-200; }
echo "x3:  $(b --precision bf16x3) $(b --precision bf16x3) $(b --precision bf16x3)"
echo "f32: $(b) $(b) $(b)"
