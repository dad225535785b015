#!/bin/bash
# usage: experiments/ab_env.sh VAR=value   -> three alternating runs per precision with and without the setting
b() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras "$@" 2>/tmp/ab_err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || tail -2 /tmp/ab_err | cut -c1-300; }
for r in 1 2 3; do
echo "with $1   f32: $(env $1 bash -c "$(declare -f b); b")  x3: $(env $1 bash -c "$(declare -f b); b --precision bf16x3")"
echo "without    f32: $(b)  x3: $(b --precision bf16x3)"
done
