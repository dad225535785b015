#!/bin/bash
# A/B of scheduling knobs under precision = bf16x3 (and fp32) on ONE box (box-to-box spread is ~3 %): updates/s
b() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras "$@" 2>/tmp/ab_err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || tail -2 /tmp/ab_err | cut -c1-200; }
for side in 4 2 8 1; do for ch in 4 2 8; do echo "x3 side gm=$side chain gm=$ch: $(FQL_XCD_GM=$side FQL_XCD_GM_CHAIN=$ch b --precision bf16x3)"; done; done
echo "x3 FQL_NO_XCD=1: $(FQL_NO_XCD=1 b --precision bf16x3)"
echo "f32 side 4 chain 4: $(FQL_XCD_GM=4 FQL_XCD_GM_CHAIN=4 b)  side 4 chain 2: $(FQL_XCD_GM=4 FQL_XCD_GM_CHAIN=2 b)  side 8 chain 4: $(FQL_XCD_GM=8 FQL_XCD_GM_CHAIN=4 b)  none: $(FQL_NO_XCD=1 b)"
