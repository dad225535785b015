#!/bin/bash
# A/B of scheduling knobs under precision = bf16x3 (and fp32) on ONE box (box-to-box spread is ~3 %): updates/s
b() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras "$@" 2>/tmp/ab_err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || tail -2 /tmp/ab_err | cut -c1-200; }
echo "x3 default:               $(b --precision bf16x3) $(b --precision bf16x3)"
echo "x3 bc pass on a 4th lane: $(FQL_LANE_bcf=3 FQL_LANE_bc=3 b --precision bf16x3) $(FQL_LANE_bcf=3 FQL_LANE_bc=3 b --precision bf16x3)"
echo "x3 c1 pass on a 4th lane: $(FQL_LANE_c1f=3 FQL_LANE_c1=3 b --precision bf16x3)"
echo "f32 default:              $(b) $(b)"
echo "f32 bc pass on a 4th lane:$(FQL_LANE_bcf=3 FQL_LANE_bc=3 b) $(FQL_LANE_bcf=3 FQL_LANE_bc=3 b)"
