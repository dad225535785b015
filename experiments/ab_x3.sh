#!/bin/bash
# A/B of lane placements (FQL_LANE_<pass>) under precision = bf16x3 on ONE box (box-to-box spread is ~3 %): updates/s, two runs each
b() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --precision bf16x3 2>/tmp/ab_err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>/dev/null || tail -2 /tmp/ab_err | cut -c1-200; }
echo "default:        $(b) $(b)"
echo "bcf=1:          $(FQL_LANE_bcf=1 b) $(FQL_LANE_bcf=1 b)"
echo "ct=2:           $(FQL_LANE_ct=2 b)"
echo "bcf=1 bc=1:     $(FQL_LANE_bcf=1 FQL_LANE_bc=1 b)"
echo "c1f=1:          $(FQL_LANE_c1f=1 b)"
echo "bcf=1 c1f=1:    $(FQL_LANE_bcf=1 FQL_LANE_c1f=1 b)"
echo "os=2:           $(FQL_LANE_os=2 b)"
echo "default:        $(b)"
