for pair in "FQL_LANE_bcf=1|-" "-|FQL_LANE_bcf=1" "FQL_LANE_bcf=1|FQL_LANE_bcf=0"; do
  a="${pair%%|*}"; b="${pair##*|}"
  echo "== A: $a   B: $b"; timeout -k 10 200 python experiments/ab_inproc.py "$a" "$b" 9 300 2>&1 | tail -3
done
echo "== bf16x3"; timeout -k 10 200 python experiments/ab_inproc.py "-" "FQL_LANE_bcf=1" 7 300 bf16x3 2>&1 | tail -3
