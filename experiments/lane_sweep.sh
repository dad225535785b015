# in-process A/B of lane placements (FQL_LANE_<pass>) against the default: experiments/lane_sweep.sh [batch]
B=${1:-256}
for v in "FQL_LANE_bcf=2" "FQL_LANE_ct=0" "FQL_LANE_c1f=1" "FQL_LANE_bcf=2 FQL_LANE_c1f=1"; do
  echo "== B=$B $v"; timeout -k 10 200 python experiments/ab_inproc.py "-" "$v" 5 150 fp32 $B 2>&1 | tail -1
done
