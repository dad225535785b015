# in-process A/B of lane placements (FQL_LANE_<pass>) against the default, B = 256: experiments/lane_sweep.sh
for v in "FQL_LANE_bcf=1 FQL_LANE_bc=1 FQL_LANE_c1f=1 FQL_LANE_c1=1" "FQL_LANE_bcf=1" "FQL_LANE_os=2 FQL_LANE_ct=2 FQL_LANE_c2=2" ; do
  echo "== $v"; timeout -k 10 120 python experiments/ab_inproc.py "-" "$v" 5 300 2>&1 | tail -2
done
