# the data-parallel step at world size 1 with the collectives forced (RCCL all-reduce of the gradient buffer every step): its fixed cost
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 FQL_BENCH_FORCE_DP=1 FQL_DP_ALWAYS_REDUCE=1
for ov in 1 0; do
  FQL_DP_OVERLAP=$ov python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$ov', d['value'], d['ms_per_step'], d.get('data_parallel_step'))"
done
unset FQL_BENCH_FORCE_DP FQL_DP_ALWAYS_REDUCE
python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', d['value'], d['ms_per_step'], d['dispatch'])"
