#!/bin/bash
# Data-parallel step at world size 1 with both collectives forced (RCCL launches with nothing to exchange): the fixed cost of the DP step, updates/s
r() { timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 2000 --warmup 200 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
export FQL_BENCH_FORCE_DP=1 FQL_DP_ALWAYS_REDUCE=1
echo "fp32   overlapped: $(r)   plain: $(FQL_DP_OVERLAP=0 r)"
echo "bf16x3 overlapped: $(r --precision bf16x3)   plain: $(FQL_DP_OVERLAP=0 r --precision bf16x3)"
