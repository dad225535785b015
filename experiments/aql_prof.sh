# rocprofv3 kernel trace of the AQL path across ring wraps (4096 packets per ring, 44 per update on lane 0)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/aqlprof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/aqlprof.log 2>&1
echo "rc=$?"
