#!/bin/bash
# Round profile set (run through gpurun from the repo root): kernel trace + stats, then the PMC passes, each in its own run
# (no trace domains beside --pmc).  usage: tools/profile.sh <tag, e.g. r02>   -> gpurun_out/<tag>_prof/...
set -e
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
# Per-kernel numbers (durations, counters) are taken on the captured-graph path (FQL_AQL=0): rocprofv3 serialises a HIP stream's dispatches, so a
# kernel's average there is its duration ALONE on the chip - the quantity bench.py's roofline uses (fql_profile_update) and the counters are per
# launch either way.  The engine's own AQL queues (the default dispatch) are traced as they run, three lanes at once, in a pass of their own below;
# the tool's signal interception stretches the gaps between lanes there, the kernels' own durations beside the other lanes are real.
export FQL_AQL=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 300 --warmup 50 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -- $B --steps 50 --warmup 10 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -- $B --steps 50 --warmup 10 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/sq -- $B --steps 50 --warmup 10 > $OUT/pmc_sq.log 2>&1
unset FQL_AQL
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/aql_stats -- $B --steps 300 --warmup 50 > $OUT/aql_stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/aql_stats/*/*kernel_stats.csv | head -1) $OUT/aql_kernel_stats.csv
python3 tools/trace.py $OUT/stats -v > $OUT/timeline.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json "rocprofv3 --pmc passes of 'bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras' (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE: separate runs), tools/pmc_summary.py" $OUT/kernel_stats.csv > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
tail -5 $OUT/timeline.txt; cat $OUT/pmc_summary.txt
