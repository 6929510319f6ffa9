#!/bin/bash
# Collects the rocprofv3 evidence for the benchmark's dominant kernel on the GPU
# box: one kernel-trace/stats run and separate --pmc passes (counters never mixed
# with the stats run).  Usage (from the repo root): bash profiles/run_profile.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/stats.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -40
