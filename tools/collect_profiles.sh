#!/bin/bash
# Evidence collection for profiles/ (run on the GPU box through gpurun; each rocprofv3 pass is its own process, counters in
# their own passes with --kernel-trace only).  usage: bash tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/...
set -o pipefail
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary"
PU_NO_SIDE_STREAM=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- $B --steps 5 --warmup 2 > $O/serial.log 2>&1 && echo serial ok &&
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -o overlap -- $B --steps 5 --warmup 2 > $O/overlap.log 2>&1 && echo overlap ok &&
PU_NO_SIDE_STREAM=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $B --steps 1 --warmup 1 > $O/fetch.log 2>&1 && echo fetch ok &&
PU_NO_SIDE_STREAM=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $B --steps 1 --warmup 1 > $O/write.log 2>&1 && echo write ok &&
PU_NO_SIDE_STREAM=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o sq -- $B --steps 1 --warmup 1 > $O/sq.log 2>&1 && echo sq ok &&
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sample -o sample -- python3 $R/bench.py --mode sample --steps 5 --warmup 3 > $O/sample.log 2>&1 && echo sample ok
cd $R
python3 tools/pmc_summary.py $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv $O/pmc_traffic.json
python3 tools/pmc_sq_summary.py $O/sq/sq_counter_collection.csv $O/sq/sq_kernel_trace.csv $O/pmc_sq_mfma.json > $O/pmc_sq.txt
find $O -name "*.csv" -size +3M -delete      # keep the stats / small counter files only (64 MiB merge cap)
ls -la $O/*/* | head -60
