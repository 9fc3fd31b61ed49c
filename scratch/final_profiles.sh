#!/bin/bash
# Final evidence collection for profiles/ (run on the GPU box through gpurun).  Each rocprofv3 pass is its own process.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PU_NO_SIDE_STREAM=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/serial.log 2>&1 && echo serial ok &&
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -o overlap -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/overlap.log 2>&1 && echo overlap ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 && echo fetch ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 && echo write ok &&
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msssim -o msssim -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --recon wmse_msssim --members 1 > $O/msssim.log 2>&1 && echo msssim ok &&
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sample -o sample -- python $R/bench.py --mode sample --steps 5 --warmup 2 > $O/sample.log 2>&1 && echo sample ok
cd $R
find $O -name "*.csv" -size +3M -delete      # keep the stats / small counter files only (64 MiB merge cap)
ls -la $O/*/* | head -40
# default bench run (with the bounded cpu_baseline) -> JSON line for profiles/r1_final_bench.json
timeout -k 10 500 python $R/bench.py > $O/default_bench.log 2>&1 && echo bench ok
timeout -k 10 200 python $R/bench.py --flat-adamw --no-cpu-baseline --steps 20 --warmup 5 > $O/flat_bench.log 2>&1 && echo flat ok
timeout -k 10 200 python $R/bench.py --recon wmse_msssim --members 1 --no-cpu-baseline --steps 20 --warmup 5 > $O/msssim_bench.log 2>&1 && echo ms ok
timeout -k 10 200 python $R/bench.py --mode sample --steps 10 > $O/sample_bench.log 2>&1 && echo sample ok
timeout -k 10 200 python $R/bench.py --mode sample --hr --steps 10 > $O/sample_hr_bench.log 2>&1 && echo samplehr ok
