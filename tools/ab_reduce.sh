set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for i in 1 2; do
  for v in OLD NEW; do
    if [ $v = OLD ]; then export PU_WG_REDUCE_OLD=1; else unset PU_WG_REDUCE_OLD; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'])" || exit 1
  done
done
