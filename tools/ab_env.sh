# same-box A/B of environment switches: bash tools/ab_env.sh "<VAR=val ...>" "<VAR=val ...>" ... ; each setting runs bench.py twice, alternating
set -o pipefail
for i in 1 2; do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" || exit 1
  done
done
