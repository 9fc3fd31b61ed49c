# same-box A/B of two builds of the library: PU_LIB_PATH=<prev .so> against the in-tree one.  usage: bash tools/ab_lib.sh <prev.so> [rounds]
set -o pipefail
PREV=$1; N=${2:-2}
for i in $(seq $N); do
  for v in PREV NEW; do
    if [ $v = PREV ]; then export PU_LIB_PATH=$PREV; else unset PU_LIB_PATH; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])" || exit 1
  done
done
