R=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2; do
  for setting in "DEFAULT=1" "CATSEG_PREP_ASYNC=0"; do
    ms=$(env $setting python3 $R/bench.py --eager --steps 12 --warmup 4 --no-cpu-baseline --no-side-figures --no-roofline 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "eager round $i $setting: $ms ms/step"
  done
done
CATSEG_PREP_ASYNC=0 python3 $R/tools/stage_times.py 2>&1 | grep "region_\|TOTAL" | tail -12
