R=${GRAFT_REPO_ROOT:-$PWD}
for m in ocrnet_r50 deeplabv3plus_r50; do
  for setting in "DEFAULT=1" "CATSEG_G1=0" "CATSEG_G1=0 CATSEG_P1=0"; do
    ms=$(env $setting python3 $R/bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-side-figures --no-roofline 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$m $setting: $ms ms/step"
  done
done
for setting in "DEFAULT=1" "CATSEG_G1_MIN_ROWS=8192" "CATSEG_G1_MIN_ROWS=4096"; do
  ms=$(env $setting python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-figures --no-roofline 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "hrnet48 $setting: $ms ms/step"
done
