for v in 0 1 0 1; do
  if [ $v = 1 ]; then export SGC_SKIP_EPI=1; else unset SGC_SKIP_EPI; fi
  echo -n "skip=$v "
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print(d['ms_per_step'], k['fc1_fwd'], k['fc1_dgrad'], k['fc1_wgrad'], k['conv2_wgrad'])"
done
