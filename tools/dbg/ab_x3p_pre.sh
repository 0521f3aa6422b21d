for i in 1 2 3; do for v in 0 1; do
  AOD_X3P_PRE=$v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-precision-check 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('AOD_X3P_PRE=$v', d['value'], 'img/s', d['ms_per_step'], 'ms/step  train', d['phase_rates']['train_ms_per_batch'], 'score', d['phase_rates']['score_ms_per_batch'])"
done; done
