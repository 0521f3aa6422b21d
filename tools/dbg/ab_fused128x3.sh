for rep in 1 2; do for f in 1 0; do
AOD_FUSE_BOTTLENECK128_X3=$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-precision-check --phase-iters 20 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('fused128x3=$f rep=$rep', d['value'], d['ms_per_step'], d['config']['ms_per_step_scores_read_every_step'], d['phase_rates']['train_ms_per_batch'], d['phase_rates']['score_ms_per_batch'])"
done; done
