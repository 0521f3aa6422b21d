for fuse in 1 0; do for steps in 1 3; do
AOD_FUSE_BOTTLENECK128_X3=$fuse AOD_CONV_PREC=bf16x3 MULTIRANK_LR=${LR:-2e-4} MULTIRANK_STEPS=$steps MULTIRANK_DETERMINISTIC=1 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2957$steps tests/multirank_worker.py 2>&1 | grep MULTIRANK | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l[len('MULTIRANK '):]); print('fuse=$fuse steps=$steps', d['graph_vs_mean_gradient_run'], d['eager_vs_mean_gradient_run'], d['eager_worst_keys'])
"
done; done
