#!/bin/bash
# HBM-side traffic of the bench kernels (separate --pmc pass, no other trace domain); summary under gpurun_out/pmc_bench_$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/pmc_bench_$1
mkdir -p $D
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $D -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-graph > $D/bench.json 2>$D/err.txt
python3 - <<PY
import csv, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open('$D/out_counter_collection.csv')):
    k = r['Kernel_Name'].split('(')[0][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
out = {}
for k, d in agg.items():
    n = cnt[(k, 'TCC_EA0_RDREQ_sum')]
    rd = d['TCC_EA0_RDREQ_sum']; rd32 = d['TCC_EA0_RDREQ_32B_sum']; wr = d['TCC_EA0_WRREQ_sum']; wr64 = d['TCC_EA0_WRREQ_64B_sum']
    # guide: FETCH_SIZE = RDREQ x 64 B under-reports wide coalesced reads by 2x on gfx950 -> 128 B per non-32B request; writes: 64-B requests exact
    rbytes = (rd - rd32) * 128 + rd32 * 32
    wbytes = wr64 * 64 + (wr - wr64) * 32
    out[k] = dict(launches=n, read_MB_per_launch=rbytes / n / 1e6, write_MB_per_launch=wbytes / n / 1e6)
top = sorted(out.items(), key=lambda kv: -(kv[1]['read_MB_per_launch'] + kv[1]['write_MB_per_launch']) * kv[1]['launches'])[:14]
for k, v in top: print('%-60s n=%5d  read %8.2f MB  write %8.2f MB per launch' % (k, v['launches'], v['read_MB_per_launch'], v['write_MB_per_launch']))
json.dump(out, open('$D/traffic.json', 'w'), indent=1)
PY
