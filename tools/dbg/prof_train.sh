cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_train
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o out --output-format csv -- python3 $R/bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-precision-check > $D/bench.json 2>/dev/null
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$D/out_kernel_stats.csv')))
for r in rows[:40]:
    print("%-80s %6d calls %9.1f us avg %6.2f%%" % (r["Name"][:80], int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
