#!/bin/bash
# VALU utilisation of the HUA kernels in the bench's scoring phase (separate --pmc passes, kernel trace only); writes
# gpurun_out/pmc_hua_$1/hua_pmc.json = {kernel: {counter: per-launch mean}, valu_util: ...} -- copy to profiles/r02_hua_pmc.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/pmc_hua_$1
mkdir -p $D
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $D/g$i -o out --output-format csv -- python3 $R/bench.py --mode score --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-graph > $D/bench_$i.json 2>$D/err_$i.txt
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob('$D/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        if 'hua' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for f in glob.glob('$D/g1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        if 'hua' in k: dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
out = {}
for k, d in agg.items():
    out[k] = {c: v / cnt[(k, c)] for c, v in d.items()}
    if dur[k]: out[k]['avg_us'] = sum(dur[k]) / len(dur[k]) / 1e3
s = next((v for k, v in out.items() if 'hua_sample' in k), None)
res = dict(kernels=out)
if s and s.get('SQ_BUSY_CYCLES'):
    # SQ_* cycle counters tick in quad-cycles summed over the SEs/XCDs that were busy; utilisation as the guide's VALUBusy ratio:
    # cycles with a VALU instruction active / cycles the wave slots were busy
    res['valu_util'] = round(s['SQ_ACTIVE_INST_VALU'] / max(s['SQ_ACTIVE_INST_ANY'] + s['SQ_WAIT_ANY'] + s['SQ_WAIT_INST_ANY'], 1.0), 4)
    res['valu_active_over_wave_cycles'] = round(s['SQ_ACTIVE_INST_VALU'] / max(s['SQ_WAVE_CYCLES'], 1.0), 4)
    res['valu_insts_per_launch'] = s.get('SQ_INSTS_VALU')
    # issue-rate view: wave-level VALU instructions / (kernel time x 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction)
    if s.get('avg_us') and s.get('SQ_INSTS_VALU'):
        res['valu_issue_frac_of_peak'] = round(s['SQ_INSTS_VALU'] / (s['avg_us'] * 1e-6 * 1024 * 2.4e9 / 2), 4)
json.dump(res, open('$D/hua_pmc.json', 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
