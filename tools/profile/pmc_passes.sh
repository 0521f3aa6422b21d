#!/bin/bash
# The documented SECOND command behind bench.py's `roofline.traffic` and `hua.valu_util` (counters cannot be collected inside a timed
# run): separate rocprofv3 --pmc passes (kernel trace only, no other trace domain) of the bench on the GPU box.  Writes
#   profiles/pmc_traffic.json  HBM-side bytes per launch of every kernel (TCC_EA0_RDREQ/WRREQ, corrected as MI355X_MICROARCH.md prescribes)
#   profiles/pmc_hua.json      VALU utilisation of the HUA sampler
# each tagged with `kernels_sha16` = aod_meh_hua_amd.build.source_digest(); bench.py merges them only when the tag equals its own build's.
#   gpurun -- 'bash tools/profile/pmc_passes.sh'   then copy gpurun_out/profiles/pmc_*.json to profiles/ and commit
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/pmc_passes
mkdir -p $D $R/gpurun_out/profiles
SHA=$(cd $R && python3 -c "from aod_meh_hua_amd.build import source_digest; print(source_digest())")
# (one traffic pass per precision mode: pmc_traffic.json = the bench's default, reference-precision mode; pmc_traffic_bf16.json = the fast mode)
for PREC in bf16x3 bf16; do
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $D/traffic_$PREC -o out --output-format csv -- python3 $R/bench.py --precision $PREC --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-graph --phase-iters 2 > $D/bench_traffic_$PREC.json 2>$D/err_traffic_$PREC.txt
done
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $D/hua$i -o out --output-format csv -- python3 $R/bench.py --mode score --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-graph --phase-iters 2 > $D/bench_hua$i.json 2>$D/err_hua$i.txt
done
python3 - <<PY
import csv, glob, collections, json
sha = '$SHA'
# ---- HBM-side traffic per launch
for prec, fname in (('bf16x3', 'pmc_traffic.json'), ('bf16', 'pmc_traffic_bf16.json')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('$D/traffic_%s/**/*counter_collection.csv' % prec, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:72]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    out = {}
    for k, d in agg.items():
        n = cnt[(k, 'TCC_EA0_RDREQ_sum')]
        if not n: continue
        rd, rd32, wr, wr64 = d['TCC_EA0_RDREQ_sum'], d['TCC_EA0_RDREQ_32B_sum'], d['TCC_EA0_WRREQ_sum'], d['TCC_EA0_WRREQ_64B_sum']
        # guide: FETCH_SIZE = RDREQ x 64 B under-reports wide coalesced reads by 2x on gfx950 -> 128 B per non-32B request; writes: 64-B requests exact
        out[k] = dict(launches=n, read_MB_per_launch=((rd - rd32) * 128 + rd32 * 32) / n / 1e6, write_MB_per_launch=(wr64 * 64 + (wr - wr64) * 32) / n / 1e6)
    json.dump(dict(kernels_sha16=sha, precision=prec,
                   command='rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -- python3 bench.py --precision %s --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-graph' % prec,
                   kernels=out), open('$R/gpurun_out/profiles/' + fname, 'w'), indent=1)
    print('----', prec)
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]['read_MB_per_launch'] + kv[1]['write_MB_per_launch']) * kv[1]['launches'])[:16]:
        print('%-72s n=%5d  read %8.2f MB  write %8.2f MB per launch' % (k, v['launches'], v['read_MB_per_launch'], v['write_MB_per_launch']))
# ---- HUA sampler VALU utilisation
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob('$D/hua*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:48]
        if 'hua' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for f in glob.glob('$D/hua1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:48]
        if 'hua' in k: dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
ko = {}
for k, d in agg.items():
    ko[k] = {c: v / cnt[(k, c)] for c, v in d.items()}
    if dur[k]: ko[k]['avg_us'] = sum(dur[k]) / len(dur[k]) / 1e3
s = next((v for k, v in ko.items() if 'hua_sample' in k), None)
res = dict(kernels_sha16=sha, kernels=ko)
if s and s.get('SQ_BUSY_CYCLES'):
    res['valu_util'] = round(s['SQ_ACTIVE_INST_VALU'] / max(s['SQ_ACTIVE_INST_ANY'] + s['SQ_WAIT_ANY'] + s['SQ_WAIT_INST_ANY'], 1.0), 4)
    res['valu_active_over_wave_cycles'] = round(s['SQ_ACTIVE_INST_VALU'] / max(s['SQ_WAVE_CYCLES'], 1.0), 4)
    res['valu_insts_per_launch'] = s.get('SQ_INSTS_VALU')
    if s.get('avg_us') and s.get('SQ_INSTS_VALU'):      # wave-level VALU instructions / (time x 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction)
        res['valu_issue_frac_of_peak'] = round(s['SQ_INSTS_VALU'] / (s['avg_us'] * 1e-6 * 1024 * 2.4e9 / 2), 4)
json.dump(res, open('$R/gpurun_out/profiles/pmc_hua.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'kernels'}))
PY
# ---- effective clock and matrix-pipe occupancy of the conv kernels (round 5: the reference-precision step is bound by what the chip delivers
# at the clock it holds under matrix load, not by HBM -- DESIGN 10).  One more pass, counters only.
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $D/clock -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-graph --phase-iters 2 > $D/bench_clock.json 2>$D/err_clock.txt
python3 - <<PY
import csv, glob, collections, json
sha = '$SHA'
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:72]
for f in glob.glob('$D/clock/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = name(r); agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for f in glob.glob('$D/clock/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[name(r)].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
out = {}
for k, d in agg.items():
    n = cnt[(k, 'GRBM_GUI_ACTIVE')]
    if not n or not dur[k]: continue
    us = sum(dur[k]) / len(dur[k]) / 1e3
    gui = d['GRBM_GUI_ACTIVE'] / n                      # summed over the 8 XCDs (MI355X_MICROARCH.md, 'DVFS give-back')
    cyc = gui / 8.0
    e = dict(launches=n, avg_us=round(us, 2), effective_clock_ghz=round(cyc / (us * 1e3), 3) if us >= 100 else None,
             mfma_busy_frac=round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / n / max(cyc * 1024.0, 1.0), 4),
             wave_cycles_waiting=round(d['SQ_WAIT_ANY'] / max(d['SQ_WAVE_CYCLES'], 1.0), 4),
             wave_cycles_issue_stalled=round(d['SQ_WAIT_INST_ANY'] / max(d['SQ_WAVE_CYCLES'], 1.0), 4),
             wave_cycles_issuing=round(d['SQ_ACTIVE_INST_ANY'] / max(d['SQ_WAVE_CYCLES'], 1.0), 4))
    out[k] = e
json.dump(dict(kernels_sha16=sha, note='effective clock = GRBM_GUI_ACTIVE / 8 / duration (reads high below ~0.1 ms: omitted there); mfma_busy_frac = '
               'SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)', kernels=out), open('$R/gpurun_out/profiles/pmc_clock.json', 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['launches'])[:14]:
    print('%-72s n=%4d %8.1f us  clock %s GHz  mfma busy %.3f  waiting %.2f' % (k, v['launches'], v['avg_us'], v['effective_clock_ghz'], v['mfma_busy_frac'], v['wave_cycles_waiting']))
PY
