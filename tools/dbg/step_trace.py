"""One steady-state step of a rocprofv3 --kernel-trace CSV of bench.py: every kernel shorter than 8 us in launch order with its neighbours
(the launch-bound glue between the conv kernels), the idle gaps, and totals.  usage: step_trace.py out_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
S = [int(r['Start_Timestamp']) for r in rows]
E = [int(r['End_Timestamp']) for r in rows]
names = [r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '') for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith('sgd_multi_kernel')]
# steps = spans between the first SGD launch of consecutive iterations (two optimizers -> several launches per step: take gaps > 5 ms)
starts = [marks[0]] + [b for a, b in zip(marks[:-1], marks[1:]) if S[b] - S[a] > 5e6]
per = [(S[b] - S[a]) / 1e6 for a, b in zip(starts[:-1], starts[1:])]
med = sorted(per)[len(per) // 2]
k = max(i for i, p in enumerate(per) if abs(p - med) < 0.03 * med)
a, b = starts[k], starts[k + 1]
print(f'step {k}: {b - a} kernels, {(S[b] - S[a]) / 1e6:.3f} ms (median {med:.3f})')
iv = sorted((S[i], E[i]) for i in range(a, b))
cs, ce = iv[0]
busy, idle = 0, 0
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; idle += s - ce; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f'busy {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms')
CONV = ('conv_igemm', 'conv_x3p', 'conv_wgrad', 'bottleneck', 'stem_pool', 'splitk', 'unpack_wgrad', 'halo_x3')
t_conv = sum(E[i] - S[i] for i in range(a, b) if any(x in names[i] for x in CONV))
t_else = sum(E[i] - S[i] for i in range(a, b)) - t_conv
print(f'conv-class kernels (igemm, x3p, wgrad, fused blocks, stem, halo, split-K finalize, unpack) {t_conv / 1e6:.3f} ms, everything else {t_else / 1e6:.3f} ms')
other = collections.Counter()
for i in range(a, b):
    if not any(x in names[i] for x in CONV):
        other[names[i].split('(')[0][:70]] += E[i] - S[i]
print('everything else, by kernel: ' + ', '.join(f'{n} {v / 1e3:.0f} us' for n, v in other.most_common(12)))
tiny = [i for i in range(a, b) if E[i] - S[i] < 8000]
print(len(tiny), 'kernels < 8 us:', sum(E[i] - S[i] for i in tiny) / 1e3, 'us')
for n, c in collections.Counter(names[i][:90] for i in tiny).most_common(40):
    print(f'  {c:3d}  {n}')
if len(sys.argv) > 2:
    for i in range(a, b):
        print(f'{i - a:4d} {(S[i] - S[a]) / 1e3:9.1f} +{(E[i] - S[i]) / 1e3:7.1f} q{rows[i]["Queue_Id"]} {names[i][:100]}')
