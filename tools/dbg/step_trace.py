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
tiny = [i for i in range(a, b) if E[i] - S[i] < 8000]
print(len(tiny), 'kernels < 8 us:', sum(E[i] - S[i] for i in tiny) / 1e3, 'us')
for n, c in collections.Counter(names[i][:90] for i in tiny).most_common(40):
    print(f'  {c:3d}  {n}')
if len(sys.argv) > 2:
    for i in range(a, b):
        print(f'{i - a:4d} {(S[i] - S[a]) / 1e3:9.1f} +{(E[i] - S[i]) / 1e3:7.1f} q{rows[i]["Queue_Id"]} {names[i][:100]}')
