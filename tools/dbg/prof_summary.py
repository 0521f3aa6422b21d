"""rocprofv3 --stats kernel summary of a bench.py run, per scoring pass (= per bench step: the run makes as many training iterations as
scoring passes).  The bench's own calibration launches (torch's spin kernel behind the measured-peak timers, the library GEMM of
roofline.measured_peaks) are listed apart and not counted."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calib = lambda n: n.startswith('at::cuda::(anonymous namespace)::spin_kernel') or n.startswith('Cijk_') or 'mfma_clock_probe_kernel' in n
work = [r for r in rows if not calib(r["Name"])]
tot = sum(float(r["TotalDurationNs"]) for r in work)
nms = [int(r["Calls"]) for r in rows if r["Name"].startswith("nms_kernel")][0]
conv = sum(float(r["TotalDurationNs"]) for r in work if 'conv_' in r["Name"] or 'bottleneck' in r["Name"] or 'stem_pool' in r["Name"] or 'unpack_wgrad' in r["Name"])
print("scoring passes", nms, "| kernel ms per pass", round(tot / nms / 1e6, 3), "| conv-class (igemm, wgrad, fused blocks, stem, split-K finalize, unpack)",
      round(conv / nms / 1e6, 3), "| everything else", round((tot - conv) / nms / 1e6, 3))
for r in work[:36]:
    print("%-72s %7.1f calls %8.3f ms %6.2f%%" % (r["Name"][:72], int(r["Calls"]) / nms, float(r["TotalDurationNs"]) / nms / 1e6, 100.0 * float(r["TotalDurationNs"]) / tot))
for r in rows:
    if calib(r["Name"]):
        print("(calibration, not counted) %-46s %7.1f calls %8.3f ms" % (r["Name"][:46], int(r["Calls"]) / nms, float(r["TotalDurationNs"]) / nms / 1e6))
