import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nms = [int(r["Calls"]) for r in rows if r["Name"].startswith("nms_kernel")][0]
print("scoring passes", nms, "total kernel ms per pass", round(tot / nms / 1e6, 3))
for r in rows[:34]:
    print("%-72s %7.1f calls %8.3f ms %6.2f%%" % (r["Name"][:72], int(r["Calls"]) / nms, float(r["TotalDurationNs"]) / nms / 1e6, float(r["Percentage"])))
