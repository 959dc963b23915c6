import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# find the last burst of queue-2 kernels
idx = [i for i, r in enumerate(rows) if r["Queue_Id"] == "2"]
# group bursts: gaps > 2 ms
bursts = []
cur = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 2_000_000:
        bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
print("bursts", [(len(b), round((int(rows[b[-1]]["End_Timestamp"]) - int(rows[b[0]]["Start_Timestamp"])) / 1e6, 3)) for b in bursts])
for b in bursts[-2:]:
    lo, hi = b[0] - 6, b[0] + 30
    print("---- burst of %d side kernels" % len(b))
    for r in rows[max(lo, 0):hi]:
        print("q%s %9.3f -> %9.3f ms  %s" % (r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r["Kernel_Name"][:70]))
    lo = b[-1] - 10
    print("   ... end of the burst")
    for r in rows[lo:b[-1] + 8]:
        print("q%s %9.3f -> %9.3f ms  %s" % (r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r["Kernel_Name"][:70]))
