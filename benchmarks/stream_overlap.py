#!/usr/bin/env python3
"""Do the kernels of the text encoder's side stream run BESIDE the main stream's?  Reads a rocprofv3 --kernel-trace CSV of bench.py and
reports, per hardware queue, the busy time and how much of the side queue's busy time another queue was busy too.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -o t -- python3 bench.py --config c1 --also "" --steps 3 --warmup 2 --no-cpu-baseline
    python3 benchmarks/stream_overlap.py /tmp/ov
"""
import collections
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    qs = sorted(byq, key=lambda q: -len(byq[q]))
    main_q = qs[0]
    print("queues:", {q: len(byq[q]) for q in qs})
    mains = sorted((a, b) for a, b, _ in byq[main_q])
    import bisect
    starts = [a for a, _ in mains]
    for q in qs[1:]:
        busy = over = 0
        names = collections.Counter()
        for a, b, n in byq[q]:
            busy += b - a
            names[n.split("(")[0][-50:]] += b - a
            i = max(0, bisect.bisect_left(starts, a) - 8)
            while i < len(mains) and mains[i][0] < b:
                lo, hi = max(a, mains[i][0]), min(b, mains[i][1])
                if hi > lo:
                    over += hi - lo
                i += 1
        print("queue %s: %d kernels, busy %.3f ms, of which %.3f ms beside a main-queue kernel (%.0f %%)" % (q, len(byq[q]), busy / 1e6, over / 1e6, 100.0 * over / max(busy, 1)))
        for n, t in names.most_common(6):
            print("      %8.3f ms  %s" % (t / 1e6, n))


if __name__ == "__main__":
    main()
