#!/usr/bin/env python3
"""Where is the GPU idle inside a training iteration?  From a rocprofv3 --kernel-trace CSV of bench.py: the union of all kernels' busy
intervals (all queues), the idle time between them, and the largest gaps with the kernels on either side.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -o t -- python3 bench.py --config c1 --also "" --steps 3 --warmup 2 --no-cpu-baseline
    python3 benchmarks/gpu_idle_gaps.py /tmp/ov [iterations_in_trace]
"""
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # keep the last `iters - 2` iterations' worth: skip the first 40 % of the kernels (warm-up: allocations, first-use layouts)
    rows = rows[int(len(rows) * 0.4):]
    busy = 0
    gaps = []
    cur_end = rows[0][1]
    cur_start = rows[0][0]
    last_name = rows[0][2]
    for a, b, n, q in rows[1:]:
        if a > cur_end:
            gaps.append((a - cur_end, cur_end, last_name, n))
            busy += cur_end - cur_start
            cur_start = a
        if b > cur_end:
            cur_end = b
            last_name = n
    busy += cur_end - cur_start
    span = rows[-1][1] - rows[0][0]
    idle = sum(g[0] for g in gaps)
    print("span %.2f ms, busy %.2f ms, idle %.2f ms (%.1f %%) in %d gaps" % (span / 1e6, busy / 1e6, idle / 1e6, 100.0 * idle / span, len(gaps)))
    for lo, hi in ((0, 2_000), (2_000, 5_000), (5_000, 10_000), (10_000, 20_000), (20_000, 100_000), (100_000, 10 ** 12)):
        g = [x[0] for x in gaps if lo <= x[0] < hi]
        print("  gaps of %6.1f .. %6.1f us: %5d, %.2f ms" % (lo / 1e3, min(hi, 10 ** 9) / 1e3, len(g), sum(g) / 1e6))
    print("largest gaps:")
    t0 = rows[0][0]
    for g, at, before, after in sorted(gaps, reverse=True)[:14]:
        print("  %7.1f us at %8.2f ms  after %-46s before %s" % (g / 1e3, (at - t0) / 1e6, before.split("(")[0][-46:], after.split("(")[0][-46:]))


if __name__ == "__main__":
    main()
