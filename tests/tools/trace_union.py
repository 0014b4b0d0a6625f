#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 --kernel-trace CSV of `bench.py`, the per-kernel statistics and the UNION of the
kernel intervals (time during which at least one kernel ran) per timed step -- the cross-check of `ms_per_step`
when kernels of neighbouring launches overlap on several streams.

    python tests/tools/trace_union.py <kernel_trace.csv> <steps> <warmup> [out.txt]

The timed region is taken as the `steps` computations after the first `warmup` ones, a computation being
delimited by its k_reset launch."""
import csv
import sys


def main():
    path, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = open(sys.argv[4], "w") if len(sys.argv) > 4 else sys.stdout
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    resets = [i for i, r in enumerate(rows) if "k_reset" in r[2]]
    if len(resets) < warmup + steps + 1:
        print("only %d k_reset launches in the trace" % len(resets), file=out)
        return
    lo, hi = resets[warmup], resets[warmup + steps]
    timed = rows[lo:hi]
    t0, t1 = timed[0][0], max(e for _, e, _ in timed)
    union = 0
    cur_s, cur_e = timed[0][0], timed[0][1]
    for s, e, _ in timed[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    per = {}
    for s, e, k in timed:
        a = per.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    print("timed region: %d steps, %.4f ms per step wall (first kernel start to last kernel end), union of kernel "
          "intervals %.4f ms per step (%.1f %% of the wall), sum of kernel durations %.4f ms per step"
          % (steps, (t1 - t0) / steps / 1e6, union / steps / 1e6, 100.0 * union / (t1 - t0),
             sum(v[1] for v in per.values()) / steps / 1e6), file=out)
    print("%-60s %8s %12s %12s" % ("kernel", "calls", "avg us", "ms per step"), file=out)
    for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-60s %8d %12.2f %12.4f" % (k[:60], n, d / n / 1e3, d / steps / 1e6), file=out)


if __name__ == "__main__":
    main()
