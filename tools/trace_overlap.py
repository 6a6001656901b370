#!/usr/bin/env python3
"""How much do the large kernels of a run overlap?  From a rocprofv3 kernel trace (csv):

    rocprofv3 --kernel-trace -f csv -d out -- python3 tools/two_in_flight.py --reads 32768 --steps 6
    python tools/trace_overlap.py out

prints the sum of the durations of the codec's large kernels, the length of the union of their intervals, and the difference: time in
which two of them ran at once (on different hardware queues).  One context alone: 0."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
names = ("svb_encode_kernel", "zstd_plan_kernel", "zstd_pack_kernel", "fast_streams_kernel", "fast_runs_kernel", "svb_decode_kernel")
big = [r for r in rows if any(n in r[2] for n in names)]
# the last third of the trace is the two-context phase of the second round; split phases by queue usage: windows where two queues are active
t0, t1 = big[0][0], max(r[1] for r in big)
span = t1 - t0
busy = sum(r[1] - r[0] for r in big)
# union length
u = 0; cur_s, cur_e = None, None
for s, e, _, _ in big:
    if cur_e is None or s > cur_e:
        if cur_e is not None: u += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
u += cur_e - cur_s
print("big kernels: %d launches, sum of durations %.1f ms, union of their intervals %.1f ms, overlap %.1f ms (%.1f %% of the sum), queues %s"
      % (len(big), busy / 1e6, u / 1e6, (busy - u) / 1e6, 100.0 * (busy - u) / busy, sorted(set(r[3] for r in big))))
