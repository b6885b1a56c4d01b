"""Timeline of one train step from a rocprofv3 kernel trace CSV: where no large kernel is running (not part of the product).
usage: trace_window.py o_kernel_trace.csv [list]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wgs = 1
    for ax in "XYZ":
        wgs *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
    K.append((s, e, r["Kernel_Name"], wgs))
K.sort()
ad = [i for i, k in enumerate(K) if k[2].startswith("adamw_kernel")]
step = K[ad[-4] + 1: ad[-3] + 1]
t0 = step[0][0]
print(f"step span {(step[-1][1] - t0) / 1e3:.1f} us, {len(step)} kernels")
ev = sorted([(s, 1, w) for s, e, n, w in step] + [(e, -1, w) for s, e, n, w in step])
big = anyk = 0
last = t0
tb = ts = ti = 0
for t, d, w in ev:
    dt = t - last
    if big > 0:
        tb += dt
    elif anyk > 0:
        ts += dt
    else:
        ti += dt
    last = t
    if w >= 256:
        big += d
    anyk += d
print(f"a kernel of >= 256 workgroups running: {tb / 1e3:.1f} us; only smaller kernels: {ts / 1e3:.1f} us; nothing: {ti / 1e3:.1f} us")
# the fusion / answer-head chain: from the end of the CNN forward's last kernel (the stage-4 spatial-attention scale pass in front of
# the loss) to the start of the CNN backward's first kernel (the stage-4 spatial-attention reduce): launches on the critical path with
# nothing large beside them.  Main-stream kernels only would need stream ids; the count is of ALL kernels in the window.
ce = [i for i, k in enumerate(step) if "cross_entropy" in k[2]]
if ce:
    fwd_end = max((k[1] for k in step[:ce[0]] if "scale_kernel" in k[2]), default=None)
    bwd_start = min((k[0] for k in step[ce[0]:] if "spatial_bwd_reduce_kernel" in k[2]), default=None)
    if fwd_end and bwd_start:
        inside = [k for k in step if k[0] >= fwd_end and k[0] < bwd_start]
        busy = sum(k[1] - k[0] for k in inside)
        print(f"fusion / answer-head chain window: {(bwd_start - fwd_end) / 1e3:.1f} us elapsed, {len(inside)} launches in it, {busy / 1e3:.1f} us of kernel time")
        if len(sys.argv) > 2 and sys.argv[2] == "chain":
            for s_, e_, n_, w_ in inside:
                print(f"{(s_ - fwd_end) / 1e3:9.1f} {(e_ - s_) / 1e3:7.1f} wg={w_:6d} {n_[:90]}")
            sys.exit(0)
if len(sys.argv) > 2:
    lo, hi = float(sys.argv[3]) * 1e3, float(sys.argv[4]) * 1e3
    agg = collections.defaultdict(lambda: [0, 0])
    prev_end = None
    for s, e, n, w in step:
        if lo <= s - t0 <= hi:
            agg[n[:60]][0] += 1; agg[n[:60]][1] += e - s
            if sys.argv[2] == "list":
                print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} wg={w:6d} {n[:70]}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{n:60s} {c:4d} {t / 1e3:9.1f} us")
