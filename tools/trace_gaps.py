"""Busy fraction and idle-gap histogram of the GPU from a rocprofv3 --kernel-trace CSV (diagnostic only).
usage: python tools/trace_gaps.py <dir-with-*_kernel_trace.csv> [lo_frac hi_frac]"""
import csv, glob, sys
d = sys.argv[1]
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 0.9)
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")))
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
win = [e for e in ev if e[0] >= a and e[1] <= b]
span = win[-1][1] - win[0][0]
busy, cur_s, cur_e = 0, win[0][0], win[0][1]
gaps = []
for s, e, n, q in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(win)}  span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms = {busy/span*100:.1f}%  sum(durations) {sum(e-s for s,e,_,_ in win)/1e6:.2f} ms")
import collections
h = collections.Counter()
tot = collections.Counter()
for g, n in gaps:
    k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
    h[k] += 1; tot[k] += g
for k in ["<2us", "<5us", "<10us", "<50us", ">=50us"]:
    print(f"  gaps {k:6s}: {h[k]:6d}  total {tot[k]/1e6:7.3f} ms")
by = collections.Counter()
for g, n in gaps:
    by[n] += g
print("  idle time before kernel (top 12):")
for n, g in by.most_common(12):
    print(f"    {g/1e6:7.3f} ms  {n}")
