"""Timeline of the last forward passes in a rocprofv3 kernel trace (+ memory-copy trace): per pass the kernels' start offsets and
durations and the copies around them, microseconds.  python tools/trace_gaps.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void p3::", "")[:28]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Kind", ""))[:24]))
ev.sort()
blocks = [i for i, e in enumerate(ev) if e[2].startswith("k_block")]
import statistics
dur = [ (ev[i][1]-ev[i][0])/1e3 for i in blocks]
print("k_block launches", len(dur), "median us", statistics.median(dur), "last 20 mean", sum(dur[-20:])/20)
# gaps between consecutive k_block starts (cycle) for the last 20
st = [ev[i][0] for i in blocks]
cyc = [(st[i+1]-st[i])/1e3 for i in range(len(st)-1)]
print("cycle (k_block start to next start) last 20 mean us", sum(cyc[-20:])/20)
i0 = blocks[-3]
t0 = ev[i0][0]
for e in ev[max(0, i0-4): i0+12]:
    print(f"  {(e[0]-t0)/1e3:10.1f} .. {(e[1]-t0)/1e3:10.1f}  ({(e[1]-e[0])/1e3:8.1f})  {e[2]}")
