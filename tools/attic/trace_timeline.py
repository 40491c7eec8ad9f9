"""Timeline of a rocprofv3 --kernel-trace (+ --memory-copy-trace) run from its results .db:
    python tools/attic/trace_timeline.py <results.db> [window_us=300] [offset_from_end_us=3000]
prints start (us), duration (us), stream and name of every dispatch / copy inside the window, then the per-stream busy
fraction and the fraction of the window in which >= 2 dispatches overlap."""
import sqlite3, sys
db = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
off = float(sys.argv[3]) if len(sys.argv) > 3 else 3000.0
cur = sqlite3.connect(db).cursor()
k = cur.execute("select name,start,end,stream_id from kernels order by start").fetchall()
try:
    m = cur.execute("select name,start,end,size,stream_id from memory_copies order by start").fetchall()
except Exception:
    m = []
t_end = k[-1][2]
w0 = t_end - off * 1000
w1 = w0 + win * 1000
ev = [(s, e, "K " + n.split("(")[0].replace("void ", "")[:48], st) for n, s, e, st in k if w0 <= s < w1]
ev += [(s, e, f"C {n} {sz}", st) for n, s, e, sz, st in m if w0 <= s < w1]
ev.sort()
for s, e, n, st in ev:
    print(f"{(s - w0) / 1000:8.1f} {(e - s) / 1000:7.1f} st{st} {n}")
# overlap statistics over the last `off` microseconds
evs = [(s, e) for n, s, e, st in k if s >= w0]
pts = sorted([(s, 1) for s, e in evs] + [(e, -1) for s, e in evs])
busy = over = 0
depth, last = 0, pts[0][0]
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
span = pts[-1][0] - pts[0][0]
print(f"span {span / 1000:.0f} us: >=1 kernel running {busy / span:.2f}, >=2 kernels running {over / span:.2f}, dispatches {len(evs)}")
