"""Per-kernel durations from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
python tools/kernel_times.py results.db [substring]  ->  name, grid, launches, average / minimum duration."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ''
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, d.grid_size_x, d.grid_size_y, count(*), avg(d.end - d.start), min(d.end - d.start) "
     f"from {kd} d join {ks} s on d.kernel_id = s.id group by 1, 2, 3 order by 1, 2, 3")
for name, gx, gy, n, avg, mn in db.execute(q):
    if want in name:
        print(f'{name[:60]:60s} grid {gx:>8} x {gy:<3} launches {n:5d}  avg {avg / 1e3:8.1f} us  min {mn / 1e3:8.1f} us')

# the timeline between dispatches (same queue): gap = start of a dispatch - end of the one before it
rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    key = (n0[:28], n1[:28])
    gaps.setdefault(key, []).append(s1 - e0)
print('gaps (previous kernel -> next kernel): launches, median us')
for key, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
    v = sorted(v)
    print(f'  {key[0]:28s} -> {key[1]:28s} {len(v):6d}  {v[len(v) // 2] / 1e3:8.2f}')
