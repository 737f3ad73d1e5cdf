#!/usr/bin/env python3
"""per-step kernel table from a rocprofv3 results .db of tools/train_steps.py: only the last N steps
(a step ends with the last adam_multi_kernel launch).  usage: db_steps.py results.db [steps] [launches_per_step_of_adam]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
per = int(sys.argv[3]) if len(sys.argv) > 3 else 2
c = db.cursor()
ad = [r for r in c.execute("select start, end from kernels where name like '%adam_multi_kernel%' order by start")]
t1 = ad[-1][1]
t0 = ad[-1 - steps * per][1]
rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels where start > ? and end <= ? group by name order by 3 desc", (t0, t1)))
tot = 0.0
for n, cnt, ms, avg in rows:
    tot += ms
    if ms / steps >= 0.003:
        print(f"{n[:86]:86s} calls/step {cnt / steps:6.1f} ms/step {ms / steps:7.3f} avg_us {avg:8.1f}")
print(f"kernel time {tot / steps:.3f} ms/step; wall {(t1 - t0) / 1e6 / steps:.3f} ms/step")
