#!/usr/bin/env python
"""Per-range summary of the roctx ranges in a rocprofv3 --marker-trace database (trainer.Trainer emits them with VC_TRACE=1):
   python tools/marker_stats.py results.db
The ranges bracket the HOST-side enqueue of a phase (the step is asynchronous: a range is short when the queue has room and long
when the host has to wait for the device); they label the kernel timeline of the same trace."""
import collections
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
acc = collections.OrderedDict()
for dur, ext in db.execute("select duration, extdata from regions where category like 'MARKER%' order by start"):
    try:
        name = json.loads(ext).get("message", "?")
    except Exception:  # noqa: BLE001
        name = "?"
    a = acc.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
print("| roctx range | count | total ms | avg us | min us | max us |")
print("|---|---|---|---|---|---|")
for n, (c, t, lo, hi) in acc.items():
    print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f |" % (n, c, t / 1e6, t / c / 1e3, lo / 1e3, hi / 1e3))
