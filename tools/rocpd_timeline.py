#!/usr/bin/env python
"""One training step's dispatches in launch order from a rocprofv3 rocpd database:
   python tools/rocpd_timeline.py results.db [marker_kernel_substring] [min_us]
The step is delimited by two consecutive launches of the marker kernel (default: preprocess_kernel,
the first kernel of a fine-tune step); the last complete step of the trace is printed."""
import sqlite3
import sys


def main(path, marker="preprocess_kernel", min_us=20.0):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select s.kernel_name, d.start, d.end, d.grid_size_x * d.grid_size_y * d.grid_size_z, d.workgroup_size_x, d.queue_id "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    lo, hi = marks[-2], marks[-1]
    t0 = rows[lo][1]
    busy = 0
    for name, s, e, grid, wg, q in rows[lo:hi]:
        busy += e - s
        if (e - s) / 1e3 >= min_us:
            short = name.replace("_ZN2vc", "").replace("NS_7TileCfgI", "<").replace("EvNS_8ConvArgsE.kd", "").replace("EvNS_8GemmArgsE.kd", "")
            print("%9.3f ms  +%8.1f us  q%-3d wgs %6d  %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, grid // max(wg, 1), short[:90]))
    print("step span %.3f ms, sum of kernel durations %.3f ms, %d dispatches" % ((rows[hi][1] - t0) / 1e6, busy / 1e6, hi - lo))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "preprocess_kernel", float(sys.argv[3]) if len(sys.argv) > 3 else 20.0)
