#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd
SQLite database -- the same table `rocprofv3 --stats` prints, for runs whose output format is
the ROCm 7.2 default (`*_results.db`).   python tools/rocpd_stats.py results.db [> summary.md]"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), "
        "max(d.workgroup_size_x), min(d.grid_size_x * d.grid_size_y * d.grid_size_z), max(d.grid_size_x * d.grid_size_y * d.grid_size_z) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | wg | grid min..max |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        name = r[0]
        if len(name) > 110:
            name = name[:107] + "..."
        print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f | %d | %d | %d | %d | %d | %d..%d |" % (
            name, r[1], r[2] / 1e6, r[2] / r[1] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[2] / total, r[5], r[6], r[7], r[8], r[9], r[10], r[11]))
    print("\ntotal kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
