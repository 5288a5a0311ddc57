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
    families(db)


# kernel families of the roofline report (substring match on the mangled kernel name)
FAMILIES = [
    ("3x3 convolution (fwd + dgrad + wgrad, incl. K-split tails and split reduces)",
     ["conv_kernel", "conv1_", "conv_tail_reduce", "wgrad_reduce_kernel", "conv_wino_kernel", "conv_wino2_kernel", "conv_wino4_kernel",
      "wino_wgrad_kernel", "wino_wgrad_reduce"]),
    ("GEMM (vc::gemm_kernel / vc::gemm_bx_kernel + split-K reduce)", ["gemm_kernel", "gemm_bx_kernel", "splitk_reduce"]),
    ("LSTM recurrence (step / gate kernels)", ["lstm_"]),
    ("softmax cross-entropy", ["xent_"]),
    ("optimiser (Adam / SGD / Momentum)", ["adam_kernel", "sgd_kernel", "momentum_kernel"]),
]


def families(db):
    """Per kernel family: dispatches, SUM of dispatch durations (the serial time) and UNION of the dispatch intervals (the time
    at least one kernel of the family was running; smaller than the sum when launches overlap on several streams).
    bench.py's roofline.frac = family FLOPs / union; frac_serial = family FLOPs / sum -- steps traced = dispatches of
    preprocess_kernel (one per fine-tune step) or of xent kernels (one per caption step)."""
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    steps = sum(1 for r in rows if "preprocess_kernel" in r[0]) or sum(1 for r in rows if "xent_" in r[0])
    print("\n| kernel family | dispatches | sum of durations ms | union of intervals ms | per step (of %d traced) sum / union ms |" % steps)
    print("|---|---|---|---|---|")
    for title, pats in FAMILIES:
        iv = [(a, b) for n, a, b in rows if any(p in n for p in pats)]
        if not iv:
            continue
        tot, uni, end = sum(b - a for a, b in iv), 0, -1
        for a, b in iv:  # already sorted by start
            if b <= end:
                continue
            uni += b - max(a, end)
            end = b
        print("| %s | %d | %.3f | %.3f | %.3f / %.3f |" % (title, len(iv), tot / 1e6, uni / 1e6, tot / 1e6 / max(steps, 1), uni / 1e6 / max(steps, 1)))


def gaps(db, last=30, top=25):
    """Where the step's time OUTSIDE the convolution family goes: over the last `last` steps (window = start of one
    preprocess_kernel to the start of the one `last` steps later, so exactly `last` steps), every elementary interval in which
    no convolution dispatch is running is charged to the kernels that ARE running (split evenly) or to `idle` (nothing on the
    device: launch latency, host enqueue, event waits)."""
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    marks = [a for n, a, b in rows if "preprocess_kernel" in n]
    if len(marks) < last + 1:
        return
    lo, hi = marks[-last - 1], marks[-1]
    conv = FAMILIES[0][1]
    ev = []
    for n, a, b in rows:
        a, b = max(a, lo), min(b, hi)
        if b > a:
            ev.append((a, 1, n))
            ev.append((b, -1, n))
    ev.sort(key=lambda e: (e[0], e[1]))
    active, charge, prev, nconv, covered = {}, {}, lo, 0, 0
    for t, sgn, n in ev:
        dt = t - prev
        if dt > 0:
            if nconv:
                covered += dt
            elif active:
                for k in active:
                    charge[k] = charge.get(k, 0) + dt / len(active)
            else:
                charge["idle (no kernel on the device)"] = charge.get("idle (no kernel on the device)", 0) + dt
        prev = t
        isconv = any(p in n for p in conv)
        if sgn > 0:
            active[n] = active.get(n, 0) + 1 if not isconv else active.get(n, 0)
            nconv += isconv
        else:
            if isconv:
                nconv -= 1
            else:
                active[n] -= 1
        active = {k: v for k, v in active.items() if v > 0}
    if hi > prev:
        charge["idle (no kernel on the device)"] = charge.get("idle (no kernel on the device)", 0) + hi - prev
    span = hi - lo
    print("\nlast %d steps: %.3f ms per step, %.3f ms with a convolution dispatch running, %.3f ms without; the time without, by "
          "what was running instead (ms per step):\n" % (last, span / 1e6 / last, covered / 1e6 / last, (span - covered) / 1e6 / last))
    print("| running while no convolution is | ms per step |")
    print("|---|---|")
    for k, v in sorted(charge.items(), key=lambda kv: -kv[1])[:top]:
        print("| `%s` | %.3f |" % (k if len(k) <= 110 else k[:107] + "...", v / 1e6 / last))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    gaps(sqlite3.connect(sys.argv[1]))
