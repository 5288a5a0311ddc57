"""tools/rocpd_stats.py on a synthetic rocpd database: the numbers profiles/*_kernel_stats.md (and through them DESIGN.md section 6)
quote -- per-family sum / union of dispatch intervals, and the step's time outside the convolutions charged to what runs there --
must come out of a hand-built timeline exactly."""
import io
import os
import sqlite3
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _db(tmp_path, rows):
    path = str(tmp_path / "t.db")
    db = sqlite3.connect(path)
    db.execute("create table rocpd_info_kernel_symbol (id integer, kernel_name text, arch_vgpr_count int, accum_vgpr_count int, sgpr_count int)")
    db.execute("create table rocpd_kernel_dispatch (kernel_id int, start int, end int, group_segment_size int, workgroup_size_x int, "
               "grid_size_x int, grid_size_y int, grid_size_z int)")
    names = sorted({r[0] for r in rows})
    for i, n in enumerate(names):
        db.execute("insert into rocpd_info_kernel_symbol values (?, ?, 64, 0, 32)", (i, n))
    for n, a, b in rows:
        db.execute("insert into rocpd_kernel_dispatch values (?, ?, ?, 0, 256, 1024, 1, 1)", (names.index(n), a, b))
    db.commit()
    db.close()
    return path


def test_family_union_and_time_outside_the_convolutions(tmp_path):
    import rocpd_stats
    ms = 1000000
    rows = []
    for s in range(3):   # three "steps" of 10 ms each (nanoseconds): preprocess marks the step
        t = s * 10 * ms
        rows += [("vc::preprocess_kernel", t, t + ms // 10),
                 ("vc::conv_wino4_kernel<0>", t + 1 * ms, t + 4 * ms),          # two overlapping convolution dispatches: union 4 ms, sum 5 ms
                 ("vc::wino_wgrad_kernel<4,8>", t + 3 * ms, t + 5 * ms),
                 ("vc::gemm_kernel<x>", t + 4 * ms, t + 6 * ms),                # 1 ms under the weight gradient, 1 ms alone
                 ("vc::lstm_rec_fwd_kernel<5>", t + 6 * ms, t + 8 * ms),        # 2 ms, the second one shared with Adam
                 ("vc::adam_kernel<1>", t + 7 * ms, t + 8 * ms)]
    path = _db(tmp_path, rows)
    out = io.StringIO()
    with redirect_stdout(out):
        rocpd_stats.main(path, 10)
        rocpd_stats.gaps(sqlite3.connect(path), last=2)
    text = out.getvalue()
    fam = [l for l in text.splitlines() if l.startswith("| 3x3 convolution")][0]
    cells = [c.strip() for c in fam.split("|")]
    assert cells[2] == "6" and abs(float(cells[3]) - 15.0) < 1e-6 and abs(float(cells[4]) - 12.0) < 1e-6, fam   # sum 3 x 5, union 3 x 4 ms
    assert "5.000 / 4.000" in fam   # per step of 3 traced
    head = [l for l in text.splitlines() if l.startswith("last 2 steps")][0]
    assert "10.000 ms per step, 4.000 ms with a convolution dispatch running, 6.000 ms without" in head, head
    charge = {l.split("|")[1].strip().strip("`"): float(l.split("|")[2]) for l in text.splitlines() if l.startswith("| `")}
    assert abs(charge["vc::gemm_kernel<x>"] - 1.0) < 1e-6
    assert abs(charge["vc::lstm_rec_fwd_kernel<5>"] - 1.5) < 1e-6 and abs(charge["vc::adam_kernel<1>"] - 0.5) < 1e-6
    assert abs(charge["vc::preprocess_kernel"] - 0.1) < 1e-6
    assert abs(charge["idle (no kernel on the device)"] - 2.9) < 1e-6   # 0.9 ms before the first convolution + 2 ms after Adam
