#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a small markdown table.

    rocprofv3 --kernel-trace --stats -d OUT -o NAME -- <cmd>
    python tools/rocprof_summary.py OUT/NAME_results.db profiles/<file>.md "<cmd>"
"""
import sqlite3
import sys


def main(db_path, out_path, cmd=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    extra = {}
    try:
        for name, vg, sg, lds, gx, wx in cur.execute(
                "select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                "from kernels group by name"):
            extra[name] = (vg, sg, lds, gx, wx)
    except sqlite3.Error:
        pass
    # min / max per kernel
    mm = {}
    try:
        for name, mn, mx in cur.execute("select name, min(duration), max(duration) from kernels group by name"):
            mm[name] = (mn, mx)
    except sqlite3.Error:
        pass
    tot = sum(r[2] for r in rows)
    ncalls = sum(r[1] for r in rows)
    try:
        t0, t1 = list(cur.execute("select min(start), max(end) from kernels"))[0]
        span = (t1 - t0) / 1000.0
    except sqlite3.Error:
        span = float("nan")
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        f.write(f"total: {ncalls} dispatches, {tot:.0f} us of kernel time over a {span:.0f} us first-to-last span\n\n")
        f.write("durations in microseconds (rocpd `top_kernels` view; total_duration in us)\n\n")
        f.write("| kernel | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds B | grid | wg |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for name, calls, total, avg, pct in rows[:40]:
            short = name if len(name) < 110 else name[:107] + "..."
            vg, sg, lds, gx, wx = extra.get(name, ("", "", "", "", ""))
            mn, mx = mm.get(name, (None, None))
            mn = f"{mn / 1000:.2f}" if mn is not None else ""
            mx = f"{mx / 1000:.2f}" if mx is not None else ""
            f.write(f"| `{short}` | {calls} | {total:.1f} | {avg:.3f} | {mn} | {mx} | {pct:.2f} | {vg} | {sg} | {lds} | {gx} | {wx} |\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
