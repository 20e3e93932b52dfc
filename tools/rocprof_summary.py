"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats [--pmc ...]) into a small text file for profiles/."""
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    lines = []
    lines.append("# per-kernel stats (rocprofv3 --kernel-trace --stats): name, calls, total_ns, average_ns, percent")
    for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        lines.append("%s | calls=%d | total_ns=%.0f | avg_ns=%.0f | pct=%.3f" % r)
    lines.append("")
    lines.append("# per-dispatch: kernel, grid, workgroup, duration_ns, vgpr, accum_vgpr, sgpr, lds, scratch")
    for r in c.execute("select name, grid_x, workgroup_x, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, "
                       "scratch_size from kernels order by start"):
        lines.append("%s | grid=%d | wg=%d | dur_ns=%d | vgpr=%d | agpr=%d | sgpr=%d | lds=%d | scratch=%d" % r)
    try:
        rows = list(c.execute("select * from counters_collection"))
        if rows:
            cur = c.execute("select * from counters_collection limit 1")
            cols = [d[0] for d in cur.description]
            lines.append("")
            lines.append("# counters_collection columns: " + ", ".join(cols))
            ki, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else None, None, None
            for name in ("counter_name", "name"):
                if name in cols:
                    ci = cols.index(name)
            for name in ("value", "counter_value"):
                if name in cols:
                    vi = cols.index(name)
            for r in rows:
                lines.append(" | ".join(str(x) for x in r))
    except sqlite3.Error as e:
        lines.append("# no counters: %s" % e)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
