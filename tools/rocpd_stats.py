"""Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel stats table (like --stats CSV)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def by_grid(path, pattern, top=40):
    """the launches of kernels whose name contains `pattern`, split by grid size (e.g. fc1 vs fc2 launches of the grouped GEMM)"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    gcols = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols]
    if not gcols:
        print("(no grid column in this rocpd schema: " + ", ".join(cols) + ")")
        return
    g = gcols[0]
    rows = cur.execute(f"select {name_col}, {g}, count(*), avg(end-start), min(end-start), max(end-start) from kernels where {name_col} like ? "
                       f"group by {name_col}, {g} order by count(*) * avg(end-start) desc", (f"%{pattern}%",)).fetchall()
    print(f"-- launches of *{pattern}* by {g}")
    for n, gx, c, a, mn, mx in rows[:top]:
        print(f"{short(n)[:70]:70s} {g}={gx:<10} calls={c:<6d} avg_us={a/1e3:10.2f} min_us={mn/1e3:9.2f} max_us={mx/1e3:9.2f}")


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                       f"group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, c, s, a, mn, mx in rows[:top]:
        print(f"{short(n):110s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/total:6.2f}")
    print(f"TOTAL kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    for pat in sys.argv[3:]:
        by_grid(sys.argv[1], pat)
