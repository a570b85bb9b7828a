"""Summarise a rocprofv3 results .db (sqlite, rocpd schema) as a per-kernel stats table (like --stats CSV output).
usage: python tools/rocprof_db_stats.py results.db [out.txt]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)
        a = agg.setdefault(short, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k[:70]:70s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / total:6.2f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
