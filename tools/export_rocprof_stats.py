"""Export the per-kernel statistics (and, for SQ passes, per-kernel counter means) of a rocprofv3 run (rocpd SQLite database) as CSV text for profiles/.
usage: python tools/export_rocprof_stats.py <results.db> <out.csv> [counters]"""
import csv
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:80]


def main():
    db = sqlite3.connect(sys.argv[1])
    if len(sys.argv) > 3 and sys.argv[3] == "counters":
        rows = db.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from (select kernel_name, counter_name, dispatch_id, sum(value) as value "
                          "from counters_collection group by dispatch_id, counter_name) group by kernel_name, counter_name").fetchall()
        with open(sys.argv[2], "w", newline="") as fh:
            w = csv.writer(fh); w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "total"])
            for k, c, n, a, t in sorted(rows, key=lambda r: (-r[4] if r[1] == "SQ_WAVE_CYCLES" else 0, r[0], r[1])):
                if any(x in k for x in ("k_", "cpe")):
                    w.writerow([short(k), c, n, a, t])
        return
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(sys.argv[2], "w", newline="") as fh:
        w = csv.writer(fh); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([short(n), c, int(t), round(a, 1), int(mn), int(mx), round(100.0 * t / tot, 3)])


if __name__ == "__main__":
    main()
