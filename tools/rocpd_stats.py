"""Kernel statistics (calls, total, average, min, max in ns) from a rocprofv3 rocpd database, as CSV on stdout.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/name_kernel_stats.csv
"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    dur = "duration" if "duration" in cols else '("end" - start)'
    rows = cur.execute(f"select {name}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels "
                       f"group by {name} order by sum({dur}) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, t, a, mn, mx in rows:
        n = re.sub(r"\s+", " ", n)
        print(f'"{n}",{c},{t},{a:.1f},{100.0 * t / total:.2f},{mn},{mx}')


if __name__ == "__main__":
    main(sys.argv[1])


def pmc(path, pattern="awq_gemm"):
    """Average counter value per (kernel, grid) for kernels whose name contains `pattern`, from a --pmc run."""
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), min(value), max(value) "
                      "from counters_collection where kernel_name like ? group by kernel_name, grid_size, counter_name",
                      (f"%{pattern}%",)).fetchall()
    out = []
    for n, g, c, cnt, a, mn, mx in rows:
        short = re.sub(r"\(.*", "", re.sub(r"\s+", " ", n))
        out.append(dict(kernel=short, grid=g, counter=c, launches=cnt, avg=a, min=mn, max=mx))
    return out
