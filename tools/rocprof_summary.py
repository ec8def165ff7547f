"""Condense rocprofv3 (ROCm 7.2, rocpd sqlite output) results into small text summaries for profiles/.

    python tools/rocprof_summary.py stats  <results.db> > profiles/rNN_kernel_stats.csv
    python tools/rocprof_summary.py pmc    <results.db> [name-substring] > profiles/rNN_pmc_<counter>.csv
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    base = name.split("(")[0]
    if base.startswith("at::native"):
        return "torch:" + (re.findall(r"([a-z_0-9]+_kernel[a-z_0-9]*)", name) or [base])[0][:60]
    return base[:80]


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("kernel,calls,total_ms,avg_us,min_us,max_us,percent")
    for n, c, s, a, mn, mx in rows:
        print(f"{short(n)},{c},{s / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * s / tot:.2f}")


def pmc(db, sub=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), min(value), max(value) from counters_collection group by kernel_name, counter_name order by 4 desc").fetchall()
    print("kernel,counter,dispatches,sum,avg_per_dispatch,min,max")
    for n, cn, c, s, a, mn, mx in rows:
        if sub and sub not in n:
            continue
        print(f"{short(n)},{cn},{c},{s:.1f},{a:.1f},{mn:.1f},{mx:.1f}")


def pmc_each(db, sub=None):
    """per-dispatch counter values (dispatch order), for runs whose launches of one kernel differ in shape"""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, kernel_name, counter_name order by dispatch_id").fetchall()
    print("dispatch,kernel,counter,value")
    for d, n, cn, v in rows:
        if sub and sub not in n:
            continue
        print(f"{d},{short(n)},{cn},{v:.1f}")


def each(db, sub=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, duration from kernels order by start").fetchall()
    print("kernel,duration_us")
    for n, dur in rows:
        if sub and sub not in n:
            continue
        print(f"{short(n)},{dur / 1e3:.2f}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "pmc_each": pmc_each, "each": each}[sys.argv[1]](*sys.argv[2:])
