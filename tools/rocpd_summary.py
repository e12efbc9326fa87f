#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into small text files for profiles/.

usage: rocpd_summary.py stats <results.db>          -> per-kernel calls / total / mean (us)
       rocpd_summary.py pmc   <results.db> [...]    -> per-kernel mean counter values
       rocpd_summary.py bygrid <results.db> <substr> -> the kernels whose name holds <substr>, one row per launch shape
"""
import sqlite3
import sys


def short(name):
    n = name.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0][-60:] or "void"


def stats(db):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration),"
                          " avg(vgpr_count), avg(sgpr_count), avg(lds_size) from kernels group by name"
                          " order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    print("%-46s %6s %12s %10s %10s %10s %6s %5s %5s %7s" % ("kernel", "calls", "total_us", "mean_us",
                                                         "min_us", "max_us", "pct", "vgpr", "sgpr", "lds"))
    for r in rows:
        print("%-46s %6d %12.1f %10.1f %10.1f %10.1f %6.2f %5d %5d %7d" % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            r[6], r[7], r[8]))


def bygrid(db, sub):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    g = [k for k in ("grid_x", "grid_y", "grid_z", "workgroup_x") if k in cols] or [k for k in cols if k.startswith("grid")]
    rows = list(c.execute("select name, %s, count(*), sum(duration), avg(duration), min(duration) from kernels where name like ?"
                          " group by name, %s order by avg(duration) desc" % (", ".join(g), ", ".join(g)), ("%" + sub + "%",)))
    print("%-40s %-28s %6s %12s %10s %10s" % ("kernel", "/".join(g), "calls", "total_us", "mean_us", "min_us"))
    for r in rows:
        k = len(g)
        print("%-40s %-28s %6d %12.1f %10.1f %10.1f" % (short(r[0])[-40:], "/".join(str(int(x)) for x in r[1:1 + k]), r[1 + k],
                                                        r[2 + k] / 1e3, r[3 + k] / 1e3, r[4 + k] / 1e3))


def pmc(dbs):
    acc = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for name, counter, val, n in c.execute(
                "select kernel_name, counter_name, avg(value), count(*) from counters_collection"
                " group by kernel_name, counter_name"):
            acc.setdefault(short(name), {})[counter] = (val, n)
    counters = sorted({k for v in acc.values() for k in v})
    print("%-46s %6s " % ("kernel", "calls") + " ".join("%16s" % k for k in counters))
    for name, v in sorted(acc.items(), key=lambda kv: -max(x[0] for x in kv[1].values())):
        n = max(x[1] for x in v.values())
        print("%-46s %6d " % (name, n) + " ".join("%16.2f" % v.get(k, (float('nan'), 0))[0] for k in counters))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "bygrid":
        bygrid(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:])
