#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (SQLite) result: per-kernel count / total / average duration and, if the run had
--pmc counters, the per-kernel average counter value per dispatch.  Usage: summarize_rocpd.py results.db [more.db]"""
import re
import sqlite3
import sys


def tables(con):
    return {r[0] for r in con.execute("select name from sqlite_master where type='table'")}


def find(tabs, stem):
    return next(t for t in tabs if t.startswith(stem))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone .*\]", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main(path):
    con = sqlite3.connect(path)
    tabs = tables(con)
    kd, ks = find(tabs, "rocpd_kernel_dispatch"), find(tabs, "rocpd_info_kernel_symbol")
    cols = [r[1] for r in con.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    rows = con.execute(f"select s.{namecol}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# {path}")
    print(f"{'kernel':110s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for n, c, t, mn, mx in rows:
        print(f"{short(n):110s} {c:6d} {t / 1e6:10.3f} {t / c / 1e3:10.1f} {mn / 1e3:10.1f} {mx / 1e3:10.1f} {100.0 * t / tot:6.2f}")
    pe = find(tabs, "rocpd_pmc_event")
    if con.execute(f"select count(*) from {pe}").fetchone()[0] > 0:
        ip = find(tabs, "rocpd_info_pmc")
        pcols = [r[1] for r in con.execute(f"pragma table_info({pe})")]
        evcol = "event_id" if "event_id" in pcols else pcols[1]
        q = (f"select s.{namecol}, p.name, count(*), sum(e.value) from {pe} e join {ip} p on e.pmc_id = p.id "
             f"join {kd} d on d.event_id = e.{evcol} join {ks} s on d.kernel_id = s.id group by s.{namecol}, p.name order by 4 desc")
        print("\n# counters: per-kernel average per dispatch")
        for n, cn, c, v in con.execute(q).fetchall():
            print(f"{short(n):110s} {cn:14s} dispatches={c:5d} avg={v / c:16.1f}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
        print()
