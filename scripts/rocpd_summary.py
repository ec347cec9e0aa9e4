#!/usr/bin/env python3
"""Condenses rocprofv3 rocpd SQLite output (bench_results.db) into the text summaries committed under profiles/.

  python scripts/rocpd_summary.py kernels <db>            -> per-kernel calls / total / avg / min / max (us), like --stats
  python scripts/rocpd_summary.py pmc <db> [<db> ...]     -> per-kernel counter averages per dispatch (FETCH_SIZE / WRITE_SIZE are KiB)
  python scripts/rocpd_summary.py clocks <db>             -> GRBM_GUI_ACTIVE / duration per kernel (a --pmc GRBM_GUI_ACTIVE --kernel-trace pass): effective GHz
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"radix_sort_onesweep_(\w+)", name)
    if "rocprim" in name and m:
        return "rocprim::radix_sort_onesweep_" + m.group(1)
    return name[-70:]


def kernels(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                       "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}  vgpr agpr sgpr   lds scratch grid wg")
    agg = {}
    for r in rows:
        k = short(r[0])
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0, r[6], r[7], r[8], r[9], r[10], r[11], r[12]])
        a[0] += r[1]; a[1] += r[2]; a[2] = min(a[2], r[4]); a[3] = max(a[3], r[5])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:72s} {a[0]:6d} {a[1]/1e3:11.1f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:10.2f} {a[3]/1e3:10.2f} {100*a[1]/total:6.2f}  "
              f"{a[4]:4d} {a[5]:4d} {a[6]:4d} {a[7]:5d} {a[8]:7d} {a[9]} {a[10]}")


def pmc(dbs):
    print(f"{'kernel':72s} {'counter':>14s} {'dispatches':>10s} {'avg/dispatch':>16s}")
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                           "group by kernel_name, counter_name order by sum(value) desc").fetchall()
        agg = {}
        for r in rows:
            a = agg.setdefault((short(r[0]), r[1]), [0, 0.0])
            a[1] = (a[1] * a[0] + r[3] * r[2]) / (a[0] + r[2])
            a[0] += r[2]
        for (k, c), a in sorted(agg.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
            print(f"{k:72s} {c:>14s} {a[0]:10d} {a[1]:16.1f}")


def clocks(db):
    """Busy cycles of the dispatch (GRBM_GUI_ACTIVE) over its duration from the same pass's kernel trace: the clock the kernel ran at under
    the profiler.  The join key is the dispatch id; the schema differs between rocprofv3 releases, so the columns are discovered."""
    con = sqlite3.connect(db)
    cur = con.cursor()
    def cols(view):
        try:
            return [r[1] for r in cur.execute(f"pragma table_info({view})").fetchall()]
        except sqlite3.Error:
            return []
    cc, kc = cols("counters_collection"), cols("kernels")
    key = next((k for k in ("dispatch_id", "id", "correlation_id") if k in cc and k in kc), None)
    if "start" in cc and "end" in cc:  # the counter view carries the dispatch's own timestamps
        rows = cur.execute("select kernel_name, value, (end - start) from counters_collection where counter_name = 'GRBM_GUI_ACTIVE'").fetchall()
    elif key:
        rows = cur.execute(f"select c.kernel_name, c.value, k.duration from counters_collection c join kernels k on c.{key} = k.{key} "
                           "where c.counter_name = 'GRBM_GUI_ACTIVE'").fetchall()
    else:
        print("cannot join counters with durations; counters_collection:", cc, "kernels:", kc)
        return
    agg = {}
    for name, cycles, dur in rows:
        if not dur:
            continue
        a = agg.setdefault(short(name), [0, 0.0, 0.0])
        a[0] += 1; a[1] += cycles; a[2] += dur
    print(f"{'kernel':72s} {'dispatches':>10s} {'avg_cycles':>14s} {'avg_us':>10s} {'GHz':>7s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print(f"{k:72s} {a[0]:10d} {a[1]/a[0]:14.0f} {a[2]/a[0]/1e3:10.2f} {a[1]/a[2]:7.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "kernels":
        kernels(sys.argv[2])
    elif sys.argv[1] == "clocks":
        clocks(sys.argv[2])
    else:
        pmc(sys.argv[2:])
