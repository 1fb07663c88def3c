"""Summarise rocprofv3 (rocpd sqlite) outputs into small text files for profiles/.
usage: prof_summary.py <kernel-trace.db> [<pmc.db> ...] > profiles/rNN_summary.txt"""
import re, sqlite3, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*?>)?)\(", name)
    return (m.group(1) if m else name)[:70]

def kernel_stats(db):
    con = sqlite3.connect(db); cur = con.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, c, s, a, mn, mx in rows[:14]:
        print(f"{short(n):72s} {c:6d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
    for pat in ("hnsw_search_kernel", "flat_scan_big_kernel"):
        d = [r[0] for r in cur.execute("select duration from kernels where name like ? order by start", (f"%{pat}%",))]
        if d:
            last = d[-20:]
            print(f"# {pat}: last {len(last)} launches (the timed steps) avg {sum(last)/len(last)/1e3:.2f} us; the last 30 of {len(d)}: "
                  + " ".join(f"{x/1e3:.0f}" for x in d[-30:]))

def pmc_stats(db):
    con = sqlite3.connect(db); cur = con.cursor()
    print(f"# PMC pass {db}")
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name order by sum(duration) desc").fetchall()
    for n, cn, c, v, d in rows[:10]:
        print(f"{short(n):72s} {cn:12s} launches {c:4d} avg_value {v:16.1f} avg_dur_us {d/1e3:10.2f}")
    for pat in ("hnsw_search_kernel",):
        r = cur.execute("select counter_name, value, duration from counters_collection where kernel_name like ? order by start", (f"%{pat}%",)).fetchall()
        if r:
            last = r[-5:]
            print(f"# {pat} last {len(last)} launches: " + "; ".join(f"{cn}={v:.0f} dur={d/1e3:.0f}us" for cn, v, d in last))

if __name__ == "__main__":
    kernel_stats(sys.argv[1])
    for p in sys.argv[2:]:
        pmc_stats(p)
