#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (``--kernel-trace``) into the per-kernel text summary kept under profiles/."""
import glob
import sqlite3
import sys


def main(path, title=""):
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    con = sqlite3.connect(dbs[0])
    rows = list(con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                            "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# {title}")
    print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:7d} {r[2]:10.2f} {r[3]:9.1f} {r[4]:8.1f} {r[5]:9.1f} {100 * r[2] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
