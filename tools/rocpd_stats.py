#!/usr/bin/env python3
"""Per-kernel summary (calls, total / avg / min / max duration) of a rocprofv3
rocpd sqlite database -- the same table `rocprofv3 --stats` prints as CSV.

    python tools/rocpd_stats.py gpurun_out/<tag>/prof/prof_results.db [out.md]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, "
        "min(end-start)/1e3, max(end-start)/1e3 from kernels group by name "
        "order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"total kernel time {tot:.1f} us over {sum(r[1] for r in rows)} dispatches", "",
             "| % | calls | total us | avg us | min us | max us | kernel |",
             "|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {r[2] / tot * 100:.2f} | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | "
                     f"{r[4]:.2f} | {r[5]:.2f} | `{r[0][:140]}` |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
