#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocpd sqlite, `--kernel-trace --stats`) into the per-kernel summary
table committed under profiles/.  Usage: tools/rocprof_summary.py <results.db> [<out.md>] [--note TEXT]"""
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    note = ""
    if "--note" in sys.argv:
        note = sys.argv[sys.argv.index("--note") + 1]
        args = [a for a in args if a != note]
    db = sqlite3.connect(args[0])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = []
    if note:
        lines += [note, ""]
    lines += ["| kernel | calls | total (us) | avg (us) | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows:
        if pct < 0.005:
            continue
        n = name.replace("void ", "").replace("vtts::", "")
        if len(n) > 110:
            n = n[:107] + "..."
        lines.append(f"| `{n}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
    total = sum(r[2] for r in rows)
    lines.append(f"| **all kernels** | {sum(r[1] for r in rows)} | {total:.1f} | | 100 |")
    out = "\n".join(lines) + "\n"
    if len(args) > 1:
        with open(args[1], "w") as f:
            f.write(out)
    else:
        print(out)


if __name__ == "__main__":
    main()
