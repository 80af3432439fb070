#!/usr/bin/env python3
"""Per-kernel time summary of a rocprofv3 kernel trace stored as a rocpd sqlite database (what `rocprofv3 --kernel-trace` writes when
no --output-format is given): name, calls, total / average duration, share. Template arguments of the engine's kernels are kept
(the tile variant matters); other C++ decoration is trimmed. Usage: tools/rocpd_stats.py <results.db> [top_n]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, st, en in rows:
        n = re.sub(r"^void ", "", name)
        n = re.sub(r"\(.*$", "", n)
        n = n.replace("fe::", "")
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += en - st
    total = sum(v[1] for v in agg.values())
    print(f"# {len(rows)} dispatches, {total / 1e6:.3f} ms of kernel time; columns: name, calls, total_ms, avg_us, percent   ({cols[:0]})")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{n[:110]:110s} {c:7d} {t / 1e6:10.3f} {t / c / 1e3:9.1f} {100.0 * t / total:6.2f}")


if __name__ == "__main__":
    main()
