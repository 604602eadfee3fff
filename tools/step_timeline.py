#!/usr/bin/env python
"""Per-dispatch timeline of ONE denoise step from a rocprofv3 --kernel-trace rocpd SQLite database: kernel durations and
the gaps between consecutive kernels of the graph replay (end of kernel i-1 -> start of kernel i).

    python tools/step_timeline.py <results.db> <out.txt> [period]

The step is found as the shortest period of the kernel-name sequence at the end of the trace (the graph replays the same
launch list every step), unless `period` is given.  Durations / gaps are averaged over the last REPS periods.
"""
import collections
import re
import sqlite3
import sys

REPS = 20


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<.*>)?", name)
    base = m.group(1) if m else name
    targs = (m.group(2) or "") if m else ""
    return (base + targs)[:70]


def load(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    for cand in ("kernels", "rocpd_kernel_dispatch"):
        view = [t for t in tabs if t == cand or t.startswith(cand)]
        if view:
            cols = [r[1] for r in c.execute(f"pragma table_info({view[0]})")]
            if "name" in cols and "start" in cols and "end" in cols:
                return list(c.execute(f"select name, start, end from {view[0]} order by start"))
    raise SystemExit("no per-dispatch view found; tables: " + ", ".join(tabs))


def main():
    db, out = sys.argv[1], sys.argv[2]
    rows = load(db)
    names = [r[0] for r in rows]
    n = len(names)
    period = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    off = 0
    if not period:
        # (the trace ends with a few non-periodic launches -- output conversion, teardown: allow a tail offset)
        for off in range(0, 400):
            m = n - off
            for p in range(50, 2000):
                if m >= 3 * p and names[m - p:m] == names[m - 2 * p:m - p] == names[m - 3 * p:m - 2 * p]:
                    period = p
                    break
            if period:
                break
    if not period:
        raise SystemExit(f"no period found in {n} dispatches; tail: " + " | ".join(short(x)[:30] for x in names[-12:]))
    rows, names = rows[:n - off], names[:n - off]
    n -= off
    reps = min(REPS, n // period - 1)
    dur = [0.0] * period
    gap = [0.0] * period
    for r in range(reps):
        base = n - (r + 1) * period
        for i in range(period):
            nm, s, e = rows[base + i]
            dur[i] += (e - s) / reps
            gap[i] += (s - rows[base + i - 1][2]) / reps
    with open(out, "w") as f:
        f.write(f"# one denoise step = {period} kernels (period of the trace's tail), averaged over {reps} replays; ns -> us\n")
        f.write(f"# sum of kernel durations {sum(dur) / 1e3:.1f} us, sum of gaps {sum(gap) / 1e3:.1f} us, "
                f"step {(sum(dur) + sum(gap)) / 1e3:.1f} us\n")
        fam = collections.OrderedDict()
        for i in range(period):
            k = short(names[n - period + i])
            a = fam.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += dur[i]
            a[2] += gap[i]
        f.write("# by kernel: count, total us, avg us, total gap-before us, avg gap-before us\n")
        for k, (cnt, d, g) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            f.write(f"#  {cnt:4d} {d / 1e3:9.1f} {d / 1e3 / cnt:8.2f} {g / 1e3:8.1f} {g / 1e3 / cnt:6.2f}  {k}\n")
        f.write("# idx  dur_us  gap_before_us  kernel\n")
        for i in range(period):
            f.write(f"{i:4d} {dur[i] / 1e3:8.2f} {gap[i] / 1e3:7.2f}  {short(names[n - period + i])}\n")
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()
