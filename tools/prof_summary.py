#!/usr/bin/env python
"""Turn a rocprofv3 (--kernel-trace --stats) rocpd SQLite database into a text kernel summary for profiles/."""
import ctypes
import os
import sqlite3
import sys

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "powerpaint_amd", "libpp_hip.so")


def build_id():
    """pp_build_id() of the in-tree library (digest of the sources it was built from), without importing torch."""
    lib = ctypes.CDLL(LIB)
    lib.pp_build_id.restype = ctypes.c_char_p
    return lib.pp_build_id().decode()


def main(db, out, title=""):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary  {title}\n# source: {db}\n")
        try:       # the build this trace was taken from (bench.py compares it with the library it runs)
            f.write(f"# lib_sha16: {build_id()}\n")
        except (OSError, AttributeError):
            pass
        f.write(f"# {'calls':>7} {'total_ms':>11} {'avg_us':>10} {'pct':>6}  kernel\n")
        for name, calls, tot, avg, pct in rows:
            if len(name) > 150:
                name = name[:147] + "..."
            f.write(f"  {calls:7d} {tot / 1e3:11.3f} {avg:10.2f} {pct:6.2f}  {name}\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
