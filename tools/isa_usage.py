#!/usr/bin/env python
"""Per-kernel resource summary from a `hipcc -save-temps` gfx950 .s file (VGPRs, SGPRs, scratch, code bytes), and
optional extraction of one kernel's body:  python tools/isa_usage.py file.s [filter-regex] [--dump DIR]"""
import re
import subprocess
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "."
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    s = open(path).read()
    for m in re.finditer(r"^(_Z\w+): *; @", s, re.M):
        nm = m.group(1)
        dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").replace("(PPGemmArgs, GemmDerived)", "")
        if not re.search(flt, dem):
            continue
        j = s.index(".Lfunc_end", m.end())
        k = s.index("; NumVgprs", j)
        blk = s[j:k + 900]
        g = lambda key: re.search(key + r":? *=? *(\d+)", blk).group(1)
        print(f"{dem[:78]:78s} vgpr {g('NumVgprs'):>3} sgpr {g('NumSGPRsForWavesPerEU'):>3} scratch {g('ScratchSize'):>4} "
              f"code {g('codeLenInByte'):>6}")
        if dump:
            tag = re.sub(r"[^0-9A-Za-z]+", "_", dem)[:80]
            open(f"{dump}/{tag}.s", "w").write(s[m.start():j])


if __name__ == "__main__":
    main()
