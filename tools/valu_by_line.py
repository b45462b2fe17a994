#!/usr/bin/env python
"""Attribute the static VALU instructions of one kernel to source lines (developer tool).

    python tools/valu_by_line.py dig_nb.hip element_stats_stream_kernelILb0ELb1 [--flags "-mllvm -disable-machine-licm"]

Compiles the file for gfx950 with -gline-tables-only, walks the kernel's assembly and counts the v_* instructions
under every .loc.  Found this way: 162 instructions of the library log inlined into the statistics stream pass through
the Fisher fallback, executed whenever a tile held a parked pair."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--flags", default="")
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    out = "/tmp/valu_by_line.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-gline-tables-only",
           os.path.join(ROOT, "digdriver_amd", "csrc", a.source), "-o", out] + a.flags.split()
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(a.kernel), l)]
    if not starts:
        sys.exit("kernel not found")
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    cur, cnt = None, collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        elif re.match(r"\s+v_", l):
            cnt[cur] += 1
    print(lines[start].rstrip(":"), "static VALU:", sum(cnt.values()))
    for (f, ln), c in sorted(cnt.items(), key=lambda kv: -kv[1])[: a.top]:
        print("%-28s %5d  %4d" % (f, ln, c))


if __name__ == "__main__":
    main()
