#!/usr/bin/env python3
"""Resolve the raw PCs of a glog-style crash trace ("    @     0x7f... (unknown)") against a /proc/<pid>/maps dump of the SAME
process (bench.py writes one when VSLAM_DUMP_MAPS=<file> is set): library, offset inside the library, nearest preceding
dynamic / static symbol.  usage: resolve_stack.py <maps file> <crash log>   (run on the box whose libraries the process mapped)"""
import bisect
import re
import subprocess
import sys


def load_maps(path):
    maps = []
    for ln in open(path):
        p = ln.split()
        if len(p) < 5:
            continue
        a, b = (int(v, 16) for v in p[0].split("-"))
        maps.append((a, b, p[1], int(p[2], 16), p[5] if len(p) > 5 else ""))
    return maps


_sym = {}


def symbols(lib):
    if lib in _sym:
        return _sym[lib]
    tab = []
    for flags in (["-D"], []):
        try:
            out = subprocess.run(["nm", "-C", "--defined-only"] + flags + [lib], capture_output=True, text=True, timeout=120).stdout
        except Exception:      # noqa: BLE001
            continue
        for ln in out.splitlines():
            m = re.match(r"^([0-9a-f]+) (\w) (.*)$", ln)
            if m and m.group(2) in "tTwWiu":
                tab.append((int(m.group(1), 16), m.group(3)))
    tab.sort()
    _sym[lib] = tab
    return tab


def resolve(maps, addr):
    for a, b, perm, off, path in maps:
        if a <= addr < b:
            if not path.startswith("/"):
                return "%s %s [%x-%x]" % (path or "anonymous", perm, a, b), None
            v = addr - a + off      # (text segments of shared objects: p_vaddr == p_offset)
            tab = symbols(path)
            name = None
            if tab:
                i = bisect.bisect_right(tab, (v, "\xff")) - 1
                if i >= 0:
                    name = "%s+0x%x" % (tab[i][1], v - tab[i][0])
            return "%s+0x%x" % (path, v), name
    return "unmapped", None


def main():
    maps = load_maps(sys.argv[1])
    for ln in open(sys.argv[2], errors="replace"):
        m = re.search(r"(PC: )?@\s+(0x[0-9a-f]+)", ln)
        f = re.search(r"SIGSEGV \(@(0x[0-9a-f]+)\)", ln)
        if f:
            a = int(f.group(1), 16)
            where = "not inside any mapping"
            for s, e, perm, off, path in maps:
                if s <= a < e:
                    where = "inside %s %s [%x-%x]" % (path or "anonymous", perm, s, e)
                if e == a:
                    where += "; the byte after the END of %s %s [%x-%x] (%d KB)" % (path or "anonymous", perm, s, e, (e - s) // 1024)
            print("fault address %s: %s" % (f.group(1), where))
        if m:
            lib, name = resolve(maps, int(m.group(2), 16))
            print("%s  %-60s %s" % (m.group(2), lib, name or ""))


if __name__ == "__main__":
    main()
