"""Compare two MTG_BUILD_REMARKS directories (per-kernel resource usage of two builds): registers, scratch, spills.
usage: python tools/compare_remarks.py <base_dir> <new_dir> [--all]"""
import glob
import os
import re
import sys


def parse(d):
    out = {}
    for f in glob.glob(os.path.join(d, "*.log")):
        name = None
        for line in open(f, errors="replace"):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {"tu": os.path.basename(f)[:-4]}
                continue
            m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|TotalSGPRs): (\d+)", line)
            if m and name:
                out[name][m.group(1).split(" [")[0]] = int(m.group(2))
    return out


def main():
    a, b = parse(sys.argv[1]), parse(sys.argv[2])
    show_all = "--all" in sys.argv
    worse = 0
    for k in sorted(b):
        if k not in a:
            continue
        x, y = a[k], b[k]
        ds = y.get("ScratchSize", 0) - x.get("ScratchSize", 0)
        dv = y.get("VGPRs Spill", 0) - x.get("VGPRs Spill", 0)
        dr = (y.get("VGPRs", 0) + y.get("AGPRs", 0)) - (x.get("VGPRs", 0) + x.get("AGPRs", 0))
        if show_all or ds or dv:
            worse += ds > 0 or dv > 0
            print(f"{y['tu']:18s} regs {x.get('VGPRs',0)+x.get('AGPRs',0):4d}->{y.get('VGPRs',0)+y.get('AGPRs',0):4d} ({dr:+d}) "
                  f"scratch {x.get('ScratchSize',0):5d}->{y.get('ScratchSize',0):5d} vspill {x.get('VGPRs Spill',0):4d}->{y.get('VGPRs Spill',0):4d}  {k[:110]}")
    print(f"{len(b)} kernels, {worse} with more scratch / spills than the base")


main()
