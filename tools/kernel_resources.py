"""Kernel resource usage from the build's remark logs (MTG_BUILD_REMARKS=<dir> python -c 'import __graft_entry__ as g; g.build()'):
every kernel with scratch (spills), or all kernels of the named translation units.  usage: kernel_resources.py DIR [tu ...]"""
import os, re, sys
d = sys.argv[1]
only = set(sys.argv[2:])
for fn in sorted(os.listdir(d)):
    tu = fn[:-4]
    txt = open(os.path.join(d, fn)).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split()[0]
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        scratch = g(r"ScratchSize \[bytes/lane\]")
        if (only and tu in only) or (not only and scratch not in ("0", "?")):
            m = re.search(r"MtgCfgILi(\d)ELi(\d)ELi(-?\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d)ELi(\d)ELi(\d+)ELi(\d)ELi(\d+)ELi(\d)E", name)
            cfg = "H%s D%s K%s MI%s WS%s DLW%s LS%s RS%s" % (m.group(1), m.group(2), m.group(3), m.group(5), m.group(9), m.group(10), m.group(11), m.group(12)) if m else ""
            kind = re.match(r"_Z\d+(\w+?)I", name)
            print(f"{tu:20s} {(kind.group(1) if kind else name[:40]):34s} {cfg:40s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} scratch {scratch:>5} occ {g('Occupancy .waves/SIMD.')}")
