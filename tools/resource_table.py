"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (build log) per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
parts = re.split(r'remark: Function Name: (\S+)', txt)
seen = set()
print("%-66s %5s %5s %7s %4s %6s %6s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "sgprSp", "vgprSp"))
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if name in seen:
        continue
    seen.add(name)
    def g(k):
        m = re.search(k + r': (\d+)', body)
        return int(m.group(1)) if m else -1
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace('(anonymous namespace)::', '').replace('void ', '').replace('(MtgParams, int)', '').replace('(MtgParams)', '')
    if len(sys.argv) > 2 and not re.search(sys.argv[2], dem):
        continue
    print("%-66s %5d %5d %7d %4d %6d %6d" % (dem[:66], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                                           g(r'Occupancy \[waves/SIMD\]'), g('SGPRs Spill'), g('VGPRs Spill')))
