import sys, ctypes
sys.path.insert(0, "/root/repo")
import mav_trajectory_generation_amd as m
ctx = m.Context(0)
for n in (-(1 << 20) - 0, -((1 << 20) + 1), -((1 << 20) + 2), -((1<<20)+3)):
    print("iters", (-n) & 3, "max rel err", ctx.selftest_rcp(n))
