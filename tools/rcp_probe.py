import sys; sys.path.insert(0,'.')
import mav_trajectory_generation_amd as m
c = m.Context(0)
for it in (0,1,2,3):
    n = -((1<<20) | it) if it != 2 else (1<<20)
    print("newton steps", it, "max rel err", c.selftest_rcp(n))
