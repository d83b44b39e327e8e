import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
import mav_trajectory_generation_amd as m
ctx=m.Context(0)
for (n,k,dim,d,mi) in ((8,8,3,3,1),(10,8,3,4,1),(12,16,3,5,1)):
    masks=m.ends_full_masks(n,k,mi)
    plan=m.Plan(ctx,n,dim,k,d,masks)
    for bsz in (1,22):
        t,f=m.random_waypoint_batch(bsz,k,dim,n,masks,seed=17*n+k,device="cuda",layout="soa")
        co,fr,cost=plan.solve(t,f,layout="soa",want_free=True,want_cost=True)
        cf,ff,jf=plan.solve(t,f,layout="soa",want_free=True,want_cost=True,dims="fused")
        cs,fs,js=plan.solve(t,f,layout="soa",want_free=True,want_cost=True,dims="split")
        ctx.sync()
        print(n,k,bsz, plan.launch_form(bsz,"soa",extra_outputs=True), ((cost-jf).abs()/jf.abs()).max().item(), ((js-jf).abs()/jf.abs()).max().item(), cost[:2].tolist(), jf[:2].tolist())
