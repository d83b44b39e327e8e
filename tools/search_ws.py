import subprocess, sys, re, json, concurrent.futures as cf
CSRC='/root/repo/mav_trajectory_generation_amd/csrc'
SHAPES={4:(15,1,15,3), 5:(31,1,31,4), 6:(63,1,63,5)}
LDSMAX={4:8,5:5,6:3}
def res(H,K,WS,LS):
    ms,mi,me,dv=SHAPES[H]
    cfg=f"MtgCfg<{H},1,{K},{ms},{mi},{me},{dv},0,{WS},{3 if WS>0 else 0},{LS}>"
    src=f"/tmp/s_{H}_{K}_{WS}.hip"
    open(src,'w').write('#include "mtg_dimlane.h"\ntemplate __global__ void mtg_solve_dl_kernel<'+cfg+', 3, 1, 0, 18>(const double*, const double*, double*, int*, int*, int, int, int, double*);\n')
    p=subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-I'+CSRC,'-I/root/repo/include','-mllvm','-disable-machine-licm','-mllvm','-amdgpu-kernarg-preload-count=14','-mllvm','-pragma-unroll-threshold=1000000','--cuda-device-only','-c',src,'-o',src+'.o','-Rpass-analysis=kernel-resource-usage'],capture_output=True,text=True)
    t=p.stderr
    if 'error' in t: return None
    sc=int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)',t).group(1)); sp=int(re.search(r'VGPRs Spill: (\d+)',t).group(1))
    return sc,sp
def best(H,K):
    kc=(K+1)//2
    for WS in range(0,kc+1):
        LS=min(WS,LDSMAX[H])
        r=res(H,K,WS,LS)
        if r is None: return (H,K,None,None,'compile error')
        if r[0]==0 and r[1]<=16: return (H,K,WS,LS,r)
    return (H,K,None,None,'no fit')
jobs=[(H,K) for H in (4,5,6) for K in range(3,17) if K % 2 == 1]
with cf.ThreadPoolExecutor(7) as ex:
    for r in ex.map(lambda a: best(*a), jobs):
        print(r, flush=True)
