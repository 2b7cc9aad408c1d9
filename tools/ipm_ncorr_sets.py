import sys, numpy as np
sys.path.insert(0,'.')
from tests import ipm_model
from oracle import lmpc_oracle as orc
dd=np.load('./build_tmp/ipm_sets.npz'); d40={k:v for k,v in np.load('./build_tmp/n40_set.npz').items()}
sets={}
for name in ("bench","fast"):
    N=dd[name+"_A"].shape[1]; p=orc.QPParams.lmpc_default(N)
    sets[name]=[ipm_model.StructQP(p,*[dd["%s_%s"%(name,k)][i] for k in ("A","B","C","x0","uOld","SS","Qsel")]) for i in range(dd[name+"_x0"].shape[0])]
p40=orc.QPParams.lmpc_default(40)
sets["n40"]=[ipm_model.StructQP(p40,d40["A"][i],d40["B"][i],d40["C"][i],d40["x0"][i],d40["uOld"][i],d40["SS"][i],d40["Qsel"][i]) for i in range(0,1024,8)]
for nc in eval(sys.argv[1]):
    for name,qs in sets.items():
        its=[];ncs=[]
        for q in qs:
            with np.errstate(all="ignore"): r=ipm_model.ipm_solve(q,ncorr=nc)
            its.append(r["iters"]); ncs.append(r.get("ncorr",0))
        its=np.array(its); ncs=np.array(ncs); eff=its+ncs/3.0
        print("%-28s %-6s mean %.2f max %d | extra solves mean %.2f | effective (1/3 per solve) mean %.2f max %.2f  hist %s"%(nc,name,its.mean(),its.max(),ncs.mean(),eff.mean(),eff.max(),np.bincount(its)[5:].tolist()),flush=True)
