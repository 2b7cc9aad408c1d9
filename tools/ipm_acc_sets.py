"""Developer tool (CPU): the termination rule of the interior-point iteration (lmpc_kernels.hip.h: accuracy_ok; tests/ipm_model.py: acc_rule) on the model's problem sets --
bench batch, fast laps (build_tmp/ipm_sets.npz: tools/ipm_model_sets.py build) and every 8th N = 40 bench problem (build_tmp/n40_set.npz: tools/n40_model.py build):
iterations and the worst distance of (x, u) from a solve at tolerances 1e-15 / 1e-11.       python tools/ipm_acc_sets.py"""
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
tight={}
for name,qps in sets.items():
    with np.errstate(all="ignore"):
        tight[name]=[ipm_model.ipm_solve(q,tol_gap=1e-15,tol_res=1e-11,acc_rule=None) for q in qps]
for rule in (None, dict(ratio=1e-2, step=0.0, floor=1e-12), dict(ratio=1e-3, step=0.0, floor=1e-12), dict(ratio=1e-3, step=0.0, floor=3e-13), "kernel"):
    for name,qps in sets.items():
        its=[];err=[]
        for q,t in zip(qps,tight[name]):
            with np.errstate(all="ignore"): r=ipm_model.ipm_solve(q,acc_rule=rule)
            its.append(r["iters"]); err.append(np.abs(np.concatenate([r["x"].ravel(),r["u"].ravel()])-np.concatenate([t["x"].ravel(),t["u"].ravel()])).max())
        its=np.array(its); err=np.array(err)
        print("rule %-52s %-6s mean %.3f max %d hist %s worst err %.2e n>1e-6 %d"%(str(rule),name,its.mean(),its.max(),np.bincount(its)[6:].tolist(),err.max(),(err>1e-6).sum()),flush=True)
