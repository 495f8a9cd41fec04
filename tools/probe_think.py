import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import terastructure_amd as ts
n,l,k=1_000_000,64,8
e=ts.Engine(n,l,k)
e.set_gamma(np.random.default_rng(1).gamma(100,0.01,size=(n,k)))
for think in (0,500,1000,2000,3000):
    os.environ["TSAMD_PROBE_THINK_NS"]=str(think)
    print(think, e.probe_stream(30))
