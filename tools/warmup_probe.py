import sys, time, numpy as np
sys.path.insert(0,'.')
import dpomp_amd as dp
y = dp.get_observations("tests/golden/pooley.csv")
pf = dp.ParticleFilter(dp.generate_model("SIS",[100,1]), y)
P=1<<20
ts=[]
t0=time.perf_counter()
for i in range(400):
    t=time.perf_counter(); pf.loglik([0.003,0.1],P,seed=i); ts.append(time.perf_counter()-t)
ts=np.array(ts)*1e3
for i in range(0,400,20): print(i, round(ts[i:i+20].mean(),4), round(ts[i:i+20].min(),4))
