import sys, numpy as np
sys.path.insert(0,'.')
import dpomp_amd as dp
y = dp.get_observations("tests/golden/pooley.csv")
pf = dp.ParticleFilter(dp.generate_model("SIS",[100,1]), y)
for variant in (0,1):
    pf.set_propagate_variant(variant)
    for P in (1<<20, 1<<18):
        for i in range(5): pf.loglik([0.0,0.0],P,seed=i)
        pf.set_profiling(1)
        for i in range(20): pf.loglik([0.0,0.0],P,seed=i)
        st=pf.kernel_stats(); pf.set_profiling(0)
        print(variant, P, {k:round(ms/n*1e3,2) for k,(n,ms) in st.items() if n})
        pf._lib.dpomp_set_profiling(pf._h, 0)
        pf2 = None
    # reset stats by recreating
    pf.close(); pf = dp.ParticleFilter(dp.generate_model("SIS",[100,1]), y)
