import sys, time, numpy as np
sys.path.insert(0, '.')
import dpomp_amd as dp
y = dp.get_observations("tests/golden/pooley.csv")
pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)
for variant in (0, 1):
    pf.set_propagate_variant(variant)
    for B, P in [(1, 64), (1, 200), (4, 512), (4, 1000), (64, 200), (64, 1000), (1000, 200), (4000, 200), (1, 1 << 14), (1, 1 << 17)]:
        th = np.tile(np.array([[0.003], [0.1]]), (1, B))
        pf.loglik(th, P, seed=1)
        t = time.perf_counter(); n = 20
        for i in range(n): pf.loglik(th, P, seed=i)
        dt = (time.perf_counter() - t) / n
        pf.set_profiling(True); pf.loglik(th, P, seed=99); pf.set_profiling(False)
        ks = pf.kernel_stats()
        print(f"variant {variant:2d} B={B:5d} P={P:7d}: {dt*1e3:7.3f} ms/call  ({B*P*5/dt/1e6:9.1f} M particle-steps/s)")
pf.set_profiling(False)
print({k: (n, round(ms / max(n, 1) * 1e3, 1)) for k, (n, ms) in pf.kernel_stats().items()})
