import sys, time, numpy as np
sys.path.insert(0, '.')
import dpomp_amd as dp
y = dp.get_observations("tests/golden/pooley.csv")
model = dp.generate_model("SIS", [100, 1]); model.prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
r = []
for s in range(1, 11):
    t = time.time()
    o = dp.run_arq_mcmc_analysis(model, y, [0.0005, 0.02], seed=s, steps=20000, n_chains=5, verbose=False)
    r.append((o.imp_sample.mu[0], o.imp_sample.mu[1], o.samples.mu[0], o.samples.mu[1], len(o.sample_cache), time.time() - t))
r = np.array(r); np.set_printoptions(precision=6, suppress=True)
print(r); print("mean", r.mean(0)); print("sd", r.std(0))
