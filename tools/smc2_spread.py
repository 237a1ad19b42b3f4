import sys, time, numpy as np
sys.path.insert(0, '.')
import dpomp_amd as dp
y = dp.get_observations("tests/golden/pooley.csv")
model = dp.generate_model("SIS", [100, 1]); model.prior = dp.UniformProductPrior([0.0, 0.0], [0.01, 0.5])
for corr in (False, True):
    r = []
    t = time.time()
    for s in range(1, 21):
        o = dp.run_ibis_analysis(model, y, seed=s, verbose=False, proposal_correction=corr)
        r.append((o.bme[0], o.bme[1], o.mu[0], o.mu[1], o.n_resample, o.acceptance_rate))
    r = np.array(r)
    print("correction", corr, "time/run %.2fs" % ((time.time() - t) / 20))
    print(" mean", np.round(r.mean(0), 5)); print(" sd  ", np.round(r.std(0), 5)); print(" bme0 all", np.round(r[:, 0], 3))
