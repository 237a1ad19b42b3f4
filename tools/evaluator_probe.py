import sys, time, numpy as np
sys.path.insert(0, '.')
import dpomp_amd as dp
y = dp.get_observations("tests/golden/pooley.csv")
SRC = """__device__ void dpomp_user_rates(const double* theta, const int* x, double* r) {
    r[0] = theta[0] * (double)x[0] * (double)x[1];
    r[1] = theta[1] * (double)x[1];
}"""
P = 1 << 20
def run(pf, label):
    pf.loglik([0.003, 0.1], P, seed=1)
    t = time.perf_counter()
    lls = [pf.loglik([0.003, 0.1], P, seed=10 + i)[0] for i in range(5)]
    dt = (time.perf_counter() - t) / 5
    print(f"{label:28s} {dt*1e3:7.3f} ms/call  {P*5/dt/1e9:6.3f} G particle-steps/s  ll[0]={lls[0]!r}")
    return lls
a = run(dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y), "built-in RSir<2>")
b = run(dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y, force_generic=True, jit=False), "table interpreted RGeneric<2,2>")
t0 = time.perf_counter()
pfj = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y, force_generic=True, jit=True)
print(f"table -> source -> hipRTC (first create): {time.perf_counter() - t0:.2f} s")
t0 = time.perf_counter()
dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y, force_generic=True, jit=True).close()
print(f"second create of the same table model: {(time.perf_counter() - t0) * 1e3:.1f} ms")
d = run(pfj, "table compiled at create")
assert d == a
t0 = time.perf_counter()
pfc = dp.ParticleFilter(dp.generate_custom_model("SIS", dp.RateSource(SRC, 2, 2), [100, 1], [[-1, 1], [1, -1]]), y)
print(f"hipRTC compile + module load: {time.perf_counter() - t0:.2f} s")
c = run(pfc, "run-time compiled source")
assert a == b == c, "the three evaluators must agree bit for bit"
print("all three evaluators: identical log-likelihoods")
