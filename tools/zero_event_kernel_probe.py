#!/usr/bin/env python3
"""Fixed cost of a propagation launch: a model in which nothing happens (all rates zero), per-kernel durations at two
sizes and both default schedules.  With DPOMP_HIP_LIB pointing at an experimental build the same probe dissects that
cost (e.g. a build without stream warm-up)."""
import sys

sys.path.insert(0, ".")
import dpomp_amd as dp  # noqa: E402

y = dp.get_observations("tests/golden/pooley.csv")
for variant in (0, 1):
    for P in (1 << 20, 1 << 18):
        pf = dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), y)
        pf.set_propagate_variant(variant)
        for i in range(5):
            pf.loglik([0.0, 0.0], P, seed=i)
        pf.set_profiling(1)
        for i in range(20):
            pf.loglik([0.0, 0.0], P, seed=i)
        st = pf.kernel_stats()
        pf.close()
        print(variant, P, {k: round(ms / n * 1e3, 2) for k, (n, ms) in st.items() if n})
