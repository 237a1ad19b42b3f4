"""The CPU restatement under AddressSanitizer + UBSan (GPU sanitizers are not available on the GPU pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "dpomp.h"
int orc_pf_loglik(const dpomp_model_desc*, int64_t, const double*, uint64_t, uint32_t, int, int, int, double*, uint32_t*, uint64_t*, const void*);
int64_t orc_simulate_events(const dpomp_model_desc*, const double*, uint64_t, uint32_t, uint32_t, int64_t, double*, int32_t*, int64_t*);
int main(void) {
    static const double t[5] = {20, 40, 60, 80, 100};
    static const int64_t id[5] = {1, 1, 0, 1, 1}, val[10] = {0, 18, 0, 65, 0, 70, 0, 66, 0, 67};
    dpomp_model_desc d; memset(&d, 0, sizeof d);
    d.abi_version = DPOMP_ABI_VERSION; d.n_states = 2; d.n_events = 2; d.n_params = 2;
    d.initial_condition[0] = 100; d.initial_condition[1] = 1;
    d.transition[0][0] = -1; d.transition[0][1] = 1; d.transition[1][0] = 1; d.transition[1][1] = -1;
    d.rate[0].n_factors = 2; d.rate[0].factor_mask[0] = 1; d.rate[0].factor_mask[1] = 2;
    d.rate[1].theta_index = 1; d.rate[1].n_factors = 1; d.rate[1].factor_mask[0] = 2;
    d.obs.kind = 1; d.obs.state_mask = 2; d.obs.value_mask = 2; d.obs.sigma = 2.0;
    d.n_obs = 5; d.n_obs_values = 2; d.obs_time = t; d.obs_id = id; d.obs_value = val;
    const double theta[2] = {0.003, 0.1};
    for (int rs = 1; rs <= 3; ++rs) for (int mode = 0; mode <= 1; ++mode) for (int P = 1; P <= 333; P += 83) {
        double ll; uint32_t st; uint64_t ev;
        if (orc_pf_loglik(&d, P, theta, 7, 2, mode, rs, 1, &ll, &st, &ev, 0)) return 1;
    }
    static double times[500]; static int32_t types[500]; int64_t states[10];
    orc_simulate_events(&d, theta, 3, 0, 0, 500, times, types, states);   /* truncation path */
    puts("ok");
    return 0;
}
'''


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    src = tmp_path / "drv.c"
    src.write_text(DRIVER)
    exe = tmp_path / "drv"
    fma = ["-mfma"] if " fma" in open("/proc/cpuinfo").read() else []
    r = subprocess.run(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-ffp-contract=off", *fma, "-I", os.path.join(ROOT, "include"), str(src),
                        os.path.join(ROOT, "oracle", "dpomp_oracle.c"), "-o", str(exe), "-lm"], capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower():
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stderr[-2000:]
