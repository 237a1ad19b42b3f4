/* A plain C client of include/dpomp.h: what a ccall / cgo / JNI binding does, without Python or torch.
 * Usage: c_abi_client <particles> <seed>
 * Builds the Pooley SIS model by hand, runs dpomp_pf_loglik for two parameter vectors and prints
 *   ll0 ll1 status0 status1 events
 * Exit code 3 + the library's message when no GPU is present (there is no CPU path). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dpomp.h"

int main(int argc, char** argv) {
    int64_t P = argc > 1 ? atoll(argv[1]) : 1024;
    uint64_t seed = argc > 2 ? strtoull(argv[2], NULL, 10) : 1;
    static const double obs_time[5] = {20.0, 40.0, 60.0, 80.0, 100.0};   /* data/pooley.csv */
    static const int64_t obs_id[5] = {1, 1, 1, 1, 1};
    static const int64_t obs_value[10] = {0, 18, 0, 65, 0, 70, 0, 66, 0, 67};
    dpomp_model_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = DPOMP_ABI_VERSION;
    d.n_states = 2; d.n_events = 2; d.n_params = 2;
    d.builtin_id = DPOMP_BUILTIN_SIR;            /* sir_rf, hmm_examples.jl:107-110 */
    d.initial_condition[0] = 100; d.initial_condition[1] = 1;
    d.transition[0][0] = -1; d.transition[0][1] = 1;   /* m_transition [-1 1; 1 -1], :181 */
    d.transition[1][0] = 1;  d.transition[1][1] = -1;
    d.rate[0].theta_index = 0; d.rate[0].form = DPOMP_RATE_LEFT; d.rate[0].n_factors = 2;
    d.rate[0].factor_mask[0] = 1u; d.rate[0].factor_mask[1] = 2u;      /* theta1 * S * I */
    d.rate[1].theta_index = 1; d.rate[1].form = DPOMP_RATE_LEFT; d.rate[1].n_factors = 1;
    d.rate[1].factor_mask[0] = 2u;                                      /* theta2 * I */
    d.obs.kind = DPOMP_OBS_GAUSSIAN; d.obs.state_mask = 2u; d.obs.value_mask = 2u; d.obs.sigma = 2.0;
    d.n_obs = 5; d.n_obs_values = 2;
    d.obs_time = obs_time; d.obs_id = obs_id; d.obs_value = obs_value;

    dpomp_handle* h = NULL;
    int rc = dpomp_create(&d, 0, &h);
    if (rc != DPOMP_OK) {
        fprintf(stderr, "dpomp_create failed (%d): %s\n", rc, dpomp_last_global_error());
        return rc == DPOMP_EDEVICE ? 3 : 4;
    }
    const double theta[4] = {0.003, 0.1, 0.0032, 0.09};   /* K x B column-major: two parameter vectors */
    double ll[2];
    uint32_t status[2];
    rc = dpomp_pf_loglik(h, theta, 2, P, seed, 0, ll, status);
    if (rc != DPOMP_OK) {
        fprintf(stderr, "dpomp_pf_loglik failed (%d): %s\n", rc, dpomp_last_error(h));
        dpomp_destroy(h);
        return 5;
    }
    uint64_t events = 0;
    dpomp_last_event_count(h, &events);
    printf("%.17g %.17g %u %u %llu\n", ll[0], ll[1], status[0], status[1], (unsigned long long)events);
    dpomp_destroy(h);
    return 0;
}
