"""MI355X-native bootstrap particle filter behind DiscretePOMP.jl's model plugin surface.

The directory name carries a dot (``discretepomp.jl_amd``), which Python cannot import by
name; import it through the repo-root shim ``dpomp_amd`` (``import dpomp_amd``).
"""
from ._abi import (DpompLibraryMissing, ST_EVENT_CAP, ST_ZERO_WEIGHT, RS_SYSTEMATIC, RS_STRATIFIED,
                   RS_MULTINOMIAL, DBG_LOGW, DBG_OFFSPRING_END, load_library)
from .model import (DPOMPModel, GaussianObsModel, ModelDescriptor, Observation, RateFunction, RateSource, RateTerm, SourceObsModel,
                    UniformProductPrior, dmy_obs_fn, generate_custom_model, generate_model,
                    generate_weak_prior, get_observations, partial_gaussian_obs_model)
from .smc2 import (ImportanceSample, compute_ess, compute_is_mu_covar, rs_systematic, run_ibis_analysis, run_pibis,
                   run_smc2_analysis)
from .pmcmc import MCMCSample, RejectionSample, gelman_sre, handle_rej_samples, run_pmcmc, run_pmcmc_analysis
from .sim import Event, Particle, SimResults, generate_observations, gillespie_sim, save_to_file
from .arq import ARQMCMCSample, ARQModel, GridPoint, run_arq_mcmc_analysis, run_inner_mcmc_analysis
from . import parallel  # noqa: F401
from .particle_filter import (C_DF_ESS_CRIT, C_DF_PF_P, DpompError, EventCapWarning, ParticleFilter, Population, check_rate_source,
                              get_particle_filter_lpdf)

__all__ = [n for n in dir() if not n.startswith("_")]
