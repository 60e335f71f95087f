"""redclust.jl_amd — MI355X-native Gibbs-sweep hot path of RedClust.jl behind the reference's
runsampler / MCMCData / MCMCOptionsList / MCMCResult surface.

The directory name is not a valid Python identifier; import it through the repo-root shim:

    import redclust_amd as rc
"""
from ._lib import Context, Comm, measure_read_ceiling, RedClustHIPError, RedClustDomainError, build, build_diag, lib, SIGNATURES  # noqa: F401
from .types import MCMCData, MCMCOptionsList, MCMCResult, MCMCState, PriorHyperparamsList  # noqa: F401
from .sampler import runsampler, sample_r, sample_p, iac_ess_acf  # noqa: F401
from .datagen import generatemixture, likelihood_hyperparams, likelihood_hyperparams_device  # noqa: F401
from .chains import chain_seed, merge_chains, run_chains, run_chains_single_process, library_merge, agreed_merge, device_counts_tensor  # noqa: F401
from .pointestimate import (getpointestimate, lossmatrix, binderloss, infodist, varinfo, evaluateclustering,  # noqa: F401
                            summarise)
