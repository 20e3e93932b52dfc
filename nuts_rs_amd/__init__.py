"""nuts_rs_amd — MI355X-native many-chain NUTS engine behind the nuts-rs chain-driver interface.

The compute path is hand-written HIP for gfx950 in csrc/ exposed through the C ABI of include/nuts_amd.h
(libnuts_amd.so); this package is the thin host-side mirror of the reference's Settings/Chain interface.
Importing the package does not need a GPU; creating a ChainBatch does (there is no CPU fallback).
"""
from ._lib import NutsAmdError, STATS_DTYPE, VECTOR_STATS, load as load_library  # noqa: F401
from .sampler import (AdamOptions, ChainBatch, DiagAdaptExpSettings, DiagNutsSettings, DualAverageOptions,  # noqa: F401
                      EuclideanAdaptOptions, LogpSpec, LowRankNutsSettings, KineticEnergyKind, DiagMclmcSettings, LowRankMclmcSettings, MclmcTrajectoryKind, RecoverableLogpError, LowRankSettings, Progress, StepSizeSettings, sample,
                      ADAPT_DIAG, ADAPT_LOW_RANK,
                      LOGP_IID_NORMAL, LOGP_DIAG_NORMAL, LOGP_FUNNEL, LOGP_EIGHT_SCHOOLS, LOGP_MVN_PREC, LOGP_MODULE, LOGP_HOST_CALLBACK,
                      STEP_DUAL_AVERAGE, STEP_ADAM, STEP_FIXED)

from .controller import ChainProgress, ProgressCallback, Sampler, SamplerWaitResult  # noqa: F401

__all__ = ["ChainProgress", "ProgressCallback", "Sampler", "SamplerWaitResult", "AdamOptions", "ChainBatch", "DiagNutsSettings", "EuclideanAdaptOptions", "StepSizeSettings", "DualAverageOptions",
           "DiagAdaptExpSettings", "LowRankNutsSettings", "KineticEnergyKind", "DiagMclmcSettings", "LowRankMclmcSettings", "MclmcTrajectoryKind", "LowRankSettings", "LogpSpec", "Progress", "sample", "NutsAmdError", "STATS_DTYPE", "VECTOR_STATS",
           "load_library"]
