"""agenda_amd -- MI355X-native (gfx950) implementation of the AGenDA data-generation hot path:
Stable-Diffusion UNet denoise loop + DAAM cross-attention heat maps + VAE decode, as hand-written
HIP kernels behind a C ABI (include/agenda_hip.h), with the Python call surfaces the reference
scripts use (StableDiffusionPipeline, daam.trace, the attention-processor hooker)."""
from . import config, synthetic  # noqa: F401
from .pipeline import StableDiffusionPipeline, PipelineOutput, Engine  # noqa: F401
from .trace import trace, GlobalHeatMap, WordHeatMap, compute_token_merge_indices  # noqa: F401
from .hook import UNetCrossAttentionHooker  # noqa: F401
from .scheduler import DDIMScheduler  # noqa: F401

__all__ = ["StableDiffusionPipeline", "PipelineOutput", "Engine", "trace", "GlobalHeatMap", "WordHeatMap",
           "compute_token_merge_indices", "UNetCrossAttentionHooker", "DDIMScheduler", "config", "synthetic"]
