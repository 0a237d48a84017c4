"""audio-pathtracer_amd — MI355X-native acoustic BDPT path (drop-in for FrequenSee's per-frame
ray-trace + energy-buffer loop).  The directory name carries a hyphen, so import it through
`__graft_entry__.load_package()` (registers it as `audio_pathtracer_amd`).

Contents: csrc/ (HIP kernels + C ABI -> libfrequensee.so), _capi (ctypes binding), component (host
mirror of UAudioRayTracingSubsystem / UFrequenSeeAudioComponent), scenes (procedural inputs),
sharding (multi-GPU pair partition).
"""
from . import _capi, scenes, sharding  # noqa: F401
from ._capi import FrequenSeeError, default_config, default_params  # noqa: F401
from .component import (AudioRayTracingSubsystem, Context, FrequenSeeAudioComponent,  # noqa: F401
                        FrequenSeeAudioReverbPlugin, MaterialAcousticProcessor)

__all__ = ["AudioRayTracingSubsystem", "FrequenSeeAudioComponent", "FrequenSeeAudioReverbPlugin", "MaterialAcousticProcessor", "Context",
           "FrequenSeeError",
           "default_config", "default_params", "scenes", "sharding"]
