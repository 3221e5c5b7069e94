"""MI355X-native DFoT denoising engine (hot path only): HIP kernels behind a C ABI
(``csrc/`` -> ``libdfot_hip.so``, ``include/dfot_hip.h``) plus the host-side mirror of the
reference's backbone / sampler interfaces.  Importing it requires the built HIP library --
there is no CPU fallback."""
from . import capi  # noqa: F401  (raises ImportError when libdfot_hip.so is missing)
from . import ops  # noqa: F401  (registers the torch.library operators dfot::*)
from .backbone import UViT3DPose  # noqa: F401
from .dit_backbone import DiT3D, DifferenceDiT3D  # noqa: F401
from .diffusion import DiffusionConfig, Schedule  # noqa: F401
from .guidance import HistoryGuidance  # noqa: F401
from .sampler import (DFoTVideoPoseSampler, DFoTVideoSampler, DifferenceDFoTVideoSampler, SamplerConfig,  # noqa: F401
                      device_noise_fn)  # noqa: F401
from . import parallel  # noqa: F401,E402
from .training import ContextTraining, TrainingNoise, training_step_forward  # noqa: F401,E402
from .checkpoint import load_reference_checkpoint  # noqa: F401,E402
from .trainer import DiT3DTrainer  # noqa: F401,E402
from . import uvit_train  # noqa: F401,E402
from .vae import VideoVAEDecoder, decode_latents  # noqa: F401,E402
