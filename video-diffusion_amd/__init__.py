"""MI355X-native sampling engine for the flexible video-diffusion model.

Host side mirrors the reference's Python call boundary; all
device math is hand-written HIP for gfx950 behind the C-ABI in `csrc/`
(declared in `include/vd_amd.h`).
"""
from . import weights_init  # noqa: F401
from . import _lib  # noqa: F401
from . import gaussian_diffusion, inference_util, respace, script_util, unet  # noqa: F401
from .script_util import (args_to_dict, create_gaussian_diffusion, create_video_model_and_diffusion,  # noqa: F401
                          str2bool, video_model_and_diffusion_defaults)


def param_specs(cfg):
    """[(checkpoint key, shape)] of the model a defaults-style config dict describes."""
    keys = video_model_and_diffusion_defaults().keys()
    model, _ = create_video_model_and_diffusion(**{k: cfg[k] for k in keys})
    return model.param_specs()
