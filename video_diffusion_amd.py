"""Import shim: the package directory is `video-diffusion_amd/` (a hyphen cannot
be imported), so `import video_diffusion_amd` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-diffusion_amd")
_spec = importlib.util.spec_from_file_location("video_diffusion_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["video_diffusion_amd"] = _mod
_spec.loader.exec_module(_mod)
