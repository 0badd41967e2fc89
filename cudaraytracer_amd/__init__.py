"""MI355X-native renderer core for the CudaRaytracer scene model.

The compute path is libacr_hip.so (hand-written HIP kernels behind the C ABI of
include/acr_hip.h); the C++ host side (scene model, BIH build, loaders,
Renderer front-end) is libacr_host.so.  This package only binds them."""
from .api import HipCore, HipGroup, Renderer, Scene  # noqa: F401
from .scene_io import ImportData, load_blend  # noqa: F401
