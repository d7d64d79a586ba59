"""Encoder registry with the reference's keys (tomosar2height/encoder/__init__.py:3-8).  The two encoders no
reference config selects by default (``pointnet_plus_plus``, ``hourglass``) are out of the hot-path scope
 and raise a clear error instead of silently missing."""
from . import alto, pointnet, unet


class _NotBuilt:
    def __init__(self, name):
        self.name = name

    def __call__(self, *args, **kwargs):
        raise NotImplementedError(f"encoder '{self.name}' is outside the MI355X hot-path scope (SURVEY.md section 8)")


encoder_dict = {
    "pointnet_local_pool": pointnet.LocalPoolPointnet,
    "pointnet_plus_plus": _NotBuilt("pointnet_plus_plus"),
    "hourglass": _NotBuilt("hourglass"),
    "unet": unet.UNet,
}
