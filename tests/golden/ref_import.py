"""Import the reference's Python modules from /root/reference (build container only).

The reference's ``utils/__init__.py`` pulls in open3d / laspy / rasterio and the
encoders import ``torch_scatter`` -- none of which is installed.  Empty stub
modules stand in for the IO packages (nothing on the hot path touches them) and
``oracle.scatter_ref`` stands in for ``torch_scatter``.  Nothing is copied: the
reference is imported where it lies.  Never used on the GPU box.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "tomosar2height"))


def import_reference():
    """Returns the imported ``tomosar2height`` package of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    repo_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo_root not in sys.path:
        sys.path.insert(0, repo_root)
    from oracle import scatter_ref

    class _AnyMeta(type):
        """Class whose every attribute is again such a class (``o3d.geometry.PointCloud`` ...)."""

        def __getattr__(cls, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return _AnyMeta(item, (), {})

    def stub(name, **attrs):
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)

        def _module_getattr(item):  # PEP 562; dunders must stay missing (inspect.getmodule probes them)
            if item.startswith("__"):
                raise AttributeError(item)
            return _AnyMeta(item, (), {})

        mod.__getattr__ = _module_getattr
        sys.modules.setdefault(name, mod)
        return sys.modules[name]

    stub("open3d")
    stub("laspy")
    rio = stub("rasterio")
    rio.transform = stub("rasterio.transform")
    stub("torch_scatter", scatter_max=scatter_ref.scatter_max, scatter_mean=scatter_ref.scatter_mean)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import tomosar2height  # noqa: F401  (the reference package)
    return tomosar2height


class Cfg(dict):
    """dict with attribute access -- what ``TomoSAR2Height(cfg)`` needs (model.py:18-41)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def make_cfg(depth=5, reso=256, hidden=32, use_image=False, use_footprint=False, z_bound=(-33.7, 156.5),
             mode="conv", image_depth=6, image_filts=32, start_filts=32):
    """Config values of conf/model/tomosar2height.yaml + conf/dataset/{berlin,munich}.yaml."""
    return Cfg(
        use_cloud=True, use_image=use_image,
        model=Cfg(
            encoder="pointnet_local_pool",
            encoder_kwargs=dict(hidden_dim=hidden, feature_dim=hidden, plane_resolution=reso,
                                scatter_type="max", unet_type="alto",
                                unet_kwargs=dict(depth=depth, merge_mode="concat", start_filts=start_filts)),
            encoder2="unet",
            encoder2_kwargs=dict(num_classes=hidden, in_channels=3, depth=image_depth, merge_mode="concat",
                                 start_filts=image_filts),
            decoder_pixel_kwargs=dict(mode=mode, use_footprint=use_footprint, hidden_dim=hidden, out_dim=1,
                                      sample_mode="bilinear", leaky=False),
            data_dim=3),
        test=Cfg(threshold=0.5),
        dataset=Cfg(normalize=Cfg(z_bound=list(z_bound))),
    )
