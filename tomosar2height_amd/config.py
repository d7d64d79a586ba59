"""Configuration values of the reference's Hydra tree (conf/config.yaml, conf/model/tomosar2height.yaml,
conf/dataset/{base,berlin,munich}.yaml) as plain attribute dictionaries -- Hydra/OmegaConf are not part of the
hot path.  ``TomoSAR2Height(cfg)`` only needs item + attribute access (model.py:18-41)."""
import copy


class AttrDict(dict):
    """dict whose keys are also attributes, recursively (what cfg.use_cloud / cfg['model'] expect)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


_MODEL = dict(                                   # conf/model/tomosar2height.yaml:3-30
    name="tomosar2height",
    encoder="pointnet_local_pool",
    encoder_kwargs=dict(hidden_dim=32, feature_dim=32, plane_resolution=256, scatter_type="max", unet_type="alto",
                        unet_kwargs=dict(depth=5, merge_mode="concat", start_filts=32)),
    encoder2="unet",
    encoder2_kwargs=dict(num_classes=32, in_channels=3, depth=6, merge_mode="concat", start_filts=32),
    decoder_pixel_kwargs=dict(mode="conv", use_footprint=False, hidden_dim=32, out_dim=1, sample_mode="bilinear",
                              leaky=False),
    data_dim=3,
)

_TRAINING = dict(                                # conf/model/tomosar2height.yaml:32-61
    batch_size=1, max_iteration=10000, optimize_every=64, learning_rate=1e-4, weight_ce=10.0,
    scheduler=dict(type="CyclicLR", kwargs=dict(base_lr=1e-4, max_lr=5e-4, mode="triangular2", gamma=1.0,
                                                step_size_up=500, step_size_down=500, cycle_momentum=False)),
)


def _base(name, z_bound, depth, use_footprint, use_image):
    cfg = AttrDict(
        use_cloud=True, use_image=use_image, use_footprint=use_footprint, gpu_id=0,
        dataloader=dict(n_workers=8),            # conf/config.yaml:20-21
        model=copy.deepcopy(_MODEL), training=copy.deepcopy(_TRAINING),
        test=dict(threshold=0.5),
        dataset=dict(name=name, patch_size=[512, 512], normalize=dict(z_bound=list(z_bound))),
    )
    cfg.model.encoder_kwargs.unet_kwargs.depth = depth
    cfg.model.decoder_pixel_kwargs.use_footprint = use_footprint   # ${use_footprint} interpolation
    return cfg


def berlin_config(use_image: bool = False) -> AttrDict:
    """conf/dataset/berlin.yaml: ALTO depth 5, no footprint head, z_bound [-33.7, 156.5] (z_scale 190.2)."""
    return _base("berlin", (-33.7, 156.5), depth=5, use_footprint=False, use_image=use_image)


def munich_config(use_image: bool = True) -> AttrDict:
    """conf/dataset/munich.yaml: ALTO depth 6, footprint head, z_bound [465.5, 599.5] (z_scale 134.0)."""
    return _base("munich", (465.5, 599.5), depth=6, use_footprint=True, use_image=use_image)
