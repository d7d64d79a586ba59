"""TomoSAR2Height network wiring (reference: tomosar2height/model.py:9-86): point-cloud encoder (+ optional
image encoder) -> pixelwise decoder -> heights in metres.  Same constructor, forward signature and
``state_dict`` keys (``point_encoder.*``, ``image_encoder.*``, ``decoder.*``) as the reference, so
``trainer.py`` / ``generator.py``-style callers and reference checkpoints work unchanged."""
from typing import Dict

import torch
import torch.nn as nn

from .decoder import decoder_dict
from .encoder import encoder_dict


class TomoSAR2Height(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        cfg_model = cfg["model"]
        self.dim = cfg_model["data_dim"]
        self.use_cloud = cfg.use_cloud
        self.use_image = cfg.use_image
        if self.use_cloud:
            self.point_encoder = encoder_dict[cfg_model["encoder"]](dim=self.dim, **cfg_model["encoder_kwargs"])
        if self.use_image:
            self.image_encoder = encoder_dict[cfg_model.get("encoder2")](**cfg_model.get("encoder2_kwargs", {}))
        self.decoder = decoder_dict["pixel"](**cfg_model["decoder_pixel_kwargs"])
        self.threshold = cfg["test"]["threshold"]
        z_bound = cfg["dataset"]["normalize"]["z_bound"]
        self.z_scale = z_bound[1] - z_bound[0]
        self._initialize_weights()
        self.set_channels_last(True)      # default: grid side in NHWC on the implicit-GEMM convolution kernels

    def _initialize_weights(self):
        """model.py:46-52: Xavier-uniform weights / zero biases on every Conv2d and Linear (ConvTranspose2d is
        not a Conv2d subclass and keeps torch's default init)."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def set_channels_last(self, flag: bool = True):
        """Run the grid side (planes + convs) in channels_last memory (the default): the point<->grid HIP kernels read
        and write pixel-major rows natively, so no layout copies remain, and the convolutions run on csrc/conv.hip.
        ``set_channels_last(False)`` restores the NCHW / MIOpen grid side; results agree to rounding."""
        if self.use_cloud and hasattr(self.point_encoder, "set_channels_last"):
            self.point_encoder.set_channels_last(flag)
        if self.use_image and hasattr(self.image_encoder, "set_channels_last"):
            self.image_encoder.set_channels_last(flag)
        if hasattr(self.decoder, "set_channels_last"):
            self.decoder.set_channels_last(flag)
        self._channels_last = bool(flag)
        fmt = torch.channels_last if flag else torch.contiguous_format
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                m.weight.data = m.weight.data.contiguous(memory_format=fmt)
        return self

    def out_of_domain_total(self, reset: bool = True) -> int:
        """Running count of input points outside [0, 1)^2 seen by the point encoder (one device sync)."""
        enc = getattr(self, "point_encoder", None)
        return enc.out_of_domain_total(reset) if hasattr(enc, "out_of_domain_total") else 0

    @staticmethod
    def set_mlp_precision(name: str):
        """'fp32' (default) or 'bf16' (BASELINE configs[2]) for the per-point MLP GEMMs (process-wide; see ``mlp.set_precision``)
        AND the 3x3 grid convolutions: in 'bf16' mode they run on the same bf16 matrix-core kernels with the operands rounded to
        bf16 once instead of split exactly (``grid.set_conv_precision``) -- one MFMA per product, fp32 accumulate, fp32 tensors."""
        from . import grid, mlp
        mlp.set_precision(name)
        grid.set_conv_precision("bf16" if name == "bf16" else None)

    def forward(self, input_cloud=None, input_image=None):
        assert self.use_image or self.use_cloud, "At least one input modality must be used."
        if not self._channels_last:
            from . import _lib
            _lib.library_fallback("NCHW grid side: convolutions on MIOpen (set_channels_last(False))")
        feature_planes = self.encode_inputs(input_cloud, input_image)
        pa, pb = self.decoder(feature_planes)
        return pa * self.z_scale, pb

    def encode_inputs(self, input_cloud=None, input_image=None) -> Dict[str, torch.Tensor]:
        feature_planes = {}
        if self.use_cloud:
            feature_planes.update(self.point_encoder(input_cloud))
        if self.use_image:
            feature_planes["image"] = self.image_encoder(input_image)
        return feature_planes
