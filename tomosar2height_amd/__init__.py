"""tomosar2height_amd -- MI355X (gfx950) implementation of the dual-topology point-cloud hot path of
zhu-xlab/tomosar2height behind the reference's own module interface.

    from tomosar2height_amd import TomoSAR2Height, encoder_dict, decoder_dict

The device arithmetic lives in ``libt2h_hip.so`` (C ABI: include/t2h.h, sources: csrc/*.hip); there is no CPU
path in this package.
"""
from .model import TomoSAR2Height
from .decoder import decoder_dict
from .encoder import encoder_dict
from .tile import TileIndex
from ._lib import allow_library_fallback, fallback_counts

__all__ = ["TomoSAR2Height", "decoder_dict", "encoder_dict", "TileIndex", "allow_library_fallback", "fallback_counts"]
