from . import pixel

decoder_dict = {
    "pixel": pixel.PixelwiseDecoder,
}
