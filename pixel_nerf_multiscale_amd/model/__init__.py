from .code import PositionalEncoding
from .encoder import ImageEncoder, SpatialEncoder
from .models import PixelNeRFNet, make_encoder, make_mlp
from .resnetfc import ResnetBlockFC, ResnetFC
from ..util import as_conf


def make_model(conf, *args, **kwargs):
    conf = as_conf(conf)
    model_type = conf.get_string("type", "pixelnerf")
    if model_type == "pixelnerf":
        return PixelNeRFNet(conf, *args, **kwargs)
    raise NotImplementedError("Unsupported model type", model_type)
