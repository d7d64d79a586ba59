from .resnet import ResnetBlockFC

__all__ = ["ResnetBlockFC"]
