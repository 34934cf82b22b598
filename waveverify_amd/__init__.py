"""waveverify_amd — MI355X-native embed/detect hot path of WaveVerify.

`from waveverify_amd import WaveVerify, WatermarkID` mirrors
`from waveverify import WaveVerify, WatermarkID` (/root/reference/waveverify/__init__.py:11-14).
"""
from .watermark_id import WatermarkID

__all__ = ["WaveVerify", "WatermarkID"]
__version__ = "0.1.0"


def __getattr__(name):
    if name == "WaveVerify":           # lazy: importing the package must not need torch / a GPU
        from .core import WaveVerify
        return WaveVerify
    raise AttributeError(name)
