"""Stereo `ImageFolder` import surface (reference compressai/datasets/utils.py:30-404).

Host-side data loading is outside the accelerated path: synthetic
inputs (masic_amd/synth.py) stand in for it here.  The class exists so that the unchanged
drivers' `from compressai.datasets import ImageFolder` resolves; constructing it needs cv2,
which this image does not ship."""


class ImageFolder:
    def __init__(self, *args, **kwargs):
        raise ImportError("compressai.datasets.ImageFolder needs cv2/torchvision for PNG decoding; "
                          "use masic_amd.synth for synthetic stereo batches")
