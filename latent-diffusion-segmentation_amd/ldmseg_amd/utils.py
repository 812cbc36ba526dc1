"""Return-type contract of the reference call surface.

``OutputDict`` mirrors /root/reference/ldmseg/utils/utils.py:26-31: an ordered
dict whose items are also attributes (``out.sample`` == ``out['sample']``).
"""
from collections import OrderedDict


class OutputDict(OrderedDict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in OrderedDict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        super().__setattr__(key, value)


class UNetOutput(OutputDict):           # unet.py:20-21
    pass


class DDIMNoiseSchedulerOutput(OutputDict):   # ddim_scheduler.py:21-23
    pass


class EncoderOutput(OutputDict):        # vae.py:31-33
    pass


class VAEOutput(OutputDict):            # vae.py:27-29
    pass
