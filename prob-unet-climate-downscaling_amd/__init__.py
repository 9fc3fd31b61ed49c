"""MI355X-native Probabilistic U-Net engine (ELBO forward+backward, prior sampling) behind the reference's
ProbabilisticUNet interface.  Compute lives in libprobunet.so (HIP, gfx950); this package is the ctypes host side."""
from . import _lib  # noqa: F401
from .prob_unet import ProbabilisticUNet, FlatAdamW  # noqa: F401,E402
from . import dp  # noqa: F401,E402
from . import trainer, data  # noqa: F401,E402
