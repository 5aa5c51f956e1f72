"""weath3rb0i_amd — MI355X-native (gfx950) implementation of the weath3rb0i hot
path: per-bit context-model prediction + binary arithmetic coding, one
wavefront lane per independent block.  The compute lives in libw3hip.so (HIP);
this package mirrors the reference crate's model/compress surface on top of
its C ABI (include/w3hip.h)."""
from .api import MAGIC_STR, Context  # noqa: F401
from .models import (ACHistory, AdaptiveModel, BestOfTwoModel, FrozenModel, Model, Order0, Order1, OrderN,  # noqa: F401
                     OrderNEntropy, RawHistory, StationaryModel, W3Error, init_model)
