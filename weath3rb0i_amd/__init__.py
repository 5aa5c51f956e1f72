"""weath3rb0i_amd — MI355X-native (gfx950) implementation of the weath3rb0i hot
path: per-bit context-model prediction + binary arithmetic coding, one
wavefront lane per independent block.  The compute lives in libw3hip.so (HIP);
this package mirrors the reference crate's model/compress surface on top of
its C ABI (include/w3hip.h)."""
from .api import (MAGIC_STR, Context, encode_blocks_sharded_device, encode_sharded_submit, encode_sharded_wait,  # noqa: F401
                  sharded_max_in_flight)
from .models import (APM, ACHistory, ACHistoryCached, AdaptiveModel, BestOfTwoModel, FrozenModel, HashMap, HuffHistory, Mixer, Model,  # noqa: F401
                     OpinionMixer2, Order0, Order1, OrderN, OrderNEntropy, RawHistory, SlotModel, StateTable,
                     StationaryModel, W3Error, full_cm, init_model)
