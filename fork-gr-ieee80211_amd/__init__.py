"""MI355X-native 802.11 OFDM RX hot path (signal -> demod/demod2 -> decode).

Import with importlib.import_module("fork-gr-ieee80211_amd") (the directory name is not a Python
identifier).  `rx` binds the HIP library through its C-ABI; `txgen` builds synthetic corpora.
"""
from . import rx, txgen  # noqa: F401
from .rx import RxContext, load_library  # noqa: F401
