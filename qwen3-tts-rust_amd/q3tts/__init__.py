"""MI355X-native Qwen3-TTS inference path: Python host mirror over the C ABI (include/q3tts.h)."""
from . import _abi  # noqa: F401
from ._abi import Q3Error, default_config, full_config_py, load_library, tiny_config  # noqa: F401
