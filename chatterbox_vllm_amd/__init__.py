"""MI355X-native T3 speech-token decode engine behind the ChatterboxTTS API surface.

Layout: ``csrc/`` HIP kernels + the C ABI (``include/t3_hip.h``), ``engine.py`` the
``vllm.LLM`` stand-in, ``tts.py`` / ``mtl_tts.py`` the ``ChatterboxTTS`` facades,
``sharding.py`` request sharding over the GPUs of a node.
"""
from .config import T3ModelConfig  # noqa: F401

__version__ = "0.1.0"
