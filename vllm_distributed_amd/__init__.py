"""MI355X-native distributed LLM serving framework.

Public API mirrors what a user of the reference reaches through vLLM:
``LLM``/``SamplingParams`` for offline use, ``launch.main`` for the
``serve`` / ``remote`` roles, ``EngineArgs`` for configuration.
"""
from .version import __version__

__all__ = ["__version__", "LLM", "SamplingParams", "BeamSearchParams", "EngineArgs", "RequestOutput"]


def __getattr__(name):
    if name == "LLM":
        from .engine.llm_engine import LLM
        return LLM
    if name == "SamplingParams":
        from .sampling_params import SamplingParams
        return SamplingParams
    if name == "BeamSearchParams":
        from .engine.beam_search import BeamSearchParams
        return BeamSearchParams
    if name == "EngineArgs":
        from .engine.arg_utils import EngineArgs
        return EngineArgs
    if name == "RequestOutput":
        from .outputs import RequestOutput
        return RequestOutput
    raise AttributeError(name)
