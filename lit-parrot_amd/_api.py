"""Public names of the package (what ``import lit_parrot_amd`` exposes)."""
from ._hip import ParrotHipError
from .config import Config, name_to_config
from .generate.base import generate
from .model import GPT, Block, CausalSelfAttention, GptNeoxMLP, LLaMAMLP, apply_rope, build_rope_cache
from .rmsnorm import RMSNorm
from .utils import quantization

__all__ = [
    "GPT", "Block", "CausalSelfAttention", "GptNeoxMLP", "LLaMAMLP", "RMSNorm", "Config", "name_to_config",
    "generate", "quantization", "apply_rope", "build_rope_cache", "ParrotHipError",
]
