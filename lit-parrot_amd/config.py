"""Model hyper-parameters for the decode path.

Same field names as the reference's ``Config`` (lit_gpt/config.py:11-92), so a ``lit_config.json`` written by the
reference loads unchanged with ``Config(**json.load(fp))``.  Only the numbers the hot path needs are kept; the
table holds the model families BASELINE.json names (Pythia, StableLM, Llama-2, Falcon) plus tiny shapes of each
family for tests.
"""
from dataclasses import dataclass, asdict
from typing import Any, Dict, Optional


def find_multiple(n: int, k: int) -> int:
    """Smallest multiple of k that is >= n (lit_gpt/utils.py:19-23)."""
    assert k > 0
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class Config:
    org: str = "Lightning-AI"
    name: str = "lit-GPT"
    block_size: int = 4096
    vocab_size: int = 50254
    padding_multiple: int = 512
    padded_vocab_size: Optional[int] = None
    n_layer: int = 16
    n_head: int = 32
    n_embd: int = 4096
    rotary_percentage: float = 0.25
    parallel_residual: bool = True
    bias: bool = True
    n_query_groups: Optional[int] = None  # n_head: MHA, 1: MQA, in between: GQA
    shared_attention_norm: bool = False
    _norm_class: str = "LayerNorm"  # or "RMSNorm"
    norm_eps: float = 1e-5
    _mlp_class: str = "GptNeoxMLP"  # or "LLaMAMLP"
    intermediate_size: Optional[int] = None
    condense_ratio: int = 1

    def __post_init__(self) -> None:
        if self.n_embd % self.n_head:
            raise ValueError(f"n_embd={self.n_embd} is not divisible by n_head={self.n_head}")
        if self.padded_vocab_size is None:
            self.padded_vocab_size = find_multiple(self.vocab_size, self.padding_multiple)
        if self.n_query_groups is None:
            self.n_query_groups = self.n_head
        elif self.n_head % self.n_query_groups:
            raise ValueError("n_head must be a multiple of n_query_groups")
        if self.intermediate_size is None:
            if self._mlp_class == "LLaMAMLP":
                raise ValueError("LLaMAMLP configs must set intermediate_size")
            self.intermediate_size = 4 * self.n_embd
        if self._norm_class not in ("LayerNorm", "RMSNorm"):
            raise ValueError(f"unknown norm class {self._norm_class}")
        if self._mlp_class not in ("GptNeoxMLP", "LLaMAMLP"):
            raise ValueError(f"unknown mlp class {self._mlp_class}")

    # ------------------------------------------------------------------ derived sizes
    @property
    def head_size(self) -> int:
        return self.n_embd // self.n_head

    @property
    def q_per_kv(self) -> int:
        return self.n_head // self.n_query_groups

    @property
    def rope_n_elem(self) -> int:
        return int(self.rotary_percentage * self.head_size)

    @property
    def qkv_size(self) -> int:
        return (self.n_head + 2 * self.n_query_groups) * self.head_size

    @classmethod
    def from_name(cls, name: str, **overrides: Any) -> "Config":
        fields = dict(name_to_config[name])
        fields.update(overrides)
        return cls(**fields)

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    def linear_shapes(self) -> Dict[str, tuple]:
        """(out_features, in_features) of every Linear in one block, plus lm_head (for byte accounting)."""
        shapes = {"attn.attn": (self.qkv_size, self.n_embd), "attn.proj": (self.n_embd, self.n_embd)}
        if self._mlp_class == "LLaMAMLP":
            shapes["mlp.fc_1"] = (self.intermediate_size, self.n_embd)
            shapes["mlp.fc_2"] = (self.intermediate_size, self.n_embd)
        else:
            shapes["mlp.fc"] = (self.intermediate_size, self.n_embd)
        shapes["mlp.proj"] = (self.n_embd, self.intermediate_size)
        shapes["lm_head"] = (self.padded_vocab_size, self.n_embd)
        return shapes

    def n_linear_params(self) -> int:
        per_block = sum(o * i for name, (o, i) in self.linear_shapes().items() if name != "lm_head")
        return per_block * self.n_layer + self.padded_vocab_size * self.n_embd


def _neox(org: str, name: str, **kw: Any) -> Dict[str, Any]:
    return dict(org=org, name=name, **kw)


def _llama2(name: str, n_layer: int, n_head: int, n_embd: int, inter: int, **kw: Any) -> Dict[str, Any]:
    return dict(org="meta-llama", name=name, block_size=4096, vocab_size=32000, padding_multiple=64, n_layer=n_layer,
                n_head=n_head, n_embd=n_embd, rotary_percentage=1.0, parallel_residual=False, bias=False,
                _norm_class="RMSNorm", norm_eps=1e-5, _mlp_class="LLaMAMLP", intermediate_size=inter, **kw)


def _falcon(name: str, n_layer: int, n_head: int, n_embd: int, groups: int, **kw: Any) -> Dict[str, Any]:
    return dict(org="tiiuae", name=name, block_size=2048, padded_vocab_size=65024, n_layer=n_layer, n_head=n_head,
                n_embd=n_embd, rotary_percentage=1.0, parallel_residual=True, n_query_groups=groups, bias=False, **kw)


_TABLE = [
    # reference lit_gpt/config.py:100, :116-118
    _neox("stabilityai", "stablelm-base-alpha-3b", padding_multiple=512),
    _neox("EleutherAI", "pythia-70m", block_size=2048, n_layer=6, n_embd=512, n_head=8, padding_multiple=128),
    _neox("EleutherAI", "pythia-160m", block_size=2048, n_layer=12, n_embd=768, n_head=12, padding_multiple=128),
    _neox("EleutherAI", "pythia-410m", block_size=2048, n_layer=24, n_embd=1024, n_head=16, padding_multiple=128),
    # reference lit_gpt/config.py:439-455 and the 13b/70b siblings
    _llama2("Llama-2-7b-hf", 32, 32, 4096, 11008),
    _llama2("Llama-2-7b-chat-hf", 32, 32, 4096, 11008),
    _llama2("Llama-2-13b-hf", 40, 40, 5120, 13824),
    _llama2("Llama-2-70b-hf", 80, 64, 8192, 28672, n_query_groups=8),
    # reference lit_gpt/config.py:201-230
    _falcon("falcon-7b", 32, 71, 4544, 1, shared_attention_norm=True),
    _falcon("falcon-40b", 60, 128, 8192, 8),
    _falcon("falcon-40b-instruct", 60, 128, 8192, 8),
    # tiny shapes of each family (tests and smoke); head sizes the attention kernel is built for
    _neox("test", "tiny-neox", block_size=128, vocab_size=500, padding_multiple=64, n_layer=2, n_head=4, n_embd=128),
    _neox("test", "tiny-neox-hs64", block_size=128, vocab_size=500, padding_multiple=64, n_layer=2, n_head=4, n_embd=256),
    _neox("test", "tiny-neox-hs128", block_size=128, vocab_size=500, padding_multiple=64, n_layer=2, n_head=3, n_embd=384),
    dict(_llama2("tiny-llama", 2, 2, 128, 352), org="test", block_size=128, vocab_size=500),
    dict(_llama2("tiny-llama-gqa", 2, 4, 256, 352, n_query_groups=2), org="test", block_size=128, vocab_size=500),
    dict(_llama2("tiny-llama-hs128", 2, 2, 256, 416), org="test", block_size=128, vocab_size=500),
    dict(_falcon("tiny-falcon-gqa", 2, 8, 256, 2), org="test", block_size=128, vocab_size=512, padded_vocab_size=512),
    dict(_falcon("tiny-falcon-mqa", 2, 4, 128, 1, shared_attention_norm=True), org="test", block_size=128,
         vocab_size=512, padded_vocab_size=512),
    # Falcon-7B's odd shapes in small: width not a multiple of 128 (4544 = 71 x 64), an odd number of query heads on one K/V head
    dict(_falcon("tiny-falcon-40b", 2, 8, 512, 2), org="test", block_size=128, vocab_size=512, padded_vocab_size=512),
    dict(_falcon("tiny-falcon-7b", 2, 7, 448, 1, shared_attention_norm=True), org="test", block_size=128,
         vocab_size=512, padded_vocab_size=512),
]

name_to_config: Dict[str, Dict[str, Any]] = {c["name"]: c for c in _TABLE}
