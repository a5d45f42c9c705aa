"""The decode loop: ``generate()`` with the reference's signature (generate/base.py:92-159), hipGraph-captured.

Reference loop per token: index_select the new token, ``model(x, max_seq_length, input_pos)``, divide by the
temperature, optional top-k crop, softmax, multinomial, out-of-place ``index_copy`` into the token buffer and, when
``eos_id`` is given, a host sync to compare (:131-157).  Here the loop state lives on the device:

    tokens  int64[max_returned_tokens]   the prompt, then every sampled token
    pos     int32[1]                     position of the row being decoded

and one decode step — embedding of ``tokens[pos]``, every block, final norm, lm_head and, for greedy decoding, the
arg-max that writes ``tokens[pos+1]`` and advances ``pos`` — is a fixed launch sequence captured ONCE per
(model, max_seq_length, sampling parameters) in a hipGraph and replayed per token.  Greedy decoding (``top_k == 1``, which
is how the reference spells it: there is no argmax branch) never touches the host inside the loop unless ``eos_id`` is set.
With ``top_k != 1`` (the reference's default call: temperature 0.8, top_k 200) the sampling step is inside the graph too:
torch draws the Exponential(1) noise that ``torch.multinomial`` draws internally (``noise.exponential_(1)``, graph-safe
Philox: the same generator consumption) and ``parrot_topk_sample`` does the rest - temperature, top-k crop, softmax,
arg-max of probs / noise - with the arithmetic of the torch device ops the reference runs (:139-144), so the same torch seed
draws the same tokens as those ops do on this device (tests/test_model_gpu.py::test_sampling_in_the_graph_draws_the_tokens_of_the_torch_ops).

The prompt is prefilled in one multi-row pass (same kernels, M = T rows) that computes only the last row of lm_head
(the reference computes all T rows and discards T-1 of them, :135-136).
"""
from typing import Dict, Optional, Tuple

import torch

from .. import ops
from .._hip import ParrotHipError
from ..model import GPT


class DecodeSession:
    """Static buffers + captured graph of the single-token step for one (model, max_seq_length, greedy) choice."""

    def __init__(self, model: GPT, max_seq_length: int, max_tokens: int, greedy: bool, use_graph: bool = True,
                 engine: Optional[bool] = None, sampler: Optional[Tuple[float, Optional[int]]] = None) -> None:
        self.model, self.S, self.greedy = model, max_seq_length, greedy
        # (temperature, top_k) of the in-graph sampling step when not greedy; None: the step ends at the logits
        self.sampler = None if greedy else sampler
        self.noise: Optional[torch.Tensor] = None
        dev = model.transformer.wte.weight.device
        if dev.type != "cuda":
            raise ParrotHipError("generate() runs on the HIP device only: move the model to cuda (no CPU fallback)")
        self.device = dev
        self.tokens = torch.zeros((max_tokens + 1,), dtype=torch.int64, device=dev)
        self.pos = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.ws = model.workspace(1, dev, 1)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.use_graph = use_graph
        if model.rope_cache is None:
            model.rope_cache = model.build_rope_cache(self.tokens)
        if not model.kv_caches or model.kv_caches[0][0].size(2) != max_seq_length or model.kv_caches[0][0].size(0) != 1:
            model.kv_caches = model.build_kv_caches(self.tokens.view(1, -1), max_seq_length, model.rope_cache[0].size(-1))
        self.caches = [(k[0], v[0]) for k, v in model.kv_caches]
        # one launch per token when the model fits the stream engine (csrc/engine.hip), else 5 launches per block
        from ..engine import StreamEngine

        self.eng = None
        if engine is None:
            engine = ENGINE_DEFAULT
        if engine == "auto":
            engine = StreamEngine.faster_than_multi_launch(model, max_seq_length, ENGINE_AUTO_MIN_WINDOW)
        if engine and StreamEngine.supported(model) is None:
            self.eng = StreamEngine(model, self.tokens, self.pos, self.caches, max_seq_length, greedy)
        # One resident image of the int4 weights: W4K (the multi-launch step's and every prompt's format; the engine keeps
        # its E4 copy beside it).  The reference-layout buffers are released; state_dict() / get_weight() rebuild them exactly.
        if RELEASE_REFERENCE_BUFFERS:
            for m in model.modules():
                if hasattr(m, "release_reference"):
                    m.release_reference()


    # one decode step = the launch sequence that gets captured
    def _step(self) -> None:
        if self.eng is not None:
            logits = self.eng.step()  # embedding .. lm_head (.. arg-max + advance when greedy) in one launch
        else:
            logits = self.model.run_rows(self.ws, self.tokens, self.pos, self.pos, self.S, self.caches, self.model.rope_cache)
            if self.greedy:
                ops.argmax_advance(logits, self.tokens, self.pos)
        if self.sampler is not None:
            self.sample(logits)

    def sample(self, logits: torch.Tensor) -> None:
        """``tokens[pos + 1]`` = the next token drawn from ``logits`` (the row at ``pos``), ``pos += 1``: arg-max when greedy,
        else the reference's temperature / top-k / softmax / multinomial step (generate/base.py:136-153) as one torch draw of
        the noise + one launch (ops.topk_sample)."""
        if self.greedy:
            ops.argmax_advance(logits, self.tokens, self.pos)
            return
        if self.sampler is None:
            raise ParrotHipError("this session was built without sampling parameters")
        if self.noise is None:
            self.noise = torch.empty((logits.numel(),), dtype=torch.bfloat16, device=self.device)
        self.noise.exponential_(1)  # what torch.multinomial draws internally: same generator consumption
        ops.topk_sample(logits.view(-1), self.sampler[0], self.sampler[1], self.noise, self.tokens, self.pos)

    def prefill(self, prompt: torch.Tensor) -> torch.Tensor:
        """Run the T prompt rows, leave ``pos`` = T-1 and return the logits of the last prompt token."""
        T = prompt.numel()
        self.tokens[:T].copy_(prompt)
        self.pos.zero_()
        ws = self.model.workspace(T, self.device, 1)
        ws.pos_is_zero = True  # the prompt starts at position 0: its attention may take the MFMA kernel (no ring wrap in the call)
        try:
            logits = self.model.run_rows(ws, self.tokens, None, self.pos, self.S, self.caches, self.model.rope_cache)
        finally:
            ws.pos_is_zero = False
        self.pos.fill_(T - 1)
        return logits

    def capture(self) -> None:
        if self.graph is not None or not self.use_graph:
            return
        # warm-up outside capture (lazy W4K repacks, workspace allocations), restoring the loop state afterwards
        saved = (self.tokens.clone(), self.pos.clone(), [(k.clone(), v.clone()) for k, v in self.caches])
        # (a sampling step draws from torch's generator: the warm-up below must not use up draws of the caller's seed)
        rng = torch.cuda.get_rng_state(self.device) if self.sampler is not None else None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._step()
        self.graph = graph
        if rng is not None:
            torch.cuda.set_rng_state(rng, self.device)
        self.tokens.copy_(saved[0])
        self.pos.copy_(saved[1])
        for (k, v), (k0, v0) in zip(self.caches, saved[2]):
            k.copy_(k0)
            v.copy_(v0)

    def step(self) -> torch.Tensor:
        """Decode the token at ``pos``; returns the logits buffer (V,) of that step."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step()
        return self.eng.logits[0] if self.eng is not None else self.ws.logits[0]

    def check_error(self) -> None:
        """Host-side check (syncs) of the executor's in-launch waits: the stream engine's error word.  A step whose waits gave
        up decoded garbage: fail loudly, never report its tokens or its speed."""
        if self.eng is not None:
            self.eng.check_error()


# The one-launch stream engine (engine.py) for the models it is built for (Llama-2 7B family, int4 g128).  Measured on
# Llama-2-7B int4: its token time barely moves with the context (K/V rows stream through the same LDS ring
# as the weights, on all 256 CUs), the multi-launch step's grows; they cross below 1k keys.  With bf16 weights (two loader
# waves, a parallel-residual block with three hand-offs instead of five) it is ahead at every window: StableLM-3B 752 vs
# 654 tokens/s behind a 512-token prompt.  "auto" picks by weight format and window size.
ENGINE_DEFAULT = "auto"
ENGINE_AUTO_MIN_WINDOW = 1024
# A decode session frees the reference-layout copy of every GPTQ int4 Linear (quantize/gptq.py::release_reference): Llama-2-7B
# int4 then holds 3.5 GB of weights instead of 7 (multi-launch step) and Falcon-40B int4 44 GB instead of 66 (engine: W4K + E4).
RELEASE_REFERENCE_BUFFERS = True
MAX_SESSIONS_PER_MODEL = 4  # captured steps kept per model (keyed by window, greedy / sampling parameters), least recently used first out


def _session(model: GPT, max_seq_length: int, max_tokens: int, greedy: bool,
             sampler: Optional[Tuple[float, Optional[int]]] = None) -> DecodeSession:
    cache: Dict[Tuple, DecodeSession] = model.__dict__.setdefault("_decode_sessions", {})
    sampler = None if greedy else sampler
    key = (max_seq_length, greedy, sampler)
    sess = cache.get(key)
    stale = (
        sess is None
        or sess.tokens.numel() < max_tokens + 1
        or not model.kv_caches
        or model.kv_caches[0][0].data_ptr() != sess.caches[0][0].data_ptr()
    )
    if stale:
        cache.pop(key, None)
        sess = None  # (the old session's buffers go before the new one's are allocated)
        while len(cache) >= MAX_SESSIONS_PER_MODEL:  # a caller that varies temperature / top_k per request must not pile up graphs
            cache.pop(next(iter(cache)))
        sess = DecodeSession(model, max_seq_length, max_tokens, greedy, sampler=sampler)
    else:
        cache.pop(key)  # (re-inserted below: the dict's order is the order of last use)
    cache[key] = sess
    return sess


@torch.no_grad()
def generate(
    model: torch.nn.Module,
    idx: torch.Tensor,
    max_returned_tokens: int,
    max_seq_length: int,
    *,
    temperature: float = 1.0,
    top_k: Optional[int] = None,
    eos_id: Optional[int] = None,
) -> torch.Tensor:
    """Continue the prompt ``idx`` (T,) up to ``max_returned_tokens`` tokens; same arguments and return value as the
    reference (generate/base.py:93-159): a 1-D tensor holding the prompt followed by the generated tokens, cut after
    ``eos_id`` when it is produced.  Call ``model.reset_cache()`` between prompts like the reference's ``main``."""
    T = idx.size(0)
    assert max_returned_tokens > T
    if not isinstance(model, GPT):
        raise ParrotHipError("generate() drives lit_parrot_amd.GPT models")
    assert max_seq_length <= model.config.block_size
    if max_returned_tokens > model.config.block_size:
        # positions index the RoPE tables (block_size rows); the reference fails at rope.index_select in this case
        # (lit_gpt/model.py:88) - here the kernels would read past the tables on the device
        raise ParrotHipError(f"max_returned_tokens={max_returned_tokens} exceeds block_size={model.config.block_size}: "
                             "positions past the RoPE tables")
    assert max_seq_length >= T, f"Cannot forward sequence of length {T}, max seq length is only {max_seq_length}"
    dtype = idx.dtype
    greedy = top_k == 1 and temperature > 0
    sess = _session(model, max_seq_length, max_returned_tokens, greedy, (float(temperature), top_k))
    logits = sess.prefill(idx.to(device=sess.device, dtype=torch.int64))
    sess.capture()

    n_new = max_returned_tokens - T
    for i in range(n_new):
        # `logits` belong to the row at `pos` = T-1+i; sample tokens[T+i] from them and advance `pos`: the first token here,
        # the later ones inside the captured step (arg-max, or the noise draw + parrot_topk_sample)
        if i == 0:
            sess.sample(logits)
        if eos_id is not None and int(sess.tokens[T + i]) == eos_id:  # host sync, as in the reference (:156-157)
            # the reference returns idx[:input_pos], which stops BEFORE the eos token despite its comment
            break
        if i + 1 < n_new:
            logits = sess.step()
    else:
        i = n_new
    out = sess.tokens[: T + i].to(dtype).clone()
    sess.check_error()  # a bounded in-launch wait gave up (e.g. fewer CUs than resident workgroups): fail loudly
    return out
