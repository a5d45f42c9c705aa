"""The streaming chat generator: ``generate()`` with the reference's signature and yield behaviour (chat/base.py:20-95).

The reference keeps the not-yet-yielded tokens in a look-back buffer as long as the longest stop sequence, compares the
end of it with every stop sequence after EVERY token on the host (``torch.equal`` — a device sync per token) and yields the
oldest buffered token once the buffer is full.  Observable behaviour kept here (pinned by tests/golden/chat.npz):

  * nothing matches before ``L`` tokens were generated (L = longest stop sequence);
  * on a hit the buffered tokens in front of the stop sequence come out as ONE multi-token tensor, then the stream ends;
  * without a hit the last ``L - 1`` generated tokens are never yielded.

Here the loop state lives on the device (``DecodeSession``).  For greedy decoding (``top_k == 1``) the captured graph of one
step ends with ``parrot_stop_check``, which latches the first hit in a device flag, and the host replays ``CHUNK`` steps
between two reads of that flag: tokens generated past a hit are simply never yielded.  With sampling (``top_k != 1``) the
loop goes token by token, so that the torch generator is consumed exactly as far as the reference consumes it.
"""
from typing import Iterator, List, Optional, Tuple

import torch

from .. import ops
from .._hip import ParrotHipError
from ..generate.base import DecodeSession
from ..model import GPT

CHUNK = 8          # greedy: graph replays between two reads of the stop flag
MAX_STOP_SEQS = 16
MAX_STOP_TOKENS = 256


class _StopState:
    """Device-side description of the stop sequences of one call + the latch."""

    def __init__(self, device) -> None:
        self.flat = torch.zeros((MAX_STOP_TOKENS,), dtype=torch.int64, device=device)
        self.off = torch.zeros((MAX_STOP_SEQS + 1,), dtype=torch.int32, device=device)
        self.first_gen = torch.zeros((1,), dtype=torch.int32, device=device)
        self.flag = torch.zeros((2,), dtype=torch.int32, device=device)
        self.n = 0
        self.longest = 1

    def arm(self, stop_tokens: Tuple[List[int], ...], first_gen: int) -> None:
        seqs = [list(s) for s in stop_tokens]
        if len(seqs) > MAX_STOP_SEQS or sum(len(s) for s in seqs) > MAX_STOP_TOKENS or any(len(s) == 0 for s in seqs):
            raise ParrotHipError(f"chat.generate: at most {MAX_STOP_SEQS} non-empty stop sequences / {MAX_STOP_TOKENS} tokens")
        flat = [t for s in seqs for t in s]
        off = [0]
        for s in seqs:
            off.append(off[-1] + len(s))
        if flat:
            self.flat[: len(flat)].copy_(torch.tensor(flat, dtype=torch.int64))
        self.off[: len(off)].copy_(torch.tensor(off, dtype=torch.int32))
        self.first_gen.fill_(first_gen)
        self.flag.copy_(torch.tensor([-1, 0], dtype=torch.int32))
        self.n, self.longest = len(seqs), max((len(s) for s in seqs), default=1)


class ChatSession(DecodeSession):
    """DecodeSession whose step - greedy, or sampling inside the graph (generate/base.py) - also runs the device-side stop
    check (captured in the same graph)."""

    def __init__(self, model: GPT, max_seq_length: int, max_tokens: int, greedy: bool, sampler=None) -> None:
        super().__init__(model, max_seq_length, max_tokens, greedy, sampler=sampler)  # the library's choice of executor (generate/base.py)
        self.stop = _StopState(self.device)
        self.n_stop_captured = None

    def check(self) -> None:
        s = self.stop
        ops.stop_check(self.tokens, self.pos, s.first_gen, s.flat, s.off, s.n, s.longest, s.flag)

    def _step(self) -> None:
        super()._step()
        self.check()


def _session(model: GPT, max_seq_length: int, max_tokens: int, greedy: bool, n_stop: int, longest: int, sampler=None) -> ChatSession:
    cache = model.__dict__.setdefault("_chat_sessions", {})
    sampler = None if greedy else sampler
    # the number of sequences and the longest one are launch arguments of the captured stop check
    key = (max_seq_length, greedy, n_stop, longest, sampler)
    sess = cache.get(key)
    stale = (sess is None or sess.tokens.numel() < max_tokens + 1 or not model.kv_caches
             or model.kv_caches[0][0].data_ptr() != sess.caches[0][0].data_ptr())
    if stale:
        cache.pop(key, None)
        sess = None
        from ..generate.base import MAX_SESSIONS_PER_MODEL
        while len(cache) >= MAX_SESSIONS_PER_MODEL:
            cache.pop(next(iter(cache)))
        sess = ChatSession(model, max_seq_length, max_tokens, greedy, sampler)
    else:
        cache.pop(key)
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
    stop_tokens: Tuple[List[int], ...] = (),
) -> Iterator[torch.Tensor]:
    """Continue the prompt ``idx`` (T,) and yield the generated tokens as they become safe to show: 0-dim tensors, plus
    one 1-D tensor with the leftovers in front of a stop sequence when one is hit (chat/base.py:20-95)."""
    T = idx.size(0)
    assert max_returned_tokens > T
    if not isinstance(model, GPT):
        raise ParrotHipError("chat.generate() drives lit_parrot_amd.GPT models")
    assert max_seq_length <= model.config.block_size
    if max_returned_tokens > model.config.block_size:
        # positions index the RoPE tables (block_size rows); the reference fails at rope.index_select in this case
        # (lit_gpt/model.py:88) - here the kernels would read past the tables on the device
        raise ParrotHipError(f"max_returned_tokens={max_returned_tokens} exceeds block_size={model.config.block_size}: "
                             "positions past the RoPE tables")
    assert max_seq_length >= T, f"Cannot forward sequence of length {T}, max seq length is only {max_seq_length}"
    greedy = top_k == 1 and temperature > 0
    n_stop = len(stop_tokens)
    L = max((len(s) for s in stop_tokens), default=1)
    sess = _session(model, max_seq_length, max_returned_tokens, greedy, n_stop, L, (float(temperature), top_k))
    sess.stop.arm(stop_tokens, T)
    dtype = idx.dtype
    n_new = max_returned_tokens - T
    logits = sess.prefill(idx.to(device=sess.device, dtype=torch.int64))
    sess.capture()

    sess.sample(logits)  # generated token 0: arg-max, or the reference's sampling step (later steps sample inside the graph)
    sess.check()
    done = 1        # generated tokens present in sess.tokens[T : T + done]
    emitted = 0     # steps t whose yield decision has been taken
    while True:
        flag = sess.stop.flag.tolist()  # one device read per chunk
        sess.check_error()  # (the time-out words of the in-launch waits: same sync)
        gen = sess.tokens[T: T + done].to(dtype)
        hit_t, hit_n = flag
        upto = done if hit_t < 0 else hit_t + 1
        for t in range(emitted, upto):
            if t == hit_t:
                if L > hit_n:
                    yield gen[t - L + 1: t - hit_n + 1].clone()
                return
            if t >= L - 1:
                yield gen[t - L + 1].clone()
        emitted = upto
        if done >= n_new:
            return
        steps = min(CHUNK, n_new - done)
        for _ in range(steps):
            sess.step()  # model + sampling + stop check: one graph replay
        done += steps
