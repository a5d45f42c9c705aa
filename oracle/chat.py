"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's streaming chat generator (chat/base.py:20-95).

Only tests/ may import this.  Pinned against tests/golden/chat.npz (outputs of the reference's own
``chat.base.generate`` run in the build container, tests/golden/make_golden.py::golden_chat).

The reference keeps the not-yet-yielded tokens in a buffer as long as the longest stop sequence (chat/base.py:48-50),
writes each new token at ``min(t, L-1)`` (:78), compares the END of the buffer with every stop sequence (:81-87) and, once
the buffer is full, yields its oldest token and rolls (:88-94).  Consequences that this restatement keeps:
  * a stop sequence shorter than the buffer cannot match while the buffer is still filling (its tail holds the filler);
  * on a hit, the buffered tokens in front of the stop sequence are yielded as ONE multi-token item (:84-86);
  * when max_returned_tokens is reached, the last L-1 generated tokens are never yielded.
"""
from typing import Iterator, List, Optional, Sequence

import torch
import torch.nn.functional as F

FILLER = -999  # chat/base.py:50


@torch.no_grad()
def generate(model, idx: torch.Tensor, max_returned_tokens: int, max_seq_length: int, *, temperature: float = 1.0,
             top_k: Optional[int] = None, stop_tokens: Sequence[List[int]] = (),
             greedy_ties_lowest: bool = False) -> Iterator[torch.Tensor]:
    T = idx.size(0)
    assert max_returned_tokens > T
    stops = [torch.tensor(list(s), dtype=torch.long) for s in stop_tokens]
    L = max((len(s) for s in stops), default=1)
    held = torch.full((L,), FILLER, dtype=torch.long)
    input_pos = torch.arange(0, T)
    yielded = 0  # tokens handed out so far (the reference tracks yield_i = yielded - 1)
    cur = idx
    for t in range(max_returned_tokens - T):
        logits = model(cur.view(1, -1), max_seq_length, input_pos)[0, -1] / temperature
        if greedy_ties_lowest:
            nxt = torch.argmax(logits.float(), dim=-1, keepdim=True)
        else:
            if top_k is not None:
                v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
                logits = torch.where(logits < v[[-1]], -float("Inf"), logits)
            nxt = torch.multinomial(F.softmax(logits, dim=-1), num_samples=1)
        cur = nxt
        input_pos = input_pos[-1:] + 1
        held[min(t, L - 1)] = nxt
        for s in stops:
            n = len(s)
            if torch.equal(held[L - n:], s):
                if L > n:
                    yield held[: L - n].clone()
                return
        if t + 1 - yielded >= L:
            yield held[0].clone()
            held = torch.roll(held, -1, 0)
            yielded += 1
