"""The streaming chat generator (lit_parrot_amd/chat/base.py, reference chat/base.py:20-95) on the GPU: its stream of
yielded items must be exactly what the reference's generator logic (oracle/chat.py, pinned against the reference's own
outputs in tests/test_oracle_golden.py) yields for the same sequence of sampled tokens."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd.chat import base as chat  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.synth import synthetic_prompt, synthetic_state_dict  # noqa: E402
from oracle import chat as oc  # noqa: E402

DEV = torch.device("cuda", 0)
BF = torch.bfloat16
T, MAXR = 6, 40


class Scripted:
    """Stands in for the model inside the oracle's generator: always 'predicts' the next token of a given continuation."""

    def __init__(self, continuation, vocab, T):
        self.c, self.V, self.T = continuation, vocab, T

    def __call__(self, x, max_seq_length, input_pos):
        nxt = self.c[int(input_pos[-1]) + 1 - self.T]
        out = torch.zeros((1, x.shape[1], self.V))
        out[0, -1, nxt] = 1.0
        return out


def items(stream):
    return [[int(v) for v in y.reshape(-1).tolist()] for y in stream]


@pytest.fixture(scope="module")
def tiny():
    cfg = Config.from_name("tiny-llama")
    model = L.GPT(cfg)
    model.load_state_dict(synthetic_state_dict(cfg, 4321, perturb=True))
    model = model.to(BF).to(DEV).eval()
    prompt = synthetic_prompt(cfg, T, 4321)
    free = items(chat.generate(model, prompt.to(DEV), MAXR, MAXR, top_k=1))
    assert all(len(i) == 1 for i in free) and len(free) == MAXR - T  # no stop sequence: one token per yield, all of them
    return cfg, model, prompt, [i[0] for i in free]


def expected(cfg, prompt, free, stops):
    return items(oc.generate(Scripted(free, cfg.padded_vocab_size, T), prompt, MAXR, MAXR, stop_tokens=stops, greedy_ties_lowest=True))


def cases(free):
    return {
        "single": ([free[9]],),
        "pair": ([free[12], free[13]],),
        "pair_and_long": ([free[20], free[21]], [1, 2, 3, 4]),
        "early": ([free[0], free[1]], [free[5], free[6], free[7]]),
        "never": ([499, 498, 497],),
        "first_token": ([free[0]],),
        "last_token": ([free[-1]], [497, 498]),
        "chunk_edge": ([free[7], free[8]],),      # hit on the last step of a CHUNK of graph replays
        "long": ([free[i] for i in range(3, 12)],),  # a 9-token stop sequence: longer than a chunk
    }


@pytest.mark.parametrize("name", ["single", "pair", "pair_and_long", "early", "never", "first_token", "last_token", "chunk_edge", "long"])
def test_greedy_stream_equals_the_reference_logic(tiny, name):
    cfg, model, prompt, free = tiny
    stops = cases(free)[name]
    model.reset_cache()
    got = items(chat.generate(model, prompt.to(DEV), MAXR, MAXR, top_k=1, stop_tokens=stops))
    assert got == expected(cfg, prompt, free, stops), name


def test_yielded_tensors_live_on_the_device_and_sessions_are_reused(tiny):
    cfg, model, prompt, free = tiny
    model.reset_cache()
    ys = list(chat.generate(model, prompt.to(DEV), MAXR, MAXR, top_k=1, stop_tokens=([free[20], free[21]], [1, 2, 3, 4])))
    assert all(y.is_cuda for y in ys) and ys[0].dim() == 0 and ys[-1].dim() == 1 and ys[0].dtype == prompt.dtype
    n = len(model._chat_sessions)
    model.reset_cache()
    list(chat.generate(model, prompt.to(DEV), MAXR, MAXR, top_k=1, stop_tokens=([free[3], free[4]], [9, 9, 9, 9])))
    assert len(model._chat_sessions) == n  # same (window, greedy, #sequences, longest): the captured graph is reused


def test_sampled_stream_is_reproducible_and_stops(tiny):
    cfg, model, prompt, free = tiny
    runs = []
    for _ in range(2):
        model.reset_cache()
        torch.manual_seed(7)
        runs.append(items(chat.generate(model, prompt.to(DEV), MAXR, MAXR, temperature=0.9, top_k=5)))
    assert runs[0] == runs[1] and len(runs[0]) == MAXR - T
    toks = [i[0] for i in runs[0]]
    stops = ([toks[10], toks[11]],)
    model.reset_cache()
    torch.manual_seed(7)
    got = items(chat.generate(model, prompt.to(DEV), MAXR, MAXR, temperature=0.9, top_k=5, stop_tokens=stops))
    assert got == expected(cfg, prompt, toks, stops)


def test_limits_fail_loudly(tiny):
    cfg, model, prompt, free = tiny
    with pytest.raises(L.ParrotHipError, match="stop sequences"):
        list(chat.generate(model, prompt.to(DEV), MAXR, MAXR, top_k=1, stop_tokens=tuple([1, 2] for _ in range(40))))
