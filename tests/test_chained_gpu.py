"""The chained single-token step (two streams + in-kernel arrival counters, include/parrot_hip.h "chained launches")
against the plain launch-after-launch step and the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd import ops  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import is_linear_key, synthetic_prompt, synthetic_state_dict  # noqa: E402
from oracle import int4 as o4  # noqa: E402
from oracle import model as om  # noqa: E402

DEV = torch.device("cuda", 0)
BF = torch.bfloat16


@pytest.fixture(autouse=True)
def cpu_rsqrt_mode():
    ops.RMSNORM_RSQRT_MODE = 1  # the CPU-run reference's rsqrt rounding
    yield
    ops.RMSNORM_RSQRT_MODE = 0


def int4_model(name, tile_cols, mode):
    cfg = Config.from_name(name)
    sd = {k: v.to(BF) for k, v in synthetic_state_dict(cfg, 4321, perturb=True).items()}
    qsd = o4.quantize_state_dict(sd, tile_cols, is_linear_key)
    with L.quantization(mode):
        model = L.GPT(cfg)
    model.load_state_dict(qsd, strict=True)
    return cfg, qsd, model.to(BF).to(DEV).eval()


def run_session(model, prompt, n_new, chained, use_graph=True):
    T = prompt.numel()
    total = T + n_new + 1
    model.reset_cache()
    model.__dict__.pop("_decode_sessions", None)
    sess = gb.DecodeSession(model, total, total, True, use_graph=use_graph, chained=chained)
    assert (sess.chain is not None) == chained
    with torch.no_grad():
        logits = sess.prefill(prompt.to(DEV))
        ops.argmax_advance(logits, sess.tokens, sess.pos)
        sess.capture()
        steps = []
        for _ in range(n_new - 1):
            steps.append(sess.step().float().cpu().clone())
    if sess.chain is not None:
        sess.chain.check()
    return sess.tokens[: T + n_new].cpu().clone(), torch.stack(steps)


@pytest.mark.parametrize("name,tile_cols,mode", [
    ("tiny-llama", 128, "gptq.int4-g128"), ("tiny-llama-hs128", 128, "gptq.int4-g128"), ("tiny-llama-gqa", 32, "gptq.int4-g32"),
    ("tiny-llama", -1, "gptq.int4"), ("tiny-neox", 32, "gptq.int4-g32"), ("tiny-falcon-mqa", 32, "gptq.int4-g32"),
])
def test_chained_step_equals_the_plain_step(name, tile_cols, mode):
    cfg, qsd, model = int4_model(name, tile_cols, mode)
    assert gb.chain_supported(model) is None
    prompt = synthetic_prompt(cfg, 9, 3)
    tok_a, log_a = run_session(model, prompt, 24, chained=False)
    tok_b, log_b = run_session(model, prompt, 24, chained=True)
    # same GEMV arithmetic (same slabs, same order); the attention may merge its key splits in a different grouping
    d = (log_a - log_b).abs()
    assert float(d.max()) <= 2 ** -6, float(d.max())
    assert float((d == 0).float().mean()) > 0.5
    assert torch.equal(tok_a, tok_b), (tok_a.tolist(), tok_b.tolist())
    # against the oracle: logits within the int4 bound
    oracle = om.OracleGPT(cfg, qsd, "gptq", tile_cols=tile_cols)
    with torch.no_grad():
        oracle(tok_b[:9].view(1, -1), 40, torch.arange(9))
        for i in range(9, 14):
            ref = oracle(tok_b[i].view(1, 1), 40, torch.tensor([i]))[0, -1].float()
            d = (log_b[i - 9] - ref).abs()
            assert float(d.max()) <= 1.5e-2 * max(1.0, float(ref.abs().max())) and float(d.mean()) <= 3e-3


def test_chained_step_replays_are_deterministic_and_generate_uses_it(monkeypatch):
    cfg, qsd, model = int4_model("tiny-llama", 128, "gptq.int4-g128")
    prompt = synthetic_prompt(cfg, 6, 5)
    monkeypatch.setattr(gb, "CHAINED_DEFAULT", True)
    y1 = L.generate(model, prompt.to(DEV), 40, 40, top_k=1).cpu()  # window == sequence; ring slots are exercised below
    sess = next(iter(model._decode_sessions.values()))
    assert sess.chain is not None and sess.graph is not None
    model.reset_cache()
    y2 = L.generate(model, prompt.to(DEV), 40, 40, top_k=1).cpu()
    assert torch.equal(y1, y2)
    oracle = om.OracleGPT(cfg, qsd, "gptq", tile_cols=128)
    y_ref = om.generate(oracle, prompt, 40, 40, greedy_ties_lowest=True)
    assert float((y1 == y_ref).float().mean()) >= 0.9
    # many replays back to back (the counters are re-armed inside the graph) never time out
    for _ in range(200):
        sess.pos.fill_(20)  # stay inside the window / token buffer: every replay advances pos
        sess.graph.replay()
    sess.chain.check()


def test_chained_step_refuses_what_it_cannot_chain():
    cfg = Config.from_name("tiny-neox")
    model = L.GPT(cfg).to(BF).to(DEV)  # dense bf16 Linears
    assert gb.chain_supported(model) is not None
    sess = gb.DecodeSession(model, 16, 16, True, chained=True)
    assert sess.chain is None  # falls back to the plain step (still HIP kernels)
    # ops without a chained form raise while a chain is being enqueued
    ch = ops.Chain(DEV, 4)
    ch.begin()
    try:
        with pytest.raises(L.ParrotHipError, match="no chained form"):
            ops.rmsnorm(torch.zeros((1, 64), dtype=BF, device=DEV), torch.ones(64, dtype=BF, device=DEV), 1e-5,
                        torch.zeros((1, 64), dtype=BF, device=DEV))
    finally:
        ch.end()
