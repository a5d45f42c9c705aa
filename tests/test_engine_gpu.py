"""The stream engine (csrc/engine.hip: one launch per decode token) against the multi-launch step and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd import _hip  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.engine import StreamEngine, e4_image, e8_image, e16_image  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.quantize.gptq import ColBlockQuantizedLinear, pack_nibbles  # noqa: E402
from lit_parrot_amd.synth import is_linear_key, synthetic_prompt, synthetic_state_dict  # noqa: E402
from oracle import int4 as o4  # noqa: E402
from oracle import model as om  # noqa: E402

DEV = torch.device("cuda", 0)
BF = torch.bfloat16


@pytest.fixture(autouse=True)
def cpu_rsqrt_mode():
    """Compare against the CPU oracle with the CPU-run reference's rsqrt rounding."""
    from lit_parrot_amd import ops

    ops.RMSNORM_RSQRT_MODE = 1
    yield
    ops.RMSNORM_RSQRT_MODE = 0


# ------------------------------------------------------------------------------------------------ E4 layout
def e4_expected(q: np.ndarray, s: np.ndarray, z: np.ndarray, q2=None, s2=None, z2=None) -> np.ndarray:
    """The E4 image as include/parrot_hip.h defines it, built element by element (independent of the repack kernel):
    q (N, K) nibbles, s / z (N, groups) uint16 bf16 bit patterns."""
    N, K = q.shape
    dual = q2 is not None
    nq = (K + 1023) // 1024
    spb = (nq + 3) // 4
    ng = (K + 127) // 128
    nblocks = N // 4 if dual else N // 8
    out = np.zeros((nblocks, (4 * nq + spb) * 1024), dtype=np.uint8)
    for B in range(nblocks):
        base = 0
        for sib in range(spb):
            nqs = min(4, nq - 4 * sib)
            for qq in range(nqs):
                for i in range(4):
                    for lane in range(64):
                        r, p = lane & 7, lane >> 3
                        t = 4 * (8 * (4 * sib + qq) + p) + i
                        if t * 32 >= K:
                            continue
                        src, row = (q, B * 8 + r) if not dual else ((q if r < 4 else q2), B * 4 + (r & 3))
                        k = src[row, t * 32:(t + 1) * 32].astype(np.uint32)
                        dws = []
                        for d in range(4):
                            v = 0
                            for ii in range(4):
                                v |= int(k[8 * d + 2 * ii]) << (4 * ii)
                                v |= int(k[8 * d + 2 * ii + 1]) << (16 + 4 * ii)
                            dws.append(v)
                        off = base + (qq * 4 + i) * 1024 + lane * 16
                        out[B, off:off + 16] = np.frombuffer(np.array(dws, dtype=np.uint32).tobytes(), dtype=np.uint8)
            meta = base + nqs * 4 * 1024
            for qq in range(nqs):
                for lane in range(64):
                    r, p = lane & 7, lane >> 3
                    G = 8 * (4 * sib + qq) + p
                    if G >= ng:
                        continue
                    ss, zz, row = (s, z, B * 8 + r) if not dual else ((s if r < 4 else s2), (z if r < 4 else z2), B * 4 + (r & 3))
                    w = int(ss[row, G]) | (int(zz[row, G]) << 16)
                    off = meta + (qq * 64 + lane) * 4
                    out[B, off:off + 4] = np.frombuffer(np.uint32(w).tobytes(), dtype=np.uint8)
            base += (nqs * 4 + 1) * 1024
    return out.reshape(-1)


def make_linear(N, K, seed):
    g = torch.Generator().manual_seed(seed)
    lin = ColBlockQuantizedLinear(K, N, False, tile_cols=128)
    q = torch.randint(0, 16, (N, K), generator=g, dtype=torch.uint8)
    ng = (K + 127) // 128
    s = (torch.rand(N, ng, generator=g) * 0.01 + 0.001).to(BF)
    z = torch.randint(0, 16, (N, ng), generator=g).to(BF)
    lin.quant_weight.copy_(pack_nibbles(q))
    lin.scales = s.clone()
    lin.zeros = z.clone()
    return lin.to(DEV), q.numpy(), s.view(torch.int16).numpy().view(np.uint16), z.view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("N,K", [(16, 128), (24, 352), (8, 1024), (16, 2080 - 32), (8, 4096), (8, 5120)])
def test_e4_repack_is_the_layout_the_header_defines(N, K):
    lin, q, s, z = make_linear(N, K, 11)
    got = e4_image(lin).cpu().numpy()
    want = e4_expected(q, s, z)
    assert got.shape == want.shape and np.array_equal(got, want)
    lin2, q2, s2, z2 = make_linear(N, K, 12)
    got = e4_image(lin, lin2).cpu().numpy()
    want = e4_expected(q, s, z, q2, s2, z2)
    assert got.shape == want.shape and np.array_equal(got, want)


def e16_expected(w: np.ndarray, w2=None) -> np.ndarray:
    """The E16 image element by element, from the layout include/parrot_hip.h defines: per 8 rows (4 + 4 of a SwiGLU pair)
    ceil(K / 64) pieces; in piece j lane l holds columns 64 j + 8 (l / 8) .. + 7 of row l % 8 (zero past K)."""
    N, K = w.shape
    dual = w2 is not None
    nblocks, pt = (N // 4 if dual else N // 8), (K + 63) // 64
    out = np.zeros((nblocks, pt, 64, 8), dtype=np.uint16)
    for B in range(nblocks):
        for ln in range(64):
            r, p = ln & 7, ln >> 3
            src = w2 if dual and r >= 4 else w
            row = B * 4 + (r & 3) if dual else B * 8 + r
            for j in range(pt):
                k0 = 64 * j + 8 * p
                if k0 < K:
                    out[B, j, ln] = src[row, k0:k0 + 8]
    return out.reshape(-1).view(np.uint8)


@pytest.mark.parametrize("N,K,dual", [(16, 128, False), (24, 384, False), (8, 2048, False), (16, 1056, False), (8, 1024, True), (12, 288, True)])
def test_e16_repack_is_the_layout_the_header_defines(N, K, dual):
    g = torch.Generator().manual_seed(N * 7 + K)
    lin = torch.nn.Linear(K, N, bias=False).to(BF)
    lin.weight.data = torch.randn((N, K), generator=g).to(BF)
    lin2 = None
    if dual:
        lin2 = torch.nn.Linear(K, N, bias=False).to(BF)
        lin2.weight.data = torch.randn((N, K), generator=g).to(BF)
    want = e16_expected(lin.weight.data.view(torch.int16).numpy().view(np.uint16),
                        lin2.weight.data.view(torch.int16).numpy().view(np.uint16) if dual else None)
    got = e16_image(lin.to(DEV), lin2.to(DEV) if dual else None).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("N,K,dual", [(16, 128, False), (24, 352, False), (8, 4096, False), (8, 2176, True), (12, 96, True)])
def test_e8_repack_is_the_layout_the_header_defines(N, K, dual):
    """E8 (LLM.int8 weights): per 8 rows (4 + 4 of a SwiGLU pair) ceil(K / 128) pieces; lane l of piece j holds the 16 columns
    128 j + 16 (l / 8) .. + 15 of row l % 8, zero past K; the rows' scales follow in block order."""
    from lit_parrot_amd.quantize.bnb import InferenceLinear8bitLt

    g = torch.Generator().manual_seed(N + K)
    lins = []
    for _ in range(2 if dual else 1):
        lin = InferenceLinear8bitLt(K, N, bias=False)
        lin.weight.data = torch.randn((N, K), generator=g)
        lins.append(lin.to(DEV))
    assert all(m.is_quantized for m in lins)
    img, scb = e8_image(lins[0], lins[1] if dual else None)
    got = img.cpu().numpy().view(np.int8)
    w = [m.weight.data.cpu().numpy() for m in lins]
    nblocks, pt = (N // 4 if dual else N // 8), (K + 127) // 128
    want = np.zeros((nblocks, pt, 64, 16), dtype=np.int8)
    for B in range(nblocks):
        for ln in range(64):
            r, p = ln & 7, ln >> 3
            src = w[1] if dual and r >= 4 else w[0]
            row = B * 4 + (r & 3) if dual else B * 8 + r
            for j in range(pt):
                k0 = 128 * j + 16 * p
                n = max(0, min(16, K - k0))
                want[B, j, ln, :n] = src[row, k0:k0 + n]
    assert np.array_equal(got, want.reshape(-1))
    s = [m.weight.SCB.cpu().numpy() for m in lins]
    want_s = np.concatenate([s[0].reshape(-1, 4), s[1].reshape(-1, 4)], axis=1) if dual else s[0].reshape(-1, 8)
    assert np.array_equal(scb.cpu().numpy().reshape(-1), want_s.reshape(-1))


# ------------------------------------------------------------------------------------------------ the step
def int8_model(name, threshold=6.0):
    cfg = Config.from_name(name)
    sd = {k: v.to(BF) for k, v in synthetic_state_dict(cfg, 4321, perturb=True).items()}
    with L.quantization("bnb.int8"):
        model = L.GPT(cfg)
    model.load_state_dict(dict(sd), strict=False)
    model = model.to(BF).to(DEV).eval()
    for m in model.modules():
        if hasattr(m, "threshold"):
            m.threshold = float(threshold)
    return cfg, sd, model


def bf16_model(name, **overrides):
    cfg = Config.from_name(name, **overrides)
    sd = {k: v.to(BF) for k, v in synthetic_state_dict(cfg, 4321, perturb=True).items()}
    model = L.GPT(cfg)
    model.load_state_dict(sd, strict=True)
    return cfg, sd, model.to(BF).to(DEV).eval()


def int4_model(name, tile_cols=128, mode="gptq.int4-g128", **overrides):
    cfg = Config.from_name(name, **overrides)
    sd = {k: v.to(BF) for k, v in synthetic_state_dict(cfg, 4321, perturb=True).items()}
    qsd = o4.quantize_state_dict(sd, tile_cols, is_linear_key)
    with L.quantization(mode):
        model = L.GPT(cfg)
    model.load_state_dict(qsd, strict=True)
    return cfg, qsd, model.to(BF).to(DEV).eval()


def run_session(model, prompt, n_new, engine, S=None, use_graph=True, follow=None):
    """Greedy decode through a DecodeSession; returns (tokens, logits of every decode step).  ``follow``: a token
    sequence that is forced after every step, so that two executors are compared on the same inputs at every position."""
    T = prompt.numel()
    total = T + n_new + 1
    model.reset_cache()
    model.__dict__.pop("_decode_sessions", None)
    sess = gb.DecodeSession(model, S or total, total, True, use_graph=use_graph, engine=engine)
    assert (sess.eng is not None) == engine
    with torch.no_grad():
        logits = sess.prefill(prompt.to(DEV))
        L.ops.argmax_advance(logits, sess.tokens, sess.pos)
        if follow is not None:
            sess.tokens[: T + 1].copy_(follow[: T + 1])
        sess.capture()
        steps = []
        for i in range(n_new - 1):
            steps.append(sess.step().float().cpu().clone())
            if follow is not None:
                sess.tokens[: T + i + 2].copy_(follow[: T + i + 2])
    if sess.eng is not None:
        sess.eng.check_error()
    assert int(sess.pos.item()) == T + n_new - 1
    return sess.tokens[: T + n_new].cpu().clone(), torch.stack(steps)


def assert_same_step(log_a, log_b, tok_a, tok_b, T):
    """Two executors fed the same tokens: logits equal up to the different fp32 summation order of the GEMVs, and the
    engine's arg-max is an arg-max of the other's logits up to that noise."""
    d = (log_a - log_b).abs()
    scale = max(1.0, float(log_a.abs().max()))
    assert float(d.max()) <= 2 ** -5 * scale, float(d.max())
    assert float(d.mean()) <= 2e-3 * scale, float(d.mean())
    for i in range(log_a.shape[0]):
        t = int(tok_b[T + 1 + i])
        assert float(log_a[i].max() - log_a[i][t]) <= 2 ** -5 * scale, (i, t)


@pytest.mark.parametrize("name", ["tiny-llama", "tiny-llama-hs128", "tiny-llama-gqa", "tiny-falcon-40b"])
def test_engine_step_equals_the_multi_launch_step(name):
    cfg, qsd, model = int4_model(name)
    assert StreamEngine.supported(model) is None
    prompt = synthetic_prompt(cfg, 9, 3)
    tok_a, log_a = run_session(model, prompt, 24, engine=False)
    tok_b, log_b = run_session(model, prompt, 24, engine=True, follow=tok_a.to(DEV))
    assert_same_step(log_a, log_b, tok_a, tok_b, 9)
    # free-running engine: graph replay, eager launches and a second run give identical tokens and logits
    tok_c, log_c = run_session(model, prompt, 24, engine=True)
    tok_d, log_d = run_session(model, prompt, 24, engine=True, use_graph=False)
    assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)
    tok_e, log_e = run_session(model, prompt, 24, engine=True)
    assert torch.equal(tok_c, tok_e) and torch.equal(log_c, log_e)
    # against the oracle: logits within the int4 bound
    oracle = om.OracleGPT(cfg, qsd, "gptq", tile_cols=128)
    with torch.no_grad():
        pos = torch.arange(9)
        oracle(tok_c[:9].view(1, -1), 40, pos)
        for i in range(9, 14):
            ref = oracle(tok_c[i].view(1, 1), 40, torch.tensor([i]))[0, -1].float()
            d = (log_c[i - 9] - ref).abs()
            assert float(d.max()) <= 1.5e-2 * max(1.0, float(ref.abs().max())) and float(d.mean()) <= 3e-3


@pytest.mark.parametrize("name", ["tiny-neox-hs64", "tiny-neox-hs128", "tiny-llama", "tiny-llama-gqa", "tiny-falcon-40b", "tiny-falcon-7b"])
def test_engine_step_on_bf16_weights_equals_the_multi_launch_step(name):
    """bf16 Linears (E16 layout) - with LayerNorm + bias, GELU, biases on every Linear, a partial rotary width and the
    parallel residual of the NeoX family (model.py:166-171), with the Llama block, or with Falcon's grouped / multi-query
    attention (4 query heads per K/V head as two virtual groups of 2; 7 on one K/V head as seven of 1; shared attention norm) -
    against the multi-launch step on the same forced tokens, run to run, and against the oracle."""
    cfg, sd, model = bf16_model(name)
    assert StreamEngine.supported(model) is None
    prompt = synthetic_prompt(cfg, 9, 3)
    tok_a, log_a = run_session(model, prompt, 24, engine=False)
    tok_b, log_b = run_session(model, prompt, 24, engine=True, follow=tok_a.to(DEV))
    assert_same_step(log_a, log_b, tok_a, tok_b, 9)
    tok_c, log_c = run_session(model, prompt, 24, engine=True)
    tok_d, log_d = run_session(model, prompt, 24, engine=True, use_graph=False)
    assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)
    # a ring window smaller than the sequence
    tok_e, log_e = run_session(model, prompt, 24, engine=False, S=16)
    tok_f, log_f = run_session(model, prompt, 24, engine=True, S=16, follow=tok_e.to(DEV))
    assert_same_step(log_e, log_f, tok_e, tok_f, 9)
    oracle = om.OracleGPT(cfg, sd, "dense")
    with torch.no_grad():
        oracle(tok_c[:9].view(1, -1), 40, torch.arange(9))
        for i in range(9, 14):
            ref = oracle(tok_c[i].view(1, 1), 40, torch.tensor([i]))[0, -1].float()
            d = (log_c[i - 9] - ref).abs()
            scale = max(1.0, float(ref.abs().max()))
            assert float(d.max()) <= 2 ** -5 * scale and float(d.mean()) <= 2e-3 * scale, (i, float(d.max()), float(d.mean()))


@pytest.mark.parametrize("name,quant", [("tiny-falcon-40b", False), ("tiny-falcon-40b", True), ("tiny-llama", True), ("tiny-neox-hs64", False)])
def test_engine_k_chunked_down_projection(name, quant, monkeypatch):
    """A down-projection whose input does not fit LDS (Falcon-40B: 32768 columns) runs as one op per K-chunk, the rows' sums
    accumulated in the CU in chunk order: forced here on small models (chunks of 256 / 128 columns, ragged last chunk),
    against the multi-launch step on the same tokens."""
    monkeypatch.setattr(StreamEngine, "CHUNK_ABOVE", 256)
    monkeypatch.setattr(StreamEngine, "CHUNK", 128 if name == "tiny-llama" else 256)
    cfg, _, model = int4_model(name) if quant else bf16_model(name)
    assert len(StreamEngine._down_chunks(cfg)) >= 3 and StreamEngine.supported(model) is None
    prompt = synthetic_prompt(cfg, 9, 3)
    tok_a, log_a = run_session(model, prompt, 20, engine=False)
    tok_b, log_b = run_session(model, prompt, 20, engine=True, follow=tok_a.to(DEV))
    assert_same_step(log_a, log_b, tok_a, tok_b, 9)
    tok_c, log_c = run_session(model, prompt, 20, engine=True)
    tok_d, log_d = run_session(model, prompt, 20, engine=True, use_graph=False)
    assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)


@pytest.mark.parametrize("name", ["tiny-llama", "tiny-falcon-7b"])
def test_engine_on_per_channel_int4(name):
    """The reference's plain "gptq.int4" (one scale and zero per output row, quantize/gptq.py with tile_cols = -1): the E4
    image repeats the row's pair for every group of 128 columns."""
    cfg = Config.from_name(name)
    sd = {k: v.to(BF) for k, v in synthetic_state_dict(cfg, 4321, perturb=True).items()}
    qsd = o4.quantize_state_dict(sd, -1, is_linear_key)
    with L.quantization("gptq.int4"):
        model = L.GPT(cfg)
    model.load_state_dict(qsd, strict=True)
    model = model.to(BF).to(DEV).eval()
    assert StreamEngine.supported(model) is None
    prompt = synthetic_prompt(cfg, 9, 3)
    tok_a, log_a = run_session(model, prompt, 20, engine=False)
    tok_b, log_b = run_session(model, prompt, 20, engine=True, follow=tok_a.to(DEV))
    assert_same_step(log_a, log_b, tok_a, tok_b, 9)


@pytest.mark.parametrize("name,threshold", [("tiny-llama", 6.0), ("tiny-llama", 1.5), ("tiny-llama-hs128", 2.0), ("tiny-llama-gqa", 6.0),
                                            ("tiny-falcon-40b", 2.0), ("tiny-falcon-7b", 6.0)])
def test_engine_step_on_llm_int8_weights(name, threshold):
    """LLM.int8 Linears on the engine (E8 layout): the activation quantiser runs in the gather (fp16 cast, outlier split at the
    threshold, row absmax over the waves, int8 image + outlier list in LDS), units are v_dot4_i32_i8 sums (exact) plus the
    outlier columns in fp16.  Against the multi-launch step on the same forced tokens - a lowered threshold makes a good part
    of every input vector outliers -, run to run, and against the oracle, at the int8 path's own bound (every Linear
    re-quantises its input: one bf16 ulp upstream can move an activation to the next int8 step)."""
    cfg, sd, model = int8_model(name, threshold)
    assert StreamEngine.supported(model) is None
    prompt = synthetic_prompt(cfg, 9, 3)
    tok_a, log_a = run_session(model, prompt, 20, engine=False)
    tok_b, log_b = run_session(model, prompt, 20, engine=True, follow=tok_a.to(DEV))
    d = (log_a - log_b).abs()
    scale = max(1.0, float(log_a.abs().max()))
    assert float(d.max()) <= 3e-2 * scale and float(d.mean()) <= 4e-3 * scale, (float(d.max()), float(d.mean()))
    tok_c, log_c = run_session(model, prompt, 20, engine=True)
    tok_d, log_d = run_session(model, prompt, 20, engine=True, use_graph=False)
    assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)
    oracle = om.OracleGPT(cfg, sd, "int8", threshold=threshold)
    with torch.no_grad():
        oracle(tok_c[:9].view(1, -1), 40, torch.arange(9))
        for i in range(9, 13):
            ref = oracle(tok_c[i].view(1, 1), 40, torch.tensor([i]))[0, -1].float()
            dd = (log_c[i - 9] - ref).abs()
            # (tokens tok_c == tok_a as long as the executors agree; the multi-launch step's own distance to the oracle is the yardstick)
            da = (log_a[i - 9] - ref).abs() if torch.equal(tok_a[: i + 1], tok_c[: i + 1]) else dd
            assert float(dd.max()) <= 3e-2 * max(1.0, float(ref.abs().max())), (i, float(dd.max()))
            assert float(dd.mean()) <= max(4e-3, 1.25 * float(da.mean())), (i, float(dd.mean()), float(da.mean()))


def test_engine_int8_outlier_list_overflow_is_reported():
    """More outlier columns in one input vector than the LDS list holds (1024): the launch sets its error word and the host
    raises - never a silently truncated sum."""
    cfg, _, model = int8_model("tiny-falcon-40b", threshold=1e-4)  # nearly all of the 2048-column MLP input is an "outlier"
    prompt = synthetic_prompt(cfg, 9, 3)
    with pytest.raises(_hip.ParrotHipError, match="error word"):
        run_session(model, prompt, 6, engine=True)


def test_engine_ring_window_and_generate(monkeypatch):
    """generate() end to end on the engine, with a window smaller than the sequence (ring slots) and sampling."""
    cfg, qsd, model = int4_model("tiny-llama")
    prompt = synthetic_prompt(cfg, 6, 5)
    monkeypatch.setattr(gb, "ENGINE_DEFAULT", True)
    y = L.generate(model, prompt.to(DEV), 40, 40, top_k=1).cpu()
    sess = next(iter(model._decode_sessions.values()))
    assert sess.eng is not None and sess.graph is not None
    sess.eng.check_error()
    oracle = om.OracleGPT(cfg, qsd, "gptq", tile_cols=128)
    y_ref = om.generate(oracle, prompt, 40, 40, greedy_ties_lowest=True)
    assert float((y == y_ref).float().mean()) >= 0.9
    # a window of 16 slots under 40 positions: the ring wraps twice; same tokens as the multi-launch step on the same window
    tok_a, log_a = run_session(model, prompt, 30, engine=False, S=16)
    tok_b, log_b = run_session(model, prompt, 30, engine=True, S=16, follow=tok_a.to(DEV))
    assert_same_step(log_a, log_b, tok_a, tok_b, 6)
    # non-greedy: the launch stops at the logits, the reference's sampling ops draw the token
    model.reset_cache()
    torch.manual_seed(3)
    a = L.generate(model, prompt.to(DEV), 20, 20, temperature=0.9, top_k=4).cpu()
    model.reset_cache()
    torch.manual_seed(3)
    b = L.generate(model, prompt.to(DEV), 20, 20, temperature=0.9, top_k=4).cpu()
    assert torch.equal(a, b)
    next(iter(model._decode_sessions.values())).eng.check_error()


def test_engine_refuses_unsupported_models():
    model = L.GPT(Config.from_name("tiny-neox")).to(BF).to(DEV)  # head size 32, dense bf16 Linears
    assert StreamEngine.supported(model) is not None
    cfg, _, m2 = int4_model("tiny-llama", tile_cols=32, mode="gptq.int4-g32")
    assert "group size" in StreamEngine.supported(m2)
    with pytest.raises(_hip.ParrotHipError):
        StreamEngine(m2, torch.zeros(8, dtype=torch.int64, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV), [], 8, True)


@torch.no_grad()
def test_engine_at_llama2_7b_width_and_a_long_window():
    """The engine on the real launch shapes (4096 / 12288 / 11008 / 32000 rows and columns: K = 11008 spans three ring slots
    per block, 8 key splits per head), two layers deep, against the multi-launch step on the same forced tokens - once behind a
    short prompt and once behind a 1100-token prompt in a 1200-slot window, where a CU's keys span several K/V ring slots."""
    from lit_parrot_amd.config import name_to_config
    from lit_parrot_amd.synth import build_synthetic_model

    cfg = Config(**{**name_to_config["Llama-2-7b-hf"], "n_layer": 2})
    model = build_synthetic_model(cfg, "gptq.int4-g128", seed=1234, device=DEV)
    assert StreamEngine.supported(model) is None
    for T, S, n in ((40, 96, 16), (1100, 1200, 12)):
        prompt = synthetic_prompt(cfg, T, 7)
        tok_a, log_a = run_session(model, prompt, n, engine=False, S=S)
        tok_b, log_b = run_session(model, prompt, n, engine=True, S=S, follow=tok_a.to(DEV))
        d = (log_a - log_b).abs()
        scale = max(1.0, float(log_a.abs().max()))
        assert float(d.max()) <= 2 ** -5 * scale and float(d.mean()) <= 2e-3 * scale, (T, float(d.max()), float(d.mean()))
        tok_c, log_c = run_session(model, prompt, n, engine=True, S=S)
        tok_d, log_d = run_session(model, prompt, n, engine=True, S=S, use_graph=False)
        assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)
    del model
    torch.cuda.empty_cache()


@torch.no_grad()
def test_engine_at_stablelm_3b_width():
    """The bf16 build with the 6-slot ring on StableLM-3B's launch shapes (4096 / 12288 / 16384 / 50688 rows and columns:
    the MLP's down-projection has 16 units per block, lm_head 24-25 blocks per CU), two layers deep, against the
    multi-launch step on the same forced tokens, behind a short and a 700-token prompt; run to run identical."""
    from lit_parrot_amd.config import name_to_config
    from lit_parrot_amd.synth import build_synthetic_model

    cfg = Config(**{**name_to_config["stablelm-base-alpha-3b"], "n_layer": 2})
    model = build_synthetic_model(cfg, None, seed=1234, device=DEV)
    assert StreamEngine.supported(model) is None
    for T, S, n in ((40, 96, 16), (700, 800, 12)):
        prompt = synthetic_prompt(cfg, T, 7)
        tok_a, log_a = run_session(model, prompt, n, engine=False, S=S)
        tok_b, log_b = run_session(model, prompt, n, engine=True, S=S, follow=tok_a.to(DEV))
        d = (log_a - log_b).abs()
        scale = max(1.0, float(log_a.abs().max()))
        assert float(d.max()) <= 2 ** -5 * scale and float(d.mean()) <= 2e-3 * scale, (T, float(d.max()), float(d.mean()))
        tok_c, log_c = run_session(model, prompt, n, engine=True, S=S)
        tok_d, log_d = run_session(model, prompt, n, engine=True, S=S, use_graph=False)
        assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)
    a = run_session(model, synthetic_prompt(cfg, 20, 9), 200, engine=True, S=128)
    b = run_session(model, synthetic_prompt(cfg, 20, 9), 200, engine=True, S=128)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    del model
    torch.cuda.empty_cache()


@torch.no_grad()
@pytest.mark.parametrize("name,mode", [("falcon-7b", None), ("falcon-40b", "gptq.int4-g128"), ("falcon-7b", "bnb.int8")])
def test_engine_at_falcon_widths(name, mode):
    """Falcon's launch shapes, two layers deep, against the multi-launch step on the same forced tokens: Falcon-7B bf16 (71
    query heads on one K/V head = 71 virtual groups on 3 CUs each, n_embd 4544 = 4.4 units of 1024 columns, LayerNorm weight
    and bias in a slot each, shared attention norm, the 18176-column down-projection as K-chunks of 8192 + 8192 + 1792) and
    Falcon-40B int4 g128 (8 K/V groups of 16 query heads = 64 virtual groups of 2, n_embd 8192, four K-chunks of 8192);
    Falcon-7B LLM.int8 on the wide int8 build."""
    from lit_parrot_amd.config import name_to_config
    from lit_parrot_amd.synth import build_synthetic_model

    cfg = Config(**{**name_to_config[name], "n_layer": 2})
    model = build_synthetic_model(cfg, mode, seed=1234, device=DEV)
    int8 = mode == "bnb.int8"  # (the wide int8 build: the 18176-column input is one image of 9 units, no K-chunks)
    assert StreamEngine.supported(model) is None and (int8 or len(StreamEngine._down_chunks(cfg)) >= 3)
    tol_max, tol_mean = (3e-2, 4e-3) if int8 else (2 ** -5, 2e-3)
    for T, S, n in ((40, 96, 12), (600, 700, 8)):
        prompt = synthetic_prompt(cfg, T, 7)
        tok_a, log_a = run_session(model, prompt, n, engine=False, S=S)
        tok_b, log_b = run_session(model, prompt, n, engine=True, S=S, follow=tok_a.to(DEV))
        d = (log_a - log_b).abs()
        scale = max(1.0, float(log_a.abs().max()))
        assert float(d.max()) <= tol_max * scale and float(d.mean()) <= tol_mean * scale, (T, float(d.max()), float(d.mean()))
        tok_c, log_c = run_session(model, prompt, n, engine=True, S=S)
        tok_d, log_d = run_session(model, prompt, n, engine=True, S=S, use_graph=False)
        assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)
    del model
    torch.cuda.empty_cache()


@torch.no_grad()
@pytest.mark.parametrize("name", ["Llama-2-13b-hf", "Llama-2-70b-hf"])
def test_engine_at_the_larger_llama2_widths(name):
    """The rest of the Llama-2 family at its launch shapes, int4 g128, two layers deep: 13B (n_embd 5120, 40 heads on 6 CUs each,
    a 13824-column down-projection: the wide build) and 70B (n_embd 8192, 8 K/V groups of 8 query heads = 32 virtual groups of
    2, a 28672-column down-projection in four K-chunks) against the multi-launch step on the same forced tokens."""
    from lit_parrot_amd.config import name_to_config
    from lit_parrot_amd.synth import build_synthetic_model

    cfg = Config(**{**name_to_config[name], "n_layer": 2})
    model = build_synthetic_model(cfg, "gptq.int4-g128", seed=1234, device=DEV)
    assert StreamEngine.supported(model) is None
    for T, S, n in ((40, 96, 10), (500, 600, 6)):
        prompt = synthetic_prompt(cfg, T, 7)
        tok_a, log_a = run_session(model, prompt, n, engine=False, S=S)
        tok_b, log_b = run_session(model, prompt, n, engine=True, S=S, follow=tok_a.to(DEV))
        d = (log_a - log_b).abs()
        scale = max(1.0, float(log_a.abs().max()))
        assert float(d.max()) <= 2 ** -5 * scale and float(d.mean()) <= 2e-3 * scale, (T, float(d.max()), float(d.mean()))
        tok_c, log_c = run_session(model, prompt, n, engine=True, S=S)
        tok_d, log_d = run_session(model, prompt, n, engine=True, S=S, use_graph=False)
        assert torch.equal(tok_c, tok_d) and torch.equal(log_c, log_d)
    del model
    torch.cuda.empty_cache()


@torch.no_grad()
def test_engine_soak_is_deterministic_and_error_free():
    """Hand-off races show up as run-to-run differences or as a tripped bounded wait: 600 free-running steps through a ring
    window (the K/V ring wraps nine times), twice, plus 300 steps at Llama-2-7B width - identical tokens and logits, error
    word clear, epoch advanced once per step."""
    from lit_parrot_amd.config import name_to_config
    from lit_parrot_amd.synth import build_synthetic_model

    cfg, _, model = int4_model("tiny-llama-hs128")
    prompt = synthetic_prompt(cfg, 5, 8)
    # (block_size 128 caps the positions: 5 + 120 tokens in a 13-slot window)
    a = run_session(model, prompt, 120, engine=True, S=13)
    b = run_session(model, prompt, 120, engine=True, S=13)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    cfg7 = Config(**{**name_to_config["Llama-2-7b-hf"], "n_layer": 2})
    model7 = build_synthetic_model(cfg7, "gptq.int4-g128", seed=1234, device=DEV)
    prompt7 = synthetic_prompt(cfg7, 20, 9)
    a = run_session(model7, prompt7, 300, engine=True, S=128)
    b = run_session(model7, prompt7, 300, engine=True, S=128)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    sess = gb.DecodeSession(model7, 64, 64, True, engine=True)
    e0 = int(sess.eng.epoch.item())
    L.ops.argmax_advance(sess.prefill(prompt7.to(DEV)), sess.tokens, sess.pos)
    for _ in range(7):
        sess.step()
    assert int(sess.eng.epoch.item()) == e0 + 7
    sess.eng.check_error()
    del model7
    torch.cuda.empty_cache()
