"""CPU-side checks: the C-ABI library loads and exports what include/parrot_hip.h declares, argument validation
works without a GPU, and the host-side mirror of the reference interface behaves like the reference's."""
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import lit_parrot_amd as L
from lit_parrot_amd import _hip
from lit_parrot_amd.config import Config, name_to_config
from lit_parrot_amd.quantize.gptq import ColBlockQuantizedLinear, pack_nibbles, rtn_quantize
from lit_parrot_amd.synth import is_linear_key, synthetic_state_dict
from oracle import int4 as o4
from oracle import model as om

REPO = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (REPO / "include" / "parrot_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(parrot_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/parrot_hip.h but not exported"
    assert sorted(_hip.SIGNATURES) == names, "ctypes signatures and the header disagree"
    assert hip_lib.parrot_version() == 1


def test_argument_validation_needs_no_gpu(hip_lib):
    # null pointers / bad shapes are refused before any HIP call
    assert hip_lib.parrot_w4_gemv(None, None, None, 0, 1, None, None, 0, None, 0, 8, 64, 64, 0, None, None) == -1
    assert "null pointer" in _hip.last_error()
    assert hip_lib.parrot_rmsnorm(None, 0, None, None, 0, 1, 64, 1e-5, 0, None) == -1
    assert hip_lib.parrot_attn_decode(None, 1, None, None, None, 1, 1, 64, 8, 1, None, None, 0, 0, None) == -1
    assert hip_lib.parrot_w4_packed_bytes(8, 100, 32) == -3  # K not a multiple of 32
    assert "multiple of 32" in _hip.last_error()
    assert hip_lib.parrot_kernel_name(0) == b"w4_gemv" and hip_lib.parrot_kernel_name(999) == b"?"
    # stream engine: layout sizes and shape limits are host arithmetic
    assert hip_lib.parrot_e4_bytes(4096, 4096, 0) == 512 * 17 * 1024  # 512 blocks of 8 rows: 16 weight pieces + 1 metadata piece
    assert hip_lib.parrot_e4_bytes(11008, 4096, 1) == (11008 // 4) * 17 * 1024  # SwiGLU pair: 4 + 4 rows per block
    assert hip_lib.parrot_e4_bytes(4096, 11008, 0) == 512 * (44 + 3) * 1024  # 11 quads in 3 slots
    assert hip_lib.parrot_e4_bytes(4096, 4100, 0) == -3 and hip_lib.parrot_e4_bytes(4095, 4096, 0) == -3
    assert hip_lib.parrot_e4_bytes(64, 16384, 0) == 8 * (64 + 4) * 1024  # 16 quads in 4 slots
    assert hip_lib.parrot_e4_bytes(64, 17408, 0) == -3 and "stream engine" in _hip.last_error()
    assert hip_lib.parrot_e16_bytes(4096, 4096, 0) == 512 * 4 * 16384 == 4096 * 4096 * 2  # bf16: no metadata, no padding here
    assert hip_lib.parrot_e16_bytes(16, 384, 0) == 2 * 6 * 1024  # pieces of 64 columns: a row's last unit is as short as K allows
    # the launch's LDS: 7 ring slots up to K = 11264, 6 beyond (StableLM's 16384-wide MLP input)
    assert hip_lib.parrot_eng_lds_total(11008, 0, 88 * 272, 32 * 272) == 7 * 17 * 1024 + 120 * 272 + 3552 + 16 * 11 * 32
    assert hip_lib.parrot_eng_lds_total(16384, 1, 128 * 272, 32 * 272) == 6 * 17 * 1024 + 160 * 272 + 3552 + 16 * 16 * 32
    assert hip_lib.parrot_eng_lds_total(16384, 1, 128 * 272, 64 * 272) == -3
    assert hip_lib.parrot_eng_lds_bytes(4096, 0, 0, 0) == 32 * 272
    assert hip_lib.parrot_eng_lds_bytes(11008, 128, 1, 8) == 88 * 272
    assert hip_lib.parrot_eng_step(None, None) == -1
    assert hip_lib.parrot_w8_quantize_rows(None, 0, 1, 1, None, None, None) == -1


@pytest.mark.parametrize("N,K,group,expect", [
    (4096, 4096, 128, 4096 * (4096 // 2 + 4096 // 128 * 4)),  # 2 slabs of 16 groups: meta 64 B each, no padding
    (4096, 11008, 128, None), (32000, 4096, 128, None), (16, 768, -1, None), (16, 64, 32, None), (8, 32768, 128, None),
])
def test_w4_packed_size(hip_lib, N, K, group, expect):
    n = hip_lib.parrot_w4_packed_bytes(N, K, group)
    g = K if group == -1 else group
    payload = N * (K // 2 + -(-K // g) * 4)
    assert n >= payload and n % 16 == 0
    assert n <= payload * 1.02 + N * 16 * 16, "padding should stay marginal"
    if expect is not None:
        assert n == expect


def test_no_cpu_fallback():
    cfg = Config.from_name("tiny-llama")
    model = L.GPT(cfg).to(torch.bfloat16)
    with pytest.raises(L.ParrotHipError):
        model(torch.zeros((1, 4), dtype=torch.int64))
    lin = ColBlockQuantizedLinear(64, 16, False, bits=4, tile_cols=32)
    with pytest.raises(L.ParrotHipError):
        lin(torch.zeros(1, 64, dtype=torch.bfloat16))
    with pytest.raises(L.ParrotHipError):
        L.generate(model, torch.zeros(4, dtype=torch.int64), 8, 8, top_k=1)
    # positions past the RoPE tables are refused on the host (the reference raises in index_select, lit_gpt/model.py:88)
    with pytest.raises(L.ParrotHipError, match="block_size"):
        L.generate(model, torch.zeros(4, dtype=torch.int64), cfg.block_size + 1, cfg.block_size, top_k=1)
    # the product never imports the oracle
    for f in (REPO / "lit-parrot_amd").rglob("*.py"):
        assert "oracle" not in f.read_text().replace("oracle/", "").replace("the oracle", ""), f


def test_config_table_matches_reference_numbers():
    l7 = Config.from_name("Llama-2-7b-hf")
    assert (l7.n_layer, l7.n_embd, l7.n_head, l7.head_size, l7.intermediate_size, l7.padded_vocab_size) == (32, 4096, 32, 128, 11008, 32000)
    assert l7.n_linear_params() == 6_607_077_376  # 6.6071 G weights
    f40 = Config.from_name("falcon-40b")
    assert (f40.n_query_groups, f40.head_size, f40.qkv_size, f40.intermediate_size) == (8, 64, 9216, 32768)
    s3 = Config.from_name("stablelm-base-alpha-3b")
    assert (s3.n_layer, s3.rope_n_elem, s3.padded_vocab_size, s3.intermediate_size) == (16, 32, 50688, 16384)
    p = Config.from_name("pythia-160m")
    assert (p.n_layer, p.n_embd, p.head_size, p.rope_n_elem, p.padded_vocab_size) == (12, 768, 64, 16, 50304)
    # a lit_config.json written by the reference round-trips
    assert Config(**l7.to_dict()) == l7


def test_quantization_context_swaps_and_restores():
    orig = torch.nn.Linear
    with L.quantization("gptq.int4"):
        assert torch.nn.Linear is not orig
        m = L.GPT.from_name("tiny-llama")
    assert torch.nn.Linear is orig
    keys = set(m.state_dict())
    assert {"lm_head.quant_weight", "lm_head.scales", "lm_head.zeros", "transformer.wte.weight",
            "transformer.h.0.attn.attn.quant_weight", "transformer.h.1.mlp.fc_2.zeros"} <= keys
    assert not any(k.endswith("attn.attn.weight") for k in keys)
    lin = m.transformer.h[0].attn.attn
    assert lin.tile_cols == lin.in_features and lin.quant_weight.shape == (lin.out_features, lin.in_features // 2)
    assert lin.quant_weight.stride() == (1, lin.out_features)  # column-major, as the reference stores it
    assert not isinstance(lin, orig)  # like the reference class (gptq.py:205)
    with L.quantization("gptq.int4-g128"):
        g = L.GPT.from_name("tiny-llama")
    assert g.transformer.h[0].mlp.proj.scales.shape == (128, 3)  # ceil(352 / 128)
    with pytest.raises(RuntimeError):
        with L.quantization("gptq.int4"):
            raise RuntimeError("boom")
    assert torch.nn.Linear is orig  # restored on error (the reference would leak the patch)
    with pytest.raises(ValueError):
        with L.quantization("nope"):
            pass
    with L.quantization(None):
        assert torch.nn.Linear is orig
    with L.quantization("bnb.int8"):
        assert issubclass(torch.nn.Linear, orig)  # bnb classes ARE nn.Linear subclasses


def test_dense_model_state_dict_keys_are_the_reference_ones():
    for name in ("tiny-neox", "tiny-llama", "tiny-falcon-mqa"):
        cfg = Config.from_name(name)
        assert set(L.GPT(cfg).state_dict()) == set(synthetic_state_dict(cfg))


@pytest.mark.parametrize("tile_cols", [-1, 128, 64])
def test_gptq_module_format_against_reference_vectors(golden_dir, tile_cols):
    g = np.load(golden_dir / "gptq_linear.npz")
    key = f"g{tile_cols}_f32"
    q = torch.from_numpy(g[key + "_q"])
    N, K = q.shape
    lin = ColBlockQuantizedLinear(K, N, True, bits=4, tile_cols=tile_cols)
    lin.scales.copy_(torch.from_numpy(g[key + "_scales"]))
    lin.zeros.copy_(torch.from_numpy(g[key + "_zeros"]))
    grp = torch.arange(K) // lin.tile_cols
    lin.pack_weight((q.float() - lin.zeros[:, grp]) * lin.scales[:, grp])
    assert lin.quant_weight.stride() == (1, N)
    assert np.array_equal(lin.quant_weight.t().contiguous().numpy(), g[key + "_qw_mem"])
    assert torch.equal(lin.get_weight(torch.float32), torch.from_numpy(g[key + "_get_weight"]))
    # loading a reference-format state dict keeps the layout
    lin2 = ColBlockQuantizedLinear(K, N, True, bits=4, tile_cols=tile_cols)
    lin2.load_state_dict(lin.state_dict())
    assert lin2.quant_weight.stride() == (1, N) and torch.equal(lin2.quant_weight, lin.quant_weight)


def test_rtn_quantize_equals_oracle():
    gen = torch.Generator().manual_seed(5)
    w = (torch.randn(40, 352, generator=gen) * 0.02).to(torch.bfloat16)
    for tc in (128, -1, 32):
        q, s, z = rtn_quantize(w, tc)
        qw, so, zo = o4.rtn_quantize(w, tc, torch.bfloat16)
        assert torch.equal(pack_nibbles(q), qw) and torch.equal(s, so) and torch.equal(z, zo)
        assert pack_nibbles(q).stride() == (1, 40)
        assert float(z.min()) >= 0 and float(z.max()) <= 15


def test_rope_cache_is_the_reference_table(golden_dir):
    g = np.load(golden_dir / "pieces.npz")
    for n_elem in (8, 64):
        cos, sin = L.build_rope_cache(128, n_elem, torch.float32, torch.device("cpu"))
        c16, s16 = cos.half(), sin.half()  # fp32-session table rounded == golden (built in an fp32 session)
        assert torch.equal(c16.float(), torch.from_numpy(g[f"rope_cos_{n_elem}"]))
        assert torch.equal(s16.float(), torch.from_numpy(g[f"rope_sin_{n_elem}"]))
        # bf16 session: same as the oracle's bf16-session table
        cb, sb = L.build_rope_cache(128, n_elem, torch.bfloat16, torch.device("cpu"))
        co, so = om.rope_tables(128, n_elem, torch.bfloat16, math_dtype=torch.bfloat16)
        assert cb.dtype == torch.float16 and torch.equal(cb, co) and torch.equal(sb, so)


def test_synthetic_weights_are_deterministic():
    cfg = Config.from_name("tiny-neox")
    a, b = synthetic_state_dict(cfg, 11, perturb=True), synthetic_state_dict(cfg, 11, perturb=True)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert abs(float(a["lm_head.weight"].std()) - 0.02) < 2e-3
    assert sum(is_linear_key(k) for k in a) == 1 + 4 * cfg.n_layer


def test_stream_engine_host_plan(hip_lib):
    """The engine's launch plan is host arithmetic: virtual groups of the attention op, K-chunks of a down-projection that does not
    fit LDS, the LDS buffers of every BASELINE config and which kernel build they select (no GPU needed)."""
    from lit_parrot_amd._hip import ENG_W_E4, ENG_W_E8, ENG_W_E16, ENG_W_TWO_LOADERS
    from lit_parrot_amd.config import Config
    from lit_parrot_amd.engine import StreamEngine as E

    l7, slm, f40, f7, l70 = (Config.from_name(n) for n in ("Llama-2-7b-hf", "stablelm-base-alpha-3b", "falcon-40b", "falcon-7b", "Llama-2-70b-hf"))
    # (query heads per virtual group, virtual groups per K/V group, CUs per virtual group)
    assert E._attn_shape(l7) == (1, 1, 8) and E._attn_shape(slm) == (1, 1, 8)
    assert E._attn_shape(f40) == (2, 8, 4)   # 8 groups x 16 heads -> 64 virtual groups of 2 heads on 4 CUs each
    assert E._attn_shape(f7) == (1, 71, 3)   # one K/V head, 71 query heads -> 71 virtual groups on 3 CUs each
    assert E._attn_shape(l70) == (2, 4, 8)
    assert E._down_chunks(l7, ENG_W_E4) == [(0, 11008)] and E._down_chunks(slm) == [(0, 16384)]
    assert E._down_chunks(f40, ENG_W_E4) == [(0, 8192), (8192, 16384), (16384, 24576), (24576, 32768)]
    assert E._down_chunks(f7) == [(0, 8192), (8192, 16384), (16384, 18176)]
    assert E._down_chunks(f7, ENG_W_E8) == [(0, 18176)]  # an int8 image is half the size, and the row's absmax needs the whole input
    assert E._state_wfmt(l7, ENG_W_E4) == ENG_W_E4 and E._state_wfmt(f40, ENG_W_E4) == ENG_W_E4 | ENG_W_TWO_LOADERS
    assert E._state_wfmt(slm, ENG_W_E16) == ENG_W_E16 and E._state_wfmt(f7, ENG_W_E8) == ENG_W_E8
    for cfg, wfmt in ((l7, ENG_W_E4), (l7, ENG_W_E8), (slm, ENG_W_E16), (f40, ENG_W_E4), (f7, ENG_W_E16), (f7, ENG_W_E8), (l70, ENG_W_E4)):
        b0, b1, ab = E._lds_buffers(cfg, wfmt)
        assert b0 > 0 and b1 > 0 and ab == int(cfg.parallel_residual)
        kmax = max([cfg.n_embd] + [k1 - k0 for k0, k1 in E._down_chunks(cfg, wfmt)])
        total = hip_lib.parrot_eng_lds_total(kmax, E._state_wfmt(cfg, wfmt), b0, b1)
        assert 100 * 1024 < total <= 160 * 1024, (cfg.name, wfmt, total)
    # StableLM-3B is the tightest fit: 6 slots + a 16384-column image + the attention scratch in buffer 1
    b0, b1, _ = E._lds_buffers(slm, ENG_W_E16)
    assert hip_lib.parrot_eng_lds_total(16384, ENG_W_E16, b0, b1) > 159 * 1024
