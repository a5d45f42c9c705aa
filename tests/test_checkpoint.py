"""Checkpoint path: HF -> lit-gpt conversion against what the reference's own convert script produced
(tests/golden/convert_hf.npz, digests of the fp16 tensors: the conversion only renames and permutes rows), memory-mapped
loading, and - on the GPU - streaming a dense checkpoint into bf16 / int4 / LLM.int8 models."""
import hashlib

import numpy as np
import pytest
import torch

from lit_parrot_amd.checkpoint import (HFConverter, convert_hf_state_dict, hf_family, interleave_qkv, lazy_load,
                                       stream_load, synthetic_hf_state_dict)
from lit_parrot_amd.config import Config, name_to_config

FAMILIES = (("llama", "tiny-llama-gqa"), ("falcon-7b", "tiny-falcon-mqa"), ("falcon-40b", "tiny-falcon-gqa"), ("neox", "tiny-neox"))


def _digest(t: torch.Tensor) -> np.ndarray:
    a = np.ascontiguousarray(t.to(torch.float16).numpy())
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("family,cfg_name", FAMILIES)
def test_convert_matches_reference_script(golden_dir, family, cfg_name):
    gold = np.load(golden_dir / "convert_hf.npz")
    want = {k.split("|", 1)[1]: gold[k] for k in gold.files if k.startswith(family + "|") and not k.endswith("|shape")}
    cfg = Config(**dict(name_to_config[cfg_name]))
    hf = synthetic_hf_state_dict(family, name_to_config[cfg_name], seed=77)
    got = convert_hf_state_dict([hf], cfg, family)
    assert sorted(got) == sorted(want)
    for k, t in got.items():
        assert list(t.shape) == gold[f"{family}|{k}|shape"].tolist(), k
        assert np.array_equal(_digest(t), want[k]), k


def test_convert_shards_in_any_order(golden_dir):
    """q, k and v of one layer arriving in three different shards, last shard first."""
    family, cfg_name = "llama", "tiny-llama-gqa"
    cfg = Config(**dict(name_to_config[cfg_name]))
    hf = synthetic_hf_state_dict(family, name_to_config[cfg_name], seed=77)
    names = list(hf)
    shards = [{n: hf[n] for n in names[i::3]} for i in (2, 0, 1)]
    got = convert_hf_state_dict(shards, cfg)
    one = convert_hf_state_dict([hf], cfg)
    assert sorted(got) == sorted(one) and all(torch.equal(got[k], one[k]) for k in one)


def test_interleave_layout():
    """Row r of group g: [q heads of g | k_g | v_g] (model.py:208-214 splits it back the same way)."""
    cfg = Config(**dict(name_to_config["tiny-llama-gqa"]))
    hs, per, groups = cfg.head_size, cfg.n_head // cfg.n_query_groups, cfg.n_query_groups
    q = torch.arange(groups * per * hs, dtype=torch.float32)[:, None].expand(-1, 4).contiguous()
    k = 1000 + torch.arange(groups * hs, dtype=torch.float32)[:, None].expand(-1, 4).contiguous()
    v = 2000 + torch.arange(groups * hs, dtype=torch.float32)[:, None].expand(-1, 4).contiguous()
    w = interleave_qkv(q, k, v, cfg).view(groups, per + 2, hs, 4)
    for g in range(groups):
        assert torch.equal(w[g, :per].reshape(-1, 4), q[g * per * hs:(g + 1) * per * hs])
        assert torch.equal(w[g, per], k[g * hs:(g + 1) * hs])
        assert torch.equal(w[g, per + 1], v[g * hs:(g + 1) * hs])
    with pytest.raises(ValueError):
        interleave_qkv(q[:-1], k, v, cfg)


def test_convert_errors():
    cfg = Config(**dict(name_to_config["tiny-llama-gqa"]))
    hf = synthetic_hf_state_dict("llama", name_to_config["tiny-llama-gqa"], seed=1)
    conv = HFConverter(cfg)
    with pytest.raises(KeyError):
        conv.add("model.layers.0.self_attn.rotary_emb.unknown", torch.zeros(1))
    conv.add_shard({n: t for n, t in hf.items() if "v_proj" not in n})
    with pytest.raises(ValueError):
        conv.finish()
    with pytest.raises(ValueError):
        HFConverter(cfg, family="gpt2")


def test_family_detection():
    assert hf_family(Config.from_name("Llama-2-7b-hf")) == "llama"
    assert hf_family(Config.from_name("falcon-40b")) == "falcon-40b"
    assert hf_family(Config.from_name("falcon-7b")) == "falcon-7b"
    assert hf_family(Config.from_name("pythia-160m")) == "neox"
    assert hf_family(Config.from_name("stablelm-base-alpha-3b")) == "neox"


def test_lazy_load_round_trip(tmp_path):
    sd = {"a.weight": torch.randn(64, 32), "b.bias": torch.arange(7, dtype=torch.int64)}
    torch.save(sd, tmp_path / "ckpt.pth")
    lazy = lazy_load(tmp_path / "ckpt.pth")
    assert sorted(lazy) == sorted(sd)
    for k in sd:
        assert torch.equal(lazy[k], sd[k])


# ------------------------------------------------------------------------------------------------------------ GPU
def _dense_checkpoint(cfg_name: str, family: str):
    cfg = Config(**dict(name_to_config[cfg_name]))
    hf = synthetic_hf_state_dict(family, name_to_config[cfg_name], seed=5)
    return cfg, convert_hf_state_dict([hf], cfg, family)


@pytest.mark.gpu
@pytest.mark.parametrize("family,cfg_name", FAMILIES)
def test_stream_load_dense_equals_load_state_dict(hip_lib, tmp_path, family, cfg_name):
    from lit_parrot_amd import GPT

    cfg, sd = _dense_checkpoint(cfg_name, family)
    torch.save(sd, tmp_path / "lit_model.pth")
    with torch.device("cuda"):
        a, b = GPT(cfg).to(torch.bfloat16).eval(), GPT(cfg).to(torch.bfloat16).eval()
    assert stream_load(a, lazy_load(tmp_path / "lit_model.pth")) == []
    b.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    idx = torch.randint(0, cfg.padded_vocab_size, (1, 12), device="cuda")
    assert torch.equal(a(idx), b(idx))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gptq.int4-g128", "gptq.int4", "bnb.int8"])
def test_stream_load_quantises_on_the_way_in(hip_lib, tmp_path, mode):
    """A dense checkpoint streamed into a quantised model gives the same module state as quantising each matrix by hand
    (round-to-nearest grid of find_params_weight for int4, row absmax for LLM.int8), and a quantised checkpoint written by
    that model streams back bit for bit."""
    from lit_parrot_amd import GPT, quantization
    from lit_parrot_amd.quantize.bnb import InferenceLinear8bitLt
    from lit_parrot_amd.quantize.gptq import ColBlockQuantizedLinear, pack_nibbles, rtn_quantize

    cfg, sd = _dense_checkpoint("tiny-llama-hs128", "llama")
    with torch.device("cuda"), quantization(mode):
        model = GPT(cfg)
    model = model.to(torch.bfloat16).eval()
    assert stream_load(model, sd) == []
    n_checked = 0
    for name, mod in model.named_modules():
        if isinstance(mod, ColBlockQuantizedLinear):
            q, s, z = rtn_quantize(sd[name + ".weight"].cuda().to(torch.bfloat16), mod.tile_cols)
            assert torch.equal(mod.quant_weight, pack_nibbles(q)) and torch.equal(mod.scales, s) and torch.equal(mod.zeros, z), name
            n_checked += 1
        elif isinstance(mod, InferenceLinear8bitLt):
            from oracle.int8 import quantize_weight_rows  # the checker (CPU): row absmax int8 of `weight.half()` (quantize/bnb.py:54)

            CB, SCB = quantize_weight_rows(sd[name + ".weight"])  # the checkpoint's own precision, not a bf16 detour
            assert mod.weight.dtype == torch.int8
            assert torch.equal(mod.weight.SCB.cpu(), SCB), name
            assert torch.equal(mod.weight.data.cpu(), CB), name
            n_checked += 1
    assert n_checked == 1 + cfg.n_layer * 5
    idx = torch.randint(0, cfg.padded_vocab_size, (1, 9), device="cuda")
    want = model(idx)
    assert torch.isfinite(want.float()).all()
    if mode.startswith("gptq"):
        torch.save(model.state_dict(), tmp_path / "q.pth")
        with torch.device("cuda"), quantization(mode):
            again = GPT(cfg)
        again = again.to(torch.bfloat16).eval()
        assert stream_load(again, lazy_load(tmp_path / "q.pth")) == []
        assert torch.equal(again(idx), want)


@pytest.mark.gpu
def test_stream_load_reports_unused_and_shape_errors(hip_lib):
    from lit_parrot_amd import GPT

    cfg, sd = _dense_checkpoint("tiny-neox", "neox")
    with torch.device("cuda"):
        model = GPT(cfg).to(torch.bfloat16)
    extra = dict(sd)
    extra["transformer.h.0.attn.bias_mask"] = torch.zeros(1)
    assert stream_load(model, extra) == ["transformer.h.0.attn.bias_mask"]
    bad = dict(sd)
    bad["lm_head.weight"] = torch.zeros(3, 3)
    with pytest.raises(ValueError):
        stream_load(model, bad)
