"""The N > 1 path of bench.py: independent replicas that only meet to agree on (max seconds, total tokens).
Exercised with world_size 2 over gloo on the CPU (the data path itself has no collective)."""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parents[1]


def _worker(rank: int, world: int, port: int, out):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    assert bench.rank_env() == (rank, rank, world)
    dev = torch.device("cpu")
    bench.barrier(world, dev)
    # rank r "decoded" 100 tokens in (1 + r) seconds: the job took 2 s and produced 200 tokens
    secs, units = bench.max_over_ranks(1.0 + rank, 100, world, dev)
    out[rank] = (secs, units)
    bench.barrier(world, dev)
    dist.destroy_process_group()


def test_replica_aggregation_world_size_2():
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out[0] == out[1] == (2.0, 200)  # whole-job value = 200 tokens / 2 s on every rank


def test_byte_accounting_matches_the_survey():
    sys.path.insert(0, str(REPO))
    import bench
    from lit_parrot_amd.config import Config

    l7 = Config.from_name("Llama-2-7b-hf")
    w, kv = bench.token_bytes(l7, "gptq.int4-g128", 1)
    assert abs(w - 3.510e9) < 2e6  # 3.3035 GB + 0.2065 GB
    assert kv - l7.n_embd * 2 == 524_288  # KV bytes per context token
    w8, _ = bench.token_bytes(l7, "bnb.int8", 1)
    assert abs(w8 - (6.607e9 + 5.6e6)) < 5e6
    s3 = Config.from_name("stablelm-base-alpha-3b")
    assert abs(bench.token_bytes(s3, None, 1)[0] - 6.858e9) < 2e6
    f40 = Config.from_name("falcon-40b")
    wf, kvf = bench.token_bytes(f40, "gptq.int4-g128", 1)
    assert abs(wf - 21.94e9) < 2e7 and kvf - f40.n_embd * 2 == 122_880
    kb = bench.kernel_bytes_per_token(l7, "gptq.int4-g128")
    assert kb["w4_gemv"][1] == 97 and kb["w4_gemv_dual"][1] == 32
    assert kb["w4_gemv"][0] + kb["w4_gemv_dual"][0] == w
