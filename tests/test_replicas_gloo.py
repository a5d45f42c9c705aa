"""The N > 1 paths of bench.py: independent replicas that only meet to agree on (max seconds, total tokens).
Under an external launcher: world_size 2 over gloo on the CPU.  Without one: the parent spawns its replicas, pins a device
to each, releases them together and aggregates their lines - exercised here with a stub child (no GPU)."""
import json
import types
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


STUB = '''
import json, os, sys, time
i = int(os.environ["PARROT_BENCH_REPLICA"])
assert os.environ["PARROT_BENCH_WORLD"] == "2" and "WORLD_SIZE" not in os.environ and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
print("replica", i, "warming up", flush=True)
print("READY", flush=True)
assert sys.stdin.readline().strip() == "GO"
steps = int(sys.argv[sys.argv.index("--steps") + 1])
elapsed = 1.0 + i  # replica i "decodes" its steps in 1 + i seconds
print(json.dumps({"metric": "m", "value": steps / elapsed, "steps": steps, "elapsed_s": elapsed, "ms_per_step": elapsed / steps * 1e3,
                  "device": os.environ["HIP_VISIBLE_DEVICES"], "roofline": {"frac": 0.5 - 0.1 * i}, "step_roofline": {"frac": 0.4},
                  "config": {"workload": "stub", "replicas": 1}}), flush=True)
'''


def test_self_spawned_replicas_with_a_stub_child(tmp_path):
    """python bench.py --gpus 2 without a launcher: device pinning, start barrier over the pipes, aggregation."""
    sys.path.insert(0, str(REPO))
    import bench

    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    for devices, expect in (("", ["0", "1"]), ("0,0", ["0", "0"])):
        args = types.SimpleNamespace(gpus=2, devices=devices)
        out = bench.spawn_replicas(args, ["--gpus", "2", "--steps", "100"], child_cmd=[sys.executable, str(stub)])
        assert out["n_gpus"] == 2 and out["value"] == 200 / 2.0 and abs(out["ms_per_step"] - 20.0) < 1e-9
        assert [r["device"] for r in out["replicas"]] == expect
        assert [r["roofline_frac"] for r in out["replicas"]] == [0.5, 0.4]
        assert out["config"]["replicas"] == 2 and "no RCCL" in out["config"]["parallelism"]
        assert "elapsed_s" not in out
    with __import__("pytest").raises(SystemExit):
        bench.spawn_replicas(types.SimpleNamespace(gpus=2, devices="0"), [], child_cmd=[sys.executable, str(stub)])
    # a replica that dies before it is ready fails the job
    bad = tmp_path / "bad.py"
    bad.write_text("import sys; sys.exit(3)")
    with __import__("pytest").raises(SystemExit):
        bench.spawn_replicas(types.SimpleNamespace(gpus=2, devices=""), ["--steps", "1"], child_cmd=[sys.executable, str(bad)])


def test_committed_pmc_summary_names_the_running_kernels():
    """roofline.traffic comes from the newest profiles/*_pmc_traffic.json: it must have been taken with kernels of the names
    the benchmark runs today (a renamed or retired kernel makes the summary stale - re-measure, do not reuse the number)."""
    sys.path.insert(0, str(REPO))
    import bench

    for kernel, run in (("w4_gemv", "llama2-7b-int4-multilaunch"), ("w4_gemv_dual", "llama2-7b-int4-multilaunch"),
                        ("eng_token", "llama2-7b-int4-engine"), ("eng_token", "stablelm-3b-bf16-engine")):
        traffic, src, problem = bench.pmc_traffic(kernel, run)
        assert problem is None and traffic and src, (kernel, run, problem)
    assert bench.pmc_traffic("no_such_kernel")[2] is not None
    assert "no PMC summary of" in bench.pmc_traffic("w4_gemv", "no-such-run")[2]
