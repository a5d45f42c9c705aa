#!/usr/bin/env python3
"""Decode benchmark of the HIP path (contract in the task statement; metric from BASELINE.json).

    python bench.py --gpus 1 --steps 256 --warmup 16
    python bench.py --gpus 4                       (spawns its own replicas, one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W     (an external launcher works too)

A *step* is one single-stream decode token of the workload (default: Llama-2-7B, GPTQ int4 g128, random-init
synthetic weights of that architecture, 128-token synthetic prompt): one replay of the captured hipGraph.  With N > 1
every replica is an independent process on its own GPU with its own prompt (seed 1234 + i) - "replicas only",
DESIGN.md §5: there is NO collective and no RCCL anywhere.  Without an external launcher the parent process (which never
touches a GPU) starts N children with HIP_VISIBLE_DEVICES=i, waits until all are warmed up, releases them together
over their stdin pipes (the start barrier) and collects one JSON line each; under torch.distributed.run the ranks meet
in a gloo (CPU) group for the same two barriers and the MAX / SUM of (seconds, tokens).  ``value`` = tokens decoded by
all replicas / max over replicas of the wall time of the K steps.

Besides the contract's fields the JSON line carries
  roofline      for the dominant kernel: algorithmic bytes per launch / its mean duration, measured live with HIP
                events taken from each dispatch (profiling sink of the library), against 8 TB/s;
  step_roofline the same for the whole token (all algorithmic bytes / wall time per token) - includes launch gaps;
  cpu_baseline  the CPU oracle (a port of the reference's generate()+model) timed on this host's cores: Pythia-160M fp32
                128 -> 64 tokens (BASELINE configs[0]) and a bounded sample of this workload's decode steps.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the pool's driver only supports dmabuf IPC (kept in every child env)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

import torch  # noqa: E402

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md; the headline figures with 2:1 sparsity are not used)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (config, quantization mode, prompt length, kv window, dtype label)
    "llama2-7b-int4": ("Llama-2-7b-hf", "gptq.int4-g128", 128, "u4 weights, bf16 activations, fp32 accumulate"),
    "llama2-7b-int8": ("Llama-2-7b-hf", "bnb.int8", 128, "int8"),
    "llama2-7b-nf4": ("Llama-2-7b-hf", "bnb.nf4", 128, "nf4 weights (16-entry codebook), bf16 activations, fp32 accumulate"),
    "stablelm-3b-bf16": ("stablelm-base-alpha-3b", None, 512, "bf16"),
    "falcon-40b-int4": ("falcon-40b", "gptq.int4-g128", 128, "u4 weights, bf16 activations, fp32 accumulate"),
    "pythia-160m-bf16": ("pythia-160m", None, 128, "bf16"),
    # the one model the reference publishes tokens/s for (tutorials/quantize.md:16-128, on its authors' GPU): bf16 / bnb.int8 / gptq.int4
    "falcon-7b-bf16": ("falcon-7b", None, 128, "bf16"),
    "falcon-7b-int8": ("falcon-7b", "bnb.int8", 128, "int8"),
    "falcon-7b-int4": ("falcon-7b", "gptq.int4", 128, "u4 weights (per-channel scales, the reference's gptq.int4), bf16 activations, fp32 accumulate"),
    "tiny-llama-int4": ("tiny-llama", "gptq.int4-g128", 16, "u4 weights, bf16 activations, fp32 accumulate"),
}


def linear_bytes(cfg, mode):
    """Algorithmic HBM bytes of each Linear per decoded token: every weight read once + quantisation
    metadata.  Returns {linear name: bytes} for one block plus lm_head."""
    out = {}
    for name, (n, k) in cfg.linear_shapes().items():
        if mode is None:
            out[name] = n * k * 2
        elif mode.startswith("gptq"):
            group = 128 if mode.endswith("g128") else k
            out[name] = n * k // 2 + n * (-(-k // group)) * 4  # packed nibbles + bf16 scale + bf16 zero per group
        elif mode == "bnb.int8":
            out[name] = n * k + n * 4  # int8 + fp32 SCB per row
        elif mode.startswith(("bnb.nf4", "bnb.fp4")):
            out[name] = n * k // 2 + n * (k // 64) * 4  # packed nibbles + fp32 absmax per block of 64
        else:
            raise ValueError(mode)
    return out


def token_bytes(cfg, mode, context):
    lb = linear_bytes(cfg, mode)
    weights = sum(v for k, v in lb.items() if k != "lm_head") * cfg.n_layer + lb["lm_head"]
    kv = 2 * cfg.n_query_groups * cfg.head_size * 2 * cfg.n_layer * context  # K and V rows 0..pos, GQA-native, bf16
    return weights, kv + cfg.n_embd * 2


def kernel_bytes_per_token(cfg, mode):
    """Algorithmic bytes handled by each kernel id of the library during one token (for the roofline object)."""
    lb = linear_bytes(cfg, mode)
    L = cfg.n_layer
    prefix = {None: "bf16_gemv", "bnb.int8": "w8_gemv"}.get(mode, "w4c_gemv" if (mode or "").startswith(("bnb.nf4", "bnb.fp4")) else "w4_gemv")
    single = [k for k in lb if k not in ("mlp.fc_1", "mlp.fc_2", "lm_head")]
    res = {prefix: (sum(lb[k] for k in single) * L + lb["lm_head"], len(single) * L + 1)}
    if "mlp.fc_1" in lb:
        dual = prefix if mode == "bnb.int8" else prefix + "_dual"
        b, n = res.get(dual, (0, 0))
        res[dual] = (b + (lb["mlp.fc_1"] + lb["mlp.fc_2"]) * L, n + L)
    return res  # {kernel name: (bytes per token, launches per token)}


PMC_KERNEL_PREFIX = {"w4_gemv": "w4_gemv_kernel<1, false", "w4_gemv_dual": "w4_gemv_kernel<1, true",
                     "bf16_gemv": "bf16_gemv_kernel<1, false", "bf16_gemv_dual": "bf16_gemv_kernel<1, true",
                     "w8_gemv": "w8_gemv_kernel", "attn_fused_decode": "attn_fused_decode_kernel", "eng_token": "eng_token_kernel"}


def pmc_traffic(kernel: str, run: str = "llama2-7b-int4-multilaunch"):
    """HBM bytes per launch of ``kernel`` in the run ``run`` (workload name + "-engine" or "-multilaunch": the executor)
    from the newest committed PMC summary (profiles/<tag>_<run>_pmc_traffic.json, written by
    tools/summarize_profiles.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this benchmark).
    Counters cannot be collected inside the timed run; returns (bytes per launch or None, source file, problem or None).
    A summary without a kernel of the expected name is STALE (the kernels changed since it was taken): that is reported as a
    problem, printed on stderr and asserted by tests/test_replicas_gloo.py - never papered over with an old number."""
    files = sorted((REPO / "profiles").glob("*_pmc_traffic.json"))
    prefix = PMC_KERNEL_PREFIX.get(kernel)
    if not files or prefix is None:
        return None, None, "no PMC summary for this kernel"
    newest = files[-1].name.split("_")[0]  # the round tag: only summaries of the newest measurement pass count
    f = REPO / "profiles" / f"{newest}_{run}_pmc_traffic.json"
    if not f.exists():
        return None, None, f"the {newest} measurement pass holds no PMC summary of {run}"
    entries = json.loads(f.read_text())["kernels"]
    tot_b = tot_n = 0.0
    for name, v in entries.items():
        if name.startswith(prefix) and "hbm_bytes_per_launch_corrected" in v:
            n = v.get("launches_FETCH_SIZE", 1)
            tot_b += v["hbm_bytes_per_launch_corrected"] * n
            tot_n += n
    if tot_n:
        return tot_b / tot_n, f.name, None
    return None, f.name, f"STALE: {f.name} holds no kernel named {prefix}*"


class Watchdog:
    """Phase markers on stderr and a host watchdog: a run that goes silent must say WHERE.  ``phase(name)`` prints one line
    per phase (model built / repacked / prefill / captured / warm-up / timed / profiled / cpu baseline); a daemon thread that
    sees no phase change (or ``touch()``) for ``limit_s`` seconds prints the phase and the stream engine's host-visible
    words (first error code, epoch of the last completed launch: pinned host memory, read without a HIP call - a stuck
    device queue cannot block this thread) and ends the process with status 3 through ``os._exit``: an exit, never a
    re-exec, and nothing that waits for the GPU."""

    def __init__(self, limit_s: float = 300.0, tag: str = "bench.py") -> None:
        import threading

        self.limit_s, self.tag = float(limit_s), tag
        self.t_start = self.t_last = time.monotonic()
        self.name = "start"
        self.words = None  # callable -> (first error code, last completed epoch), set once a stream engine exists
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._watch, daemon=True)
        if self.limit_s > 0:
            self._thread.start()

    def phase(self, name: str) -> None:
        now = time.monotonic()
        print(f"{self.tag}: phase '{name}' at +{now - self.t_start:.1f} s (previous: '{self.name}', {now - self.t_last:.1f} s)", file=sys.stderr, flush=True)
        self.name, self.t_last = name, now

    def touch(self) -> None:
        self.t_last = time.monotonic()

    def describe(self) -> str:
        txt = f"no progress for {time.monotonic() - self.t_last:.0f} s in phase '{self.name}'"
        if self.words is not None:
            try:
                err, epoch = self.words()
                txt += f"; stream engine host words: first error {err:#x}, last completed launch epoch {epoch}"
            except Exception as e:  # the report must not die on its way out
                txt += f"; stream engine host words unreadable ({e})"
        return txt

    def _watch(self) -> None:
        while not self._stop.wait(min(1.0, self.limit_s / 4)):
            if time.monotonic() - self.t_last > self.limit_s:
                print(f"{self.tag}: WATCHDOG: {self.describe()} - ending the process with status 3", file=sys.stderr, flush=True)
                os._exit(3)

    def stop(self) -> None:
        self._stop.set()


def committed_full_cpu_run(workload: str):
    """SURVEY 8(d) asks for >= 8 decode tokens of the workload on the host CPU; at ~33 s per token (the reference's CPU path
    dequantises every matrix on every call) that does not fit the default run, so the line carries the committed figure of
    the `--cpu-full` run beside the bounded sample, with its source file."""
    for f in sorted((REPO / "profiles").glob(f"*_{workload}_cpu-full_bench.json"), reverse=True):
        try:
            cb = json.loads(f.read_text())["cpu_baseline"]
            return {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"],
                    "source": f"profiles/{f.name} (bench.py --cpu-full, an earlier run on the same kind of box)"}
        except (OSError, KeyError, ValueError):
            continue
    return None


def rank_env():
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local, world


def max_over_ranks(seconds: float, units: int, world: int, device=None) -> tuple:
    """(max elapsed over ranks, total units over ranks) under an external launcher: CPU tensors over gloo (timing only)."""
    if world == 1:
        return seconds, units
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64)
    u = torch.tensor([units], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())


def barrier(world: int, device) -> None:
    """device work done, then (under an external launcher) the gloo barrier: no RCCL, no GPU buffer leaves its process"""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()


# ---------------------------------------------------------------------------------------------- self-contained replicas
def aggregate_replicas(lines):
    """One whole-job result from the replicas' JSON lines: tokens of all replicas / the slowest replica's wall time."""
    elapsed = max(r["elapsed_s"] for r in lines)
    units = sum(r["steps"] for r in lines)
    out = dict(lines[0])
    out.update(value=units / elapsed, n_gpus=len(lines), ms_per_step=elapsed / lines[0]["steps"] * 1e3,
               replicas=[{"device": r["device"], "value": r["value"], "ms_per_step": r["ms_per_step"],
                          "roofline_frac": (r.get("roofline") or {}).get("frac"),
                          "step_roofline_frac": (r.get("step_roofline") or {}).get("frac")} for r in lines])
    out["config"] = dict(out["config"], replicas=len(lines), parallelism=f"replicas x{len(lines)} (no collective, no RCCL)")
    for k in ("elapsed_s", "device"):
        out.pop(k, None)
    return out


def spawn_replicas(args, argv, child_cmd=None):
    """Parent of ``--gpus N`` without an external launcher.  Never initialises a GPU.  Children: fresh interpreters with
    HIP_VISIBLE_DEVICES pinned (``--devices 0,0`` overrides, e.g. two replicas on a one-GPU box), that print READY when
    warmed up, block on stdin until released together, and print their JSON line."""
    import subprocess

    n = args.gpus
    devs = [d.strip() for d in args.devices.split(",")] if args.devices else [str(i) for i in range(n)]
    if len(devs) != n:
        raise SystemExit(f"--devices lists {len(devs)} devices for --gpus {n}")
    procs = []
    for i in range(n):
        env = dict(os.environ, HIP_VISIBLE_DEVICES=devs[i], PARROT_BENCH_REPLICA=str(i), PARROT_BENCH_WORLD=str(n),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        cmd = (child_cmd or [sys.executable, str(Path(__file__).resolve())]) + argv
        procs.append(subprocess.Popen(cmd, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1))

    def read_until(p, word):
        for line in p.stdout:
            if line.strip() == word:
                return True
        return False

    try:
        for i, p in enumerate(procs):
            if not read_until(p, "READY"):
                raise SystemExit(f"replica {i} ended before it was ready (exit code {p.wait()})")
        for p in procs:  # the start barrier: everybody is warmed up and synchronised; release them together
            p.stdin.write("GO\n")
            p.stdin.flush()
        lines = []
        for i, p in enumerate(procs):
            got = [json.loads(l) for l in p.stdout if l.startswith("{")]
            if p.wait() != 0 or not got:
                raise SystemExit(f"replica {i} failed (exit code {p.returncode})")
            lines.append(got[-1])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return aggregate_replicas(lines)


def cpu_pythia_leg():
    """BASELINE.json configs[0] / SURVEY §8(d): Pythia-160M fp32, greedy 128 -> 64 tokens through the oracle's generate()
    (the reference's loop, restated) on the host cores.  tokens/s as the reference defines it (new tokens / wall, prefill
    included, generate/base.py:239-255)."""
    from lit_parrot_amd.config import Config
    from lit_parrot_amd.synth import synthetic_prompt, synthetic_state_dict
    from oracle import model as om

    cfg = Config.from_name("pythia-160m")
    oracle = om.OracleGPT(cfg, synthetic_state_dict(cfg, 1234))
    prompt = synthetic_prompt(cfg, 128, 1234)
    t0 = time.perf_counter()
    y = om.generate(oracle, prompt, 192, 192, greedy_ties_lowest=True)
    el = time.perf_counter() - t0
    return {"value": (y.numel() - 128) / el, "unit": "tokens/s", "seconds": el, "sample": "pythia-160m fp32, 128-token prompt + 64 greedy tokens, prefill included"}


def cpu_baseline(cfg, mode, model, prompt_cpu, budget_s: float = 20.0, min_tokens: int = 1, wd=None, workload: str = ""):
    """Time the CPU oracle (port of the reference path) on the host cores: the Pythia-160M leg always, then a bounded sample
    of THIS workload - a short prompt prefix, then single-token decode steps until ~budget_s of CPU work, at least
    ``min_tokens`` of them (``--cpu-full`` asks for the 8 that SURVEY §8(d) names: ~33 s each on Llama-2-7B, because the
    reference's CPU path dequantises every matrix on every call, quantize/gptq.py:263 - keeping the dequantised matrices
    across steps would no longer be the reference's path).  Returns the cpu_baseline object."""
    from oracle import model as om

    out = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port", "pythia_160m_fp32": cpu_pythia_leg()}
    full = committed_full_cpu_run(workload)
    if full is not None:
        out["full_run"] = full
    if wd is not None:
        wd.phase("cpu baseline: workload leg")
    tile_cols = 128 if (mode or "").endswith("g128") else -1
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    if mode == "bnb.int8" or (mode or "").startswith(("bnb.nf4", "bnb.fp4")):
        out["sample"] = "workload leg skipped: the state dict of this mode holds only quantised weights"
        return out
    oracle = om.OracleGPT(cfg, sd, "gptq" if mode and mode.startswith("gptq") else "dense", tile_cols=tile_cols)
    T0 = 4
    S = T0 + 64
    with torch.no_grad():
        pos = torch.arange(T0)
        logits = oracle(prompt_cpu[:T0].view(1, -1), S, pos)
        n, t0 = 0, time.perf_counter()
        while True:
            tok = logits[0, -1].float().argmax().view(1, 1)
            pos = pos[-1:] + 1
            logits = oracle(tok, S, pos)
            n += 1
            el = time.perf_counter() - t0
            print(f"bench.py: cpu_baseline token {n} after {el:.0f} s", file=sys.stderr, flush=True)  # a long leg stays visibly alive
            if wd is not None:
                wd.touch()
            if (el > budget_s and n >= min_tokens) or n >= 60:
                break
    out.update(value=n / el, sample=f"{n} single-token decode steps after a {T0}-token prompt ({el:.1f} s), same weights, oracle/model.py on the host CPU")
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="llama2-7b-int4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline: at least 8 decode tokens of the workload (minutes on Llama-2-7B)")
    ap.add_argument("--devices", default="", help="comma list of HIP_VISIBLE_DEVICES values for the replicas spawned by --gpus N (default 0..N-1)")
    ap.add_argument("--attn-split-keys", type=int, default=0, help="A/B: window slots per sequence split of the decode attention (default: the library's)")
    ap.add_argument("--engine", type=int, default=-1, help="1 / 0: force the one-launch stream engine on / off (default: the library's choice)")
    ap.add_argument("--no-sampled", action="store_true", help="skip the sampled-decode leg (generate's default call: temperature 0.8, top_k 200)")
    ap.add_argument("--layers", type=int, default=0, help="diagnostic: build the workload's model with this many blocks (a cache-residency probe; the line is marked and is not the metric)")
    ap.add_argument("--watchdog", type=float, default=300.0, help="seconds without a phase change before the run reports where it is stuck and exits 3 (0: off)")
    args = ap.parse_args()

    rank, local, world = rank_env()
    replica = os.environ.get("PARROT_BENCH_REPLICA")
    if replica is None and world == 1 and args.gpus > 1:
        # no external launcher: this process becomes the parent of N fresh replicas and never touches a GPU
        argv = [a for a in sys.argv[1:]]
        print(json.dumps(spawn_replicas(args, argv)), flush=True)
        return
    if replica is not None:  # a child of spawn_replicas: one replica on the one device it can see
        rank, local, world = int(replica), 0, 1
    elif world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without a launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the decode path has no CPU fallback")
    if replica is None and args.devices:  # external launcher with an explicit device map (e.g. two ranks on a one-GPU box)
        local = int(args.devices.split(",")[local])
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")  # timing barriers and two scalar reductions on CPU tensors: no RCCL

    import lit_parrot_amd as L
    from lit_parrot_amd import _hip
    from lit_parrot_amd.config import Config
    from lit_parrot_amd.generate.base import _session
    from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt

    if args.engine >= 0:
        from lit_parrot_amd.generate import base as _gb

        _gb.ENGINE_DEFAULT = bool(args.engine)
    if args.attn_split_keys > 0:
        from lit_parrot_amd import ops as _ops

        _ops.ATTN_SPLIT_KEYS = args.attn_split_keys
    wd = Watchdog(args.watchdog, tag=f"bench.py[{rank}]")
    cfg_name, mode, T, dtype_label = WORKLOADS[args.workload]
    cfg = Config.from_name(cfg_name)
    if args.layers:
        cfg.n_layer = args.layers
    total = T + args.warmup + args.steps + 1
    assert total <= cfg.block_size, "prompt + warmup + steps must fit block_size"
    t_build = time.perf_counter()
    wd.phase("building the synthetic model")
    model = build_synthetic_model(cfg, mode, seed=1234, device=device)
    prompt = synthetic_prompt(cfg, T, seed=1234 + rank, device="cpu")
    torch.cuda.synchronize(device)
    t_build = time.perf_counter() - t_build

    with torch.no_grad():
        wd.phase("session (kernel-layout repacks of the executor, KV caches)")
        sess = _session(model, total, total, greedy=True)
        if sess.eng is not None:
            wd.words = sess.eng.progress
        wd.phase("first prefill (lazy W4K repacks)")
        # one untimed prefill first: the int4 weights are repacked to the kernel layout lazily at first use (a one-time
        # load cost, 161 repack launches), workspaces are allocated, code objects are loaded
        sess.prefill(prompt.to(device))
        torch.cuda.synchronize(device)
        wd.phase("timed prefill")
        t_pre = time.perf_counter()
        logits = sess.prefill(prompt.to(device))
        L.ops.argmax_advance(logits, sess.tokens, sess.pos)
        torch.cuda.synchronize(device)
        t_pre = time.perf_counter() - t_pre
        wd.phase("graph capture")
        sess.capture()
        wd.phase(f"warm-up, {args.warmup} steps")
        for _ in range(args.warmup):
            sess.step()
        barrier(world, device)
        sess.check_error()  # an executor whose in-launch waits gave up decodes garbage FASTER than a healthy one: never time it
        if replica is not None:  # spawned replica: tell the parent, then wait to be released together with the others
            print("READY", flush=True)
            if sys.stdin.readline().strip() != "GO":
                raise SystemExit("replica: released without GO")
        wd.phase(f"timed region, {args.steps} steps")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sess.step()
        barrier(world, device)
        elapsed = time.perf_counter() - t0
        sess.check_error()  # (after the clock stopped: the check syncs)
        elapsed_max, units = max_over_ranks(elapsed, args.steps, world, device)
        pos_end = int(sess.pos.item())
        assert pos_end == T + args.warmup + args.steps, (pos_end, T, args.warmup, args.steps)

        # per-kernel durations: the same step, launched eagerly with every dispatch bracketed by HIP events
        wd.phase("per-kernel HIP events")
        prof_steps = 8
        _hip.prof_begin()
        for _ in range(prof_steps):
            sess._step()
        stats = _hip.prof_end()
        sess.check_error()

        # the reference's DEFAULT call samples (generate/base.py:165-166: temperature 0.8, top_k 200): the same model and
        # prompt with the sampling step inside the captured graph (torch's noise draw + parrot_topk_sample)
        sampled = None
        if not args.no_sampled:
            wd.phase("sampled decode (temperature 0.8, top_k 200)")
            n_s = min(args.steps, 128)
            s2 = _session(model, total, total, greedy=False, sampler=(0.8, 200))
            torch.manual_seed(1234)
            lg = s2.prefill(prompt.to(device))
            s2.sample(lg)
            s2.capture()
            for _ in range(8):
                s2.step()
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(n_s):
                s2.step()
            torch.cuda.synchronize(device)
            el_s = time.perf_counter() - t1
            s2.check_error()
            assert int(s2.pos.item()) == T + 8 + n_s, int(s2.pos.item())
            sampled = {"value": n_s / el_s, "steps": n_s, "ms_per_step": el_s / n_s * 1e3, "temperature": 0.8, "top_k": 200,
                       "engine": s2.eng is not None}
    wd.phase("summary")

    ms_per_step = elapsed_max / args.steps * 1e3
    ctx_mean = T + args.warmup + args.steps / 2.0
    w_bytes, kv_bytes = token_bytes(cfg, mode, ctx_mean)
    kb = kernel_bytes_per_token(cfg, mode)
    kb["eng_token"] = (w_bytes + kv_bytes, 1)  # the stream engine: the whole token is one launch
    dom = max(stats, key=lambda k: stats[k][0])
    kernels = {k: {"avg_us": v[0] / v[1] * 1e3, "launches_per_token": v[1] / prof_steps, "ms_per_token": v[0] / prof_steps}
               for k, v in sorted(stats.items(), key=lambda kv: -kv[1][0])}
    roofline = None
    if dom in kb:
        bytes_per_launch = kb[dom][0] / kb[dom][1]
        avg_s = stats[dom][0] / stats[dom][1] * 1e-3
        achieved = bytes_per_launch / avg_s / 1e9
        run = args.workload + ("-engine" if sess.eng is not None else "-multilaunch")
        traffic, traffic_src, traffic_problem = pmc_traffic(dom, run)
        if traffic_problem:
            print(f"bench.py: roofline.traffic unavailable - {traffic_problem}", file=sys.stderr, flush=True)
        roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_problem": traffic_problem,
                    "bytes_per_launch": bytes_per_launch,
                    "avg_launch_us": avg_s * 1e6, "launches_per_token": kb[dom][1]}
    step_gbs = (w_bytes + kv_bytes) / (ms_per_step * 1e-3) / 1e9
    prefill_flops = 2.0 * cfg.n_linear_params() * T - 2.0 * cfg.padded_vocab_size * cfg.n_embd * (T - 1)

    result = {
        "metric": "decode tokens/s (single-stream per GPU, independent replicas)",
        "value": units / elapsed_max,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype_label,
        "data": "synthetic",
        "config": {"workload": f"{cfg_name} {mode or 'bf16'} single-stream decode, {T}-token prompt, random-init weights"
                   + (f" - DIAGNOSTIC: {args.layers} of the model's blocks (not the metric)" if args.layers else ""),
                   "prompt_tokens": T, "replicas": world, "parallelism": f"replicas x{world} (no collective, no RCCL)",
                   "graph": "hipGraph replay per token"},
        "roofline": roofline,
        "step_roofline": {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": step_gbs / HBM_PEAK_GBS, "bytes_per_token": w_bytes + kv_bytes,
                          "weight_bytes": w_bytes, "kv_bytes_mean_context": kv_bytes},
        "kernels": kernels,
        # the reference's own definition (generate/base.py:239-255): new tokens / wall time INCLUDING the prefill
        "reference_definition_tokens_per_s": (args.warmup + args.steps + 1) / (t_pre + (args.warmup + args.steps) * ms_per_step * 1e-3),
        "prefill_ms": t_pre * 1e3,
        "prefill_tokens_per_s": T / t_pre,
        # the prompt's Linears against the dense bf16 matrix peak (int4 / int8 weights are multiplied as bf16 / int8 MFMA operands);
        # whole-prefill wall time, attention and norms included in the denominator, lm_head counted for the last row only
        "prefill_roofline": {"bound": "mfma", "achieved": prefill_flops / t_pre / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": prefill_flops / t_pre / 1e12 / MFMA_BF16_PEAK_TFLOPS, "linear_flops": prefill_flops},
        "build_s": t_build,
        "engine": sess.eng is not None,
        # generate()'s default call (temperature 0.8, top_k 200) with the sampling step inside the graph; greedy = `value`
        "sampled_tokens_per_s": None if sampled is None else sampled["value"],
        "sampled": sampled,
    }
    if replica is not None:  # the parent aggregates: it needs this replica's own clock
        result.update(elapsed_s=elapsed, device=os.environ.get("HIP_VISIBLE_DEVICES", "?"), cpu_baseline=None)
        if rank == 0 and not args.no_cpu_baseline and int(os.environ.get("PARROT_BENCH_WORLD", "1")) == 1:
            result["cpu_baseline"] = cpu_baseline(cfg, mode, model, prompt, args.cpu_budget, 8 if args.cpu_full else 1, wd, args.workload)
        wd.stop()
        print(json.dumps(result), flush=True)
        return
    if rank == 0:
        if args.no_cpu_baseline or world > 1:
            result["cpu_baseline"] = None
        else:
            result["cpu_baseline"] = cpu_baseline(cfg, mode, model, prompt, args.cpu_budget, 8 if args.cpu_full else 1, wd, args.workload)
        print(json.dumps(result), flush=True)
    wd.stop()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
