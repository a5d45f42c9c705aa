"""Measurement for the GPTQ quantiser row: seconds per Linear of Llama-2-7B shapes on the device (Hessian
accumulation, Cholesky, column-loop kernel, trailing updates), and the layer-output error against round-to-nearest."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lit_parrot_amd import _hip  # noqa: E402
from lit_parrot_amd.quantize.gptq import GPTQQuantizer, rtn_quantize  # noqa: E402


def main() -> None:
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    out = {}
    for name, N, K in (("attn.attn", 12288, 4096), ("attn.proj", 4096, 4096), ("mlp.fc_1", 11008, 4096), ("mlp.proj", 4096, 11008)):
        lin = torch.nn.Linear(K, N, bias=False)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(N, K, generator=g) * 0.02)
        lin = lin.to(torch.bfloat16).to(dev)
        # correlated calibration rows (a random low-rank mix + noise) so that the Hessian is not ~identity
        mix = torch.randn(256, K, generator=g) * 0.3
        X = [((torch.randn(512, 256, generator=g) @ mix) + 0.1 * torch.randn(512, K, generator=g)).to(torch.bfloat16).to(dev) for _ in range(8)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        qz = GPTQQuantizer(lin, bits=4, groupsize=128)
        for x in X:
            qz.collect_input_stats(None, (x.unsqueeze(0),), None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _hip.prof_begin()
        qmod, loss = qz.quantize()
        stats = _hip.prof_end()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        Wf = lin.weight.float()
        q, s, z = rtn_quantize(lin.weight.detach().cpu(), 128)
        gi = torch.arange(K) // 128
        Wr = ((q.float() - z.float()[:, gi]) * s.float()[:, gi]).to(dev)
        Xf = torch.cat(X).float()
        e_rtn = float(((Xf @ (Wr - Wf).t()) ** 2).mean())
        e_gptq = float(((Xf @ (qmod.get_weight(torch.float32) - Wf).t()) ** 2).mean())
        k = stats.get("gptq_block", (0.0, 0))
        out[name] = {"N": N, "K": K, "hessian_s": t1 - t0, "quantize_s": t2 - t1, "column_loop_kernel_ms": k[0], "column_loop_launches": k[1],
                     "gptq_loss": loss, "layer_output_mse_rtn": e_rtn, "layer_output_mse_gptq": e_gptq}
        print(name, json.dumps(out[name]), flush=True)
        del lin, X, qz, qmod
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
