"""Which arithmetic torch's device ops of the reference's sampling step use (generate/base.py:136-144): bf16 / python float,
softmax on bf16, multinomial == argmax(probs / exponential_) with the same generator consumption; graph capture of exponential_."""
import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = (torch.randn(32000, device=dev) * 3).to(torch.bfloat16)
for T in (0.8, 0.7, 1.3, 2.0):
    a = x / T
    b = (x.float() * (1.0 / torch.tensor(T, dtype=torch.float32))).to(torch.bfloat16)
    b2 = (x.float() * torch.tensor(1.0 / T, dtype=torch.float32).to(dev)).to(torch.bfloat16)
    c = (x.float() / torch.tensor(T, dtype=torch.float32, device=dev)).to(torch.bfloat16)
    print(f"T={T}: x/T == x*f32(1/T) {bool(torch.equal(a, b.to(dev)))} | == x*f32(1/T as double->f32) {bool(torch.equal(a, b2))} | == true division {bool(torch.equal(a, c))}")
lg = x / 0.8
v, _ = torch.topk(lg, 200)
m = torch.where(lg < v[[-1]], -float("Inf"), lg)
p = torch.softmax(m, dim=-1)
pf = torch.softmax(m.float(), dim=-1)
print("softmax(bf16) == bf16(softmax(fp32 input))", bool(torch.equal(p, pf.to(torch.bfloat16))), "nonzero probs", int((p > 0).sum()), "kept", int((m > -float('inf')).sum()))
mx = m.float().max()
e = torch.exp(m.float() - mx)
for name, s in (("sum fp32", e.sum()), ("sum fp64", e.double().sum().float())):
    q = (e / s).to(torch.bfloat16)
    print(f"  exp(x-max)/({name}) -> bf16 equals torch softmax: {bool(torch.equal(q, p))} ({int((q != p).sum())} differ)")
# multinomial vs argmax(p / exponential_)
same = 0
for seed in range(50):
    torch.manual_seed(seed)
    a = torch.multinomial(p, num_samples=1)
    torch.manual_seed(seed)
    q = torch.empty_like(p).exponential_(1)
    b = torch.argmax(p / q, dim=-1, keepdim=True)
    same += int(a.item() == b.item())
print("multinomial(p, 1) == argmax(p / empty_like(p).exponential_(1)) with the same seed:", same, "/ 50")
torch.manual_seed(5)
q = torch.empty_like(p).exponential_(1)
r = p / q
rf = (p.float() / q.float()).to(torch.bfloat16)
print("p / q (bf16) == bf16(fp32 div)", bool(torch.equal(r, rf)), "q dtype", q.dtype, "min q", float(q.min()), "zeros in q", int((q == 0).sum()))
# ties: argmax returns the first
t = torch.zeros(1000, device=dev, dtype=torch.bfloat16); t[[17, 400, 999]] = 1
print("argmax tie ->", int(torch.argmax(t)))
# graph capture of exponential_ reproduces the eager draws
buf = torch.empty_like(p)
torch.manual_seed(11)
eager = [buf.exponential_(1).clone() for _ in range(4)]
torch.manual_seed(11)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    buf.exponential_(1)  # warm-up draw 0
torch.cuda.current_stream().wait_stream(s)
first = buf.clone()
with torch.cuda.graph(g):
    buf.exponential_(1)
outs = [first]
for _ in range(3):
    g.replay()
    outs.append(buf.clone())
print("graph-replayed exponential_ == eager sequence:", [bool(torch.equal(a, b)) for a, b in zip(eager, outs)])
