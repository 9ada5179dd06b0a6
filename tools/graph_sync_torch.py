"""Control experiment: the same replay / device-synchronize pattern on a PURE PyTorch graph (no kernel of this library)."""
import torch
dev = "cuda"
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 64)).to(dev)
x = torch.randn(4096, 256, device=dev)
y = torch.randn(4096, 64, device=dev)


def fn():
    for p in m.parameters():
        p.grad = None
    loss = ((m(x) - y) ** 2).mean()
    loss.backward()
    idx = torch.randint(0, 4096, (1024,), device=dev)                 # graph-safe generator
    return loss + 0.0 * x[idx].sum() + sum(p.grad.abs().sum() for p in m.parameters())


st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for _ in range(3):
        fn()
torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
vals = []
for _ in range(3):
    g.replay()
vals.append(float(out.detach()))
for r in range(4):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    vals.append(float(out.detach()))
print("pure torch", "SAME" if len(set(vals)) == 1 else "DIFFERENT", vals, flush=True)
