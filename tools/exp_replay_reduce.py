"""Round 6: is torch's multi-workgroup reduction wrong under hipGraph REPLAY on its own (round 3 saw the bias gradient `dy.sum(0)` over [N, 40]
drift from the fourth replay of a captured train step on), or only inside that step?  A captured graph of nothing but the
reduction (fresh input written before every replay), sizes that make ATen split one output over several workgroups, 12 replays each,
compared with the eager result of the same input.    python tools/exp_replay_reduce.py"""
import torch

dev = "cuda"
torch.manual_seed(0)
for (n, f) in ((33868, 40), (169343, 40), (169343, 750), (2449029, 47)):
    x = torch.randn(n, f, device=dev)
    out = torch.empty(f, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            out.copy_(x.sum(0))
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out.copy_(x.sum(0))
    bad = []
    for it in range(12):
        x.copy_(torch.randn(n, f, device=dev))
        g.replay()
        torch.cuda.synchronize()
        ref = x.sum(0)
        if not torch.equal(out, ref):
            bad.append((it, float((out - ref).abs().max() / ref.abs().max())))
    print(f"sum(0) of [{n}, {f}] replayed 12 times: {'all bitwise equal to eager' if not bad else 'DIFFERS at replays ' + str(bad)}")
    # the scalar form (loss numerators): sum over everything
    s = torch.empty((), device=dev)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        s.copy_(x.sum())
    bad = []
    for it in range(12):
        x.copy_(torch.randn(n, f, device=dev))
        g2.replay()
        torch.cuda.synchronize()
        if not torch.equal(s, x.sum()):
            bad.append(it)
    print(f"sum() of [{n}, {f}] replayed 12 times: {'all bitwise equal to eager' if not bad else 'DIFFERS at replays ' + str(bad)}")
