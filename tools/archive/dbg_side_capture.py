"""Which arm of test_captured_step_twelve_replays_bitwise[arxiv] deviates with bot_amd.side on: serial eager (reference), eager with the
side stream, captured with the side stream.  Prints the first differing tensors per step."""
import copy
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import nn as bnn, side, synth, train as T  # noqa: E402

DEV = "cuda"
hid = int(os.environ.get("HID", "64"))
ds = synth.make_dataset("arxiv", device=DEV, seed=0, scale=0.2)
g, C = ds.graph, ds.n_classes
g.create_formats_()
mask = torch.rand(ds.train_idx.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)) < 0.5
kw = dict(use_labels=True, loss="loge", n_classes=C)


def make():
    torch.manual_seed(0)
    m = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=hid,
                norm="batch", dropout=0.0, input_drop=0.0, attn_drop=0.0, linear=True).to(DEV)
    return m, torch.optim.RMSprop(m.parameters(), lr=0.002, capturable=True)


m0, o0 = make()
m1, o1 = make()
m2, o2 = make()
m3, o3 = make()
for m in (m1, m2, m3):
    m.load_state_dict(copy.deepcopy(m0.state_dict()))


def step(m, o, on):
    side.ENABLED = on
    return T.train_step(m, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o, mask=mask, **kw)


side.ENABLED = os.environ.get("CAP_SIDE", "1") == "1"
cap = T.captured_train_step(m2, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o2, warmup=3, mask=mask, **kw)
for _ in range(3):
    step(m0, o0, False)
    step(m1, o1, True)
    step(m3, o3, True)


def diff(tag, ma, mb):
    bad = [(k, float((a.grad - b.grad).abs().max())) for (k, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters())
           if not torch.equal(a.grad, b.grad)]
    badp = [k for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()) if not torch.equal(a, b)]
    print(tag, "grads differ:", bad[:6], "| state differs:", badp[:6], flush=True)


print("after warm-up: state m0 vs m1 / m0 vs m2")
diff("  eager-side", m0, m1)
diff("  captured  ", m0, m2)
for it in range(4):
    l0, _ = step(m0, o0, False)
    l1, _ = step(m1, o1, True)
    step(m3, o3, True)
    l2, _ = cap()
    torch.cuda.synchronize()
    print(it, float(l0), float(l1), float(l2), "forks", side.FORKS)
    diff("  eager-side", m0, m1)
    diff("  side vs side", m1, m3)
    diff("  captured  ", m0, m2)
