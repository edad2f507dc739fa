import copy, sys, torch
sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import torch.nn.functional as F
from bot_amd import nn as bnn, synth, train as T
DEV = "cuda"
ds = synth.make_dataset("arxiv", device=DEV, seed=0, scale=0.2)
g, C = ds.graph, ds.n_classes
g.create_formats_()
mask = torch.rand(ds.train_idx.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)) < 0.5
kw = dict(use_labels=True, loss="loge", n_classes=C)
import os
if os.environ.get("DBG_ADDBIAS"):
    from bot_amd import ops
    def fwd(self, x):
        return ops.add_bias(x, self.bias)
    bnn.ElementWiseLinear.forward = fwd
def make():
    torch.manual_seed(0)
    m = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=64,
                norm="batch", dropout=0.0, input_drop=0.0, attn_drop=0.0, linear=True).to(DEV)
    return m, torch.optim.RMSprop(m.parameters(), lr=0.002, capturable=True)
m1, o1 = make(); m2, o2 = make()
m2.load_state_dict(copy.deepcopy(m1.state_dict()))
cap = T.captured_train_step(m2, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o2, warmup=3, mask=mask, **kw)
for _ in range(3):
    T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw)
for _ in range(3):
    T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw); cap()
torch.cuda.synchronize()
print("state equal", all(torch.equal(a, b) for a, b in zip(m1.state_dict().values(), m2.state_dict().values())))
import os
mode = os.environ.get("DBG_MODE", "none")
b1, b2 = m1.biases[0].bias, m2.biases[0].bias
for it in range(8):
    if mode == "sync":
        torch.cuda.synchronize()
    elif mode == "clone":
        x = b2.grad.detach().clone()
    elif mode == "evalflag" and it == 3:
        m2.eval()
    elif mode == "evalflag1" and it == 3:
        m1.eval()
    le, pe = T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw)
    lc, pc = cap()
    print(it, "loss", torch.equal(le, lc), "pred", torch.equal(pe, pc), "bias", float((b1 - b2).abs().max()), "grad", float((b1.grad - b2.grad).abs().max()),
          "m1.training", m1.training, "m2.training", m2.training)
