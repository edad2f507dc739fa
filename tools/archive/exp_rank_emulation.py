"""Experiment: what ONE rank of a W-way partitioned config-2 step costs (GPU time and host enqueue time), measured on a
single GPU: rank 0's block of the W-way partition is built as usual, the halo all-to-all is replaced by a local fill of the
receive buffer (wrong data, right shapes), the all-reduces run on a 1-rank RCCL group.  Communication time is NOT included:
this is the compute + launch floor of the scaling curve.

    python tools/exp_rank_emulation.py 8
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.nn.functional as F

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from bot_amd import synth, tuning, dist as bdist
from bot_amd import nn as bnn
from bot_amd import workloads, train as T
tuning.enable()
real_a2a = dist.all_to_all_single
def fake_a2a(out, inp, out_splits=None, in_splits=None, group=None):
    out.zero_()
dist.all_to_all_single = fake_a2a
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
C = ds.n_classes
torch.manual_seed(0)
model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **workloads.ARXIV_GAT).to(dev)
opt = torch.optim.RMSprop(model.parameters(), lr=0.002, capturable=True)
part = bdist.partition_dataset(ds, 0, W, dev)
model = bdist.wrap_model(model)
def step():
    return bdist.train_step(model, part, opt, use_labels=True, mask_rate=0.5, loss="loge", n_classes=C)
for _ in range(5): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
if os.environ.get("BOT_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(K): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
print(f"world {W}: rank 0 owns {part.n_owned} nodes, {part.n_edges} edges, halo {part.graph.halo.n_halo} rows, sends {part.graph.halo.n_send} rows")
print(f"per step (eager): wall {t_all / K * 1e3:.2f} ms, host enqueue {t_host / K * 1e3:.2f} ms")
cap = T.CapturedTrainStep(lambda: step(), dev)          # the same step as ONE hipGraph replay (RCCL all-reduces captured)
for _ in range(3): cap()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): cap()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"per step (hipGraph replay): wall {t_all / K * 1e3:.2f} ms, host enqueue {t_host / K * 1e3:.2f} ms")
dist.destroy_process_group()
