"""bot_amd — MI355X-native full-batch GNN message passing (GAT / GCN forward + backward).

One hot path of AiRyunn/BoT, rebuilt for gfx950: the `GraphConv` / `GATConv` layer surface of the
reference's src/no-sampling/models.py on top of hand-written HIP kernels (CSR/CSC SpMM, SDDMM,
fused attention softmax) reached through the C ABI in include/bot_gnn.h.  Importing this package
loads bot_amd/lib/libbot_gnn.so and fails if it has not been built — there is no CPU fallback.
"""
from . import _C  # noqa: F401  (fails loudly when the HIP library is missing)
from . import function, ops
from .errors import DGLError
from .graph import Graph, add_self_loop, graph, preprocess, remove_self_loop, reorder_graph, to_bidirected
from .ops import edge_softmax

__all__ = ["Graph", "graph", "to_bidirected", "add_self_loop", "remove_self_loop", "preprocess", "reorder_graph", "function", "ops",
           "edge_softmax", "DGLError"]
__version__ = "0.1.0"
