"""TEST INFRASTRUCTURE ONLY — CPU oracle for the full-batch GAT/GCN message-passing path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker — never as the thing measured or shipped.  ``bot_amd`` never
imports this package and has no CPU compute path.

Parity status:

* module logic (``GraphConv`` / ``GATConv`` / ``GCN`` / ``GAT`` / proteins ``GATConv``,
  ``add_labels`` / ``compute_loss`` / ``train``): PINNED — ``tests/golden/*.npz`` were produced by
  executing the reference's own ``src/no-sampling/models.py`` and ``run.py`` (imported from
  /root/reference, unmodified) in the authoring container; ``oracle/ref_models.py`` must reproduce them.
* ``copy_u_sum`` + degrees + "both" normalisation: PINNED by the ``GraphConv`` docstring
  known-answer rows (reference ``src/no-sampling/models.py:186-209``).
* ``edge_softmax``, ``u_mul_e_sum``, ``u_add_v``, ``copy_e_sum``, the ``eids`` variant, every backward,
  and the edge order produced by ``to_bidirected``: PARITY UNPINNED.  The arithmetic lives in the
  third-party dependency ``dgl 0.5.*`` (reference ``README.md:9``), which is neither vendored in
  /root/reference nor installable here; ``oracle/ref_ops.py`` restates DGL's published operator
  contract and is anchored on the reference's call sites only.
"""
