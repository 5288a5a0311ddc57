"""CPU oracle for the CVAE / AG-CVAE captioning training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vae_captioning_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may use it, and there only as the checker.

PARITY UNPINNED.  The reference (``/root/reference``, yiyang92/vae_captioning)
executes every FLOP inside TensorFlow 1.x and zhusuan 0.3.0, neither of which is
vendored, installed or installable here, and it ships no tests, golden vectors
or recorded outputs for this path.  This
oracle is therefore a *restatement* of the reference graph written from its
Python sources (each function cites the file:line it follows) plus the public
TF-1.4 semantics of the ops those lines call; the latter are flagged "TF-sem."
The only reference-derived known answers it is pinned against are the seed-42
cluster means (``utils/vae_utils.py:20-27``), the Q1 reshape row map
(``vae_model/decoder.py:109-110``) and the un_clusters id set
(``vae_model/decoder.py:56``); the hand-derived backward passes are pinned
against torch-CPU autograd and fp64 finite differences in ``tests/``.
"""
