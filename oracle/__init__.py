"""CPU oracle for the embed + exhaustive-kNN + retrieval-metric path.

TEST INFRASTRUCTURE ONLY: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the checker.
The product package never imports it and fails loudly without its HIP library.

Pinning status (DESIGN.md "Oracle"):
  * metric tail (oracle/metrics.py): pinned by golden vectors produced by the reference's
    own functions (tests/golden/make_golden.py, run in the build container).
  * exhaustive search (oracle/search.py, search_ref.c): the reference has no test or
    golden vector for it and its fp32 library calls have no defined tie order; the oracle
    defines fp64 scores + lowest-id tie break and is cross-checked against the
    reference's own fp32 ranking on the golden sets (identical except inside fp32
    near-ties, which the test lists).
  * DenseNet-121 (oracle/densenet.py): backbone parity UNPINNED -- torchvision (the
    reference's dependency, not vendored and not installed) defines the arithmetic; the
    restatement is checked by known answers only (parameter count, tensor names, shapes).
"""
