"""CPU oracle: a plain numpy / C restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``isaaclab_amd/`` imports this package; it is used by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` as the checker /
baseline, never as the product path.  Every function cites the reference lines it follows
(LAB = /root/reference/source/isaaclab/isaaclab, TASKS = .../source/isaaclab_tasks/isaaclab_tasks).

Pinning: the restatement is checked against golden vectors produced by importing the reference itself in
the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and against the reference's
own known-answer tests (``tests/test_oracle_math.py``, ``tests/test_oracle_terms.py``).  The ray cast has no runnable reference
(warp-lang is absent): ray-hit parity is UNPINNED by the reference and is defined against the brute-force
closest-hit routine in ``raycast_ref.c``, a self-contained C file that shares no source with the product (see its header).
"""
