"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch fp32 ops) of the reference hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package, and only as the checker / reported baseline.  The product package (``multimodal-av-model_amd``)
never imports it and fails loudly when its HIP library is missing.

Parity status: PINNED — ``tests/golden/make_golden.py`` imports the reference from ``/root/reference`` in the
build container, loads the same seeded weights into the reference modules and into this restatement, checks
them against each other and writes the small fixtures under ``tests/golden/`` that the CPU tests re-check.
"""
