"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain C + plain torch fp32) of the reference decode hot path.  Nothing under
``oracle/`` is part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.

Pin status (see DESIGN.md "Oracle"):
  * control pyramid (extractor / FDN / mask / flow normalise): pinned by goldens captured from the
    importable part of the reference (``oracle/make_goldens.py`` -> ``tests/golden/*.npz``).
  * forward splat kernel: the reference kernel is CUDA-only and has no fixtures -> hand-computed
    known-answer cases only ("parity unpinned" against a CUDA run).
  * diffusers SD-1.5 UNet / ControlNet / VAE / DDIM: diffusers is not installed and the reference holds
    no numeric fixtures for it -> restated from the published topology, "parity unpinned".
"""
