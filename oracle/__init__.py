"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Grad-TTS / DiffVC sampling path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.  The product path
(``speech-backbones_amd``) never imports this package and fails loudly when its HIP library is missing.

Parity status: the reference ships no golden vectors of its own, so the
restatement is pinned against the reference's Python run on CPU in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``, and ``tests/test_oracle_vs_reference.py``
which runs whenever ``/root/reference`` is mounted).
"""
