"""CPU oracle for the fdiff score-matching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``fourierdiffusion_amd``) never imports this package and fails loudly when
the HIP library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
(/root/reference, this container only) and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every oracle function against those
fixtures.
"""
