"""CPU oracle for the SKOOTS volumetric-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / reported baseline.  The
product path (``skoots_amd``) never imports this package and fails loudly when
the HIP library is missing.
"""
