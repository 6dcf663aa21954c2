"""CPU oracle for the ir2rgb hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package; the product (``ir2rgb_amd``) never does.
"""
