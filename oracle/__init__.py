"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the reference's hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import anything from this package -- and only as the checker.  The
product (``se3et_amd``) never imports it and has no CPU fallback.
"""
