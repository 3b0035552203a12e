"""CPU restatement of the reference's registration hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``pointcloudprocessing_amd`` may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.
"""
