"""CPU oracle for the mmseq Gibbs hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / timed CPU baseline.  Nothing under
``mmseq_amd/`` imports it; the product path fails loudly when its HIP library is missing
instead of falling back to anything here.

``oracle.binding``     ctypes view of ``liboracle.so`` (C restatement, mmseq_oracle.c)
``oracle.host_oracle`` numpy / pure-Python restatements of the host-side pieces
                       (hits-file codec, ingest/collapse, summaries)
"""
