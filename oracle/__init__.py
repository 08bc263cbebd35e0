"""CPU oracle for the MQGAN PreEncoder VQ-GAN hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this package, and only
as the checker / reported CPU baseline.  ``mqgan_amd`` never imports it and has
no CPU fallback: its ops raise if the HIP library is missing.

What is here
------------
* ``oracle.nn``         - plain-PyTorch (CPU, fp32) functional restatement of
                          every module on the path (SURVEY.md section 8a rows Q1,
                          Q2, P1-P6, D1, D2, L1-L3), written against parameter
                          dicts that use the reference's state_dict key names.
* ``oracle.train_step`` - the optimisation step of train.py:380-501 + Adam /
                          clip / LambdaLR (rows T1-T3) on top of ``oracle.nn``.
* ``oracle.vq``         - numpy restatement of FSQ and of the generic L2
                          nearest-neighbour search (row Q3), plus a ctypes
                          wrapper over ``oracle/vq_ref.c`` (same arithmetic in C,
                          used for large N).

Pinning: every function is checked in ``tests/test_oracle_*.py`` against the
golden fixtures in ``tests/golden/`` that ``tests/golden/gen_golden.py``
produced by running the reference itself in the build container (the reference
holds no tests or golden vectors of its own).  The generic
L2-argmin at K x D beyond FSQ's implicit 1000 x 4 grid and the codebook EMA have
no reference implementation: for those shapes PARITY IS UNPINNED and the oracle
is the definition.
"""
