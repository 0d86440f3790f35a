"""CPU oracle for the CLC encode/decode + RD-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``clc_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the reported baseline.

What it is: a plain-PyTorch (fp32, NCHW, CPU) restatement of
  * the third-party CompressAI 1.2.x leaves the reference executes
    (``oracle/leaves.py``; the library is NOT vendored under /root/reference and is
    not installed here — algorithm restated from its public definition, see
    SURVEY.md Appendix A),
  * the reference's own model graph (``oracle/graph.py`` follows
    /root/reference/models/CLC_run.py:108-814 and models/tcm.py:310-626),
  * the trainer's rate-distortion loss (``oracle/loss.py`` follows
    /root/reference/train_CLC.py:36-59),
  * the rANS coder and CDF quantiser in plain C (``oracle/rans_oracle.c``) and in
    pure Python (``oracle/rans_py.py``).

Pinning status:
  * graph wiring, window attention, CLM.py, Patch_Matching numeric kernels:
    PINNED against genuine reference code executed in the build container
    (tools/make_golden.py; fixtures under tests/golden/).
  * CompressAI leaf arithmetic and the rANS bitstream format: **parity unpinned**
    — the reference holds no tests/golden vectors and the dependency is absent,
    so these are pinned only against the published algorithm and against two
    independent implementations of it (C and pure Python) agreeing.
"""
