"""CPU oracle for the WDSR-B Conv3D hot path of mmbajo/PROBA-V.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product path:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU baseline.

PARITY UNPINNED.  The reference is TensorFlow 2 + tensorflow-addons (both unpinned:
``/root/reference/Dockerfile:1``, ``requirements.txt:2``).  Neither is installed in the
build container and the reference ships no tests, golden vectors or trained weights
(SURVEY.md F1, F2, F4), so this restatement cannot be checked against outputs of the
reference itself.  It is pinned instead by

* two independent formulations that must agree to 1e-9 in fp64
  (``wdsr_numpy`` = explicit tap loops / einsum, ``wdsr_torch`` = ``F.conv3d``),
* the variable inventory of the reference's own checkpoints
  (``modelInfo/ckpt_p16t9c85r12/NIR/ckpt-124.index``: 44 layers, 535 267 fp32),
* finite-difference checks of every gradient formula.

Modules
-------
``wdsr_numpy``  fp64 numpy restatement of models/modelsTF.py (WDSRConv3D) and models/loss.py
``wdsr_torch``  torch (CPU) restatement of the same, autograd for gradients; also the
                timed CPU baseline
``synth``       seeded synthetic weights / patches
"""
