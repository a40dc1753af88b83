"""CPU oracle for the ViTGAN G+D hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the thing timed as
"the reference on host cores" - never as a fallback for the HIP path.

The oracle is a functional (state-dict in, tensors out) fp32 PyTorch
restatement of the reference's algorithm.  It is pinned against outputs of the
reference itself (``tests/golden/*.npz``, produced by ``tests/golden/make_golden.py``
which imports ``/root/reference`` in the build container).
"""
