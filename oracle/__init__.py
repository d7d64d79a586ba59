"""TEST INFRASTRUCTURE ONLY -- the parity oracle for the TomoSAR2Height hot path.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, call, link or execute it, and there only as the checker / the timed CPU
baseline -- never as the thing measured or shipped.  The product path
(``tomosar2height_amd``) never imports this package and fails loudly when its
HIP library is missing.

Contents
--------
``scatter_ref.py``   restatement of the two ``torch_scatter`` entry points the
                     reference calls (pytorch-scatter 2.1.2; source NOT under
                     /root/reference -> restated from its published semantics).
``torch_ref.py``     module-for-module torch restatement of the reference model
                     (CPU fp32); pinned against the imported reference by
                     ``tests/golden/make_golden.py`` fixtures.
``t2h_oracle.c``     plain-C restatement of the operator-level arithmetic
                     (coordinate2index, scatter_max/mean, pool_local,
                     grid_sample, bilinear upsample, linear, ResnetBlockFC),
                     forward and backward, in ORIGINAL point order.
``c_oracle.py``      ctypes loader for the compiled C oracle.

Pin status
----------
* torch-only arithmetic (Linear, grid_sample, interpolate, scatter_add_ based
  scatter_mean, the whole module graph): PINNED by fixtures captured from the
  reference's own Python modules imported in the build container
  (tests/golden/*.npz, generator script committed next to them).
* ``scatter_max`` (tie-break / argmax): the reference holds no test or golden
  vector for it and pytorch-scatter cannot be imported or built here ->
  "parity unpinned" for the argmax tie-break rule; restated as documented
  (first occurrence wins, untouched cells -> value 0 / arg = N).
"""
