"""CPU oracle for the VQ-NeRF hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``vqnerf_release_amd/`` may import this package.  The only
legitimate importers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- always as the checker / the timed CPU
baseline, never as the product path.

* ``oracle.geo``    -- torch-CPU restatement of the NeuS ray marcher
                      (reference ``geo/NeuS-ours2/models/{renderer,fields,embedder}.py``).
                      PINNED: ``oracle/gen_golden_geo.py`` imports the real
                      reference in the build container and commits its outputs
                      under ``tests/golden/geo_*.npz``.
* ``oracle.decomp`` -- numpy restatement of the reflectance path
                      (reference ``decomp/nerfvq_nfr3/nerfactor/...``).
                      PARITY UNPINNED: TensorFlow / Sonnet are not installable
                      here and the reference ships no fixtures; pinned only by
                      analytic known-answer tests.
* ``oracle/vq_strict.c`` -- plain-C statement of the VQ distance/argmin in the
                      exact fmaf order the HIP kernel uses (bit-exact checker).
"""
