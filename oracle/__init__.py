"""CPU oracle for the astrild hot path — TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / C restatement of the algorithms on the hot
path named in BASELINE.json (particle->grid mass assignment, 3D-FFT P(k),
bispectrum, Ray-Ramses kappa-map stack and per-map pipeline).  It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``astrild_amd/`` imports it, and the product path raises if the HIP library
is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):

* rays sub-path (kappa->alpha/phi, unit conversion, Gaussian smoothing,
  NFW stamps): PINNED by the reference's own known-answer tests
  (/root/reference/tests/unit/rays/skys/test_skyutils.py:43-125,
  /root/reference/tests/unit/rays/utils/test_filters.py:47-83); the values
  are restated as data in tests/golden/reference_known_answers.json.
* 3D path (paint, r2c, FFTPower binning, bispectrum) and the kappa-stack
  sum: PARITY UNPINNED.  The arithmetic lives in un-vendored third-party
  packages (nbodykit 0.3.14, pmesh 0.1.55, pfft-python 0.1.21 per
  /root/reference/poetry.lock:333-336,478-481,449-452) that are not
  installed and the reference has no test or fixture for it.  The
  restatement follows their published algorithms and is anchored on the
  reference call sites (power_spectrum_3d.py:140-226,
  stats_subfind.py:125-150) plus analytic known-answer tests
  (tests/test_oracle_mesh.py).
"""
