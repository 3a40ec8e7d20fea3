"""CPU oracle for the GP-posterior + safe-set hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``safeopt_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.

Contents
--------
``gp_numpy``      NumPy/SciPy restatement of the GPy arithmetic SafeOpt calls
                  (SURVEY.md section 8a rows A0-A2).  GPy itself is a third-party
                  dependency of the reference (``requirements.txt:1``,
                  ``GPy>=0.8``, unpinned; not vendored under /root/reference and
                  not installable here), so this part restates GPy's published
                  algorithm.  It is cross-checked against scikit-learn's
                  GaussianProcessRegressor and closed-form answers
                  (tests/test_oracle_gp.py) -- GPy bit-level parity is UNPINNED.
``safeopt_numpy`` NumPy restatement of the reference's own set logic
                  (``safeopt/gp_opt.py:453-712, 874-1013``).  PINNED: checked
                  against golden vectors produced by running the reference's
                  ``gp_opt.py`` itself in the build container
                  (tests/golden/make_golden.py).
"""
