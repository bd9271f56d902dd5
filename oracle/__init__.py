"""CPU oracle for the fMRI->caption training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package (``masters-thesis_amd/``); only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the
checker / the timed CPU baseline.

PARITY UNPINNED: the reference (seang123/Masters-Thesis) ships no tests, golden
vectors or saved weights for this path, and its arithmetic
lives in TensorFlow 2.x / Keras 2.8, which is not installable here.  This oracle
is therefore a numpy restatement of the reference's op sequence
(AttemptFour/Model/{lc_NIC,NIC,layers,attention,fullyConnected}.py) with the
Keras semantics written out in SURVEY.md section 9.  It is pinned only by
  * the closed-form cross-entropy values of AttemptFour/temp.py:73-90,
  * finite-difference checks of every backward,
  * an independent torch-CPU autograd composition (tests/test_oracle_*.py).
"""
