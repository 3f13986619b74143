"""ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under ``aptai_amd/`` may import this package.

A CPU restatement, in plain fp32/fp64 torch + numpy ops, of the reference's algorithm for the
APTAI hot path: the wav2vec2 encoder (whose arithmetic lives in the un-vendored, unpinned
third-party ``transformers`` — 5.15.0 in the survey container — and ``torch`` 2.10.0), the APTAI
regression/phoneme heads, the Wav2Vec2_PR CTC head, and the Force_APTAI aligner.  Each function
cites the reference file:line it follows.

Pinning: the reference ships no tests, golden vectors or fixtures.  The oracle is
pinned against outputs of the reference itself, run in the build container by
``tests/golden/make_golden.py`` (which imports /root/reference) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks the restatement against them.
Two pieces stay *parity unpinned*: the torchaudio/flashlight CTC beam decoder
(models/w2v2_pr.py:143-159, package absent) and ``RNN.forward`` for batch>1 (NameError at
models/modules.py:207, pinned through the sub-modules instead).

Allowed importers: ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s cpu_baseline leg.
"""
