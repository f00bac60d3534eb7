"""CPU oracle for the EMG->mel transduction training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under silent_speech_amd/ may import this package; the only
permitted callers are tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py, and
there only as the checker / the timed CPU baseline, never as the product path.

Each function is an independent restatement (closed-form where the reference uses pad/view tricks)
of the reference algorithm and cites the reference file:line it follows.  Parity is PINNED: the
oracle is checked in tests/test_oracle_golden.py against golden vectors captured from the imported
reference itself (tests/golden/*.npz, generator tests/golden/make_golden.py).  The one exception is
the Slaney mel filterbank (librosa.filters.mel, third-party, not vendored, unpinned version):
'parity unpinned' for the basis values; everything downstream of the basis is pinned.
"""
