"""CPU oracle for the TS-VAD / TS-SEP hot path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch CPU restatement
(torch fp32 / numpy) of the arithmetic on the reference's hot path
(SURVEY.md section 8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it -- as the checker, never as
the thing that is measured or shipped.  The product (``tssep_amd``) never
imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * rnnp / net / enhancer / loss / init_ckpt: pinned against outputs of the
    reference classes themselves (imported in the build container through
    in-memory stubs of its absent third-party packages; generator
    ``tests/golden/make_golden.py``, fixtures ``tests/golden/*.npz``) and
    against the reference's doctest known-answer values.
  * stft / istft / Log1pMaxNormAbsSTFT: the reference delegates to
    padertorch==0.0.1 / paderbox==0.0.8 (absent).  Restated from the
    published algorithm and pinned by the reference's own doctest numbers
    (feature_extractor.py:194-202, model.py:480,563-575).
  * TorchMFCC: delegates to torchaudio==2.0.2 (absent); no test or doctest in
    the reference asserts an MFCC value -> **parity unpinned** for the 40 MFCC
    feature columns (restated from torchaudio's published algorithm).
"""
