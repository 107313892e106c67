"""CPU oracle for the mvector embedding-extraction hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / timed CPU baseline.  The
product path (``voiceprintrecognition-pytorch_amd/``) never imports this
package and fails loudly when its HIP library is missing.

What is restated here (torch-CPU fp32, written from the algorithm, not copied):

* ``frontend.py``  -- ``torchaudio.compliance.kaldi.fbank`` and
  ``torchaudio.transforms.MelSpectrogram`` (torchaudio 2.4.0 is the version the
  reference pins in prose, README.md:163; it is NOT vendored in /root/reference
  and NOT installed here) plus ``AudioFeaturizer.forward``
  (mvector/data_utils/featurizer.py:53-91, 114-132).
* ``models.py``    -- functional forwards of EcapaTdnn (mvector/models/ecapa_tdnn.py),
  CAMPPlus (mvector/models/campplus.py) and TDNN (mvector/models/tdnn.py) with
  their pooling layers (mvector/models/pooling.py) over a plain state_dict.
* ``scoring.py``   -- cosine scoring (mvector/predict.py:165-183, 275-279;
  mvector/trainer.py:454-461).

Pin status
----------
* models / scoring: PINNED.  ``make_golden.py`` imports the reference's own
  modules from /root/reference (loguru stubbed), loads weights produced by
  ``weights.py`` and stores their outputs in ``tests/golden/*.npz``;
  ``tests/test_oracle_models.py`` checks this restatement against them.
* ``AudioFeaturizer.forward`` / ``KaldiFbank.forward`` (the wrapper: per-utterance
  loop, transposes, time-mean subtraction, torch.round mask, feature_dim): PINNED.
  ``make_golden.py::save_featurizer_golden`` imports the reference's own
  ``mvector/data_utils/featurizer.py`` under a stub ``torchaudio`` whose two
  functions are the restatements below and stores its outputs on the Q1/Q2/Q3
  fixtures in ``tests/golden/featurizer_ref.npz``; the oracle wrapper agrees bit
  for bit, the product (CPU and HIP) within the front-end tolerance.
* the arithmetic INSIDE ``kaldi.fbank`` / ``MelSpectrogram``: the reference has no
  tests and that arithmetic lives in torchaudio, which cannot be run here, so
  there is no torchaudio-produced vector to pin to: **parity unpinned against
  torchaudio itself**.  The restatement is instead cross-checked against two independent
  implementations that ARE available: ``transformers.audio_utils.spectrogram``
  (the numpy code HuggingFace ships as the replacement for ``ta_kaldi.fbank``)
  and, for MelSpectrogram, ``torch.stft`` (the very op torchaudio calls) plus
  ``transformers.audio_utils.mel_filter_bank``.
  Round 5 widened the restatement to the keyword arguments the reference forwards
  (featurizer.py:42,128): kaldi.fbank's window types, ``snip_edges=False``,
  ``subtract_mean``, ``use_energy`` / ``htk_compat`` / ``raw_energy`` /
  ``energy_floor``, ``min_duration``, VTLN warping; MelSpectrogram's Slaney mel
  points / area normalisation, ``normalized`` modes, ``window_fn`` / ``wkwargs``,
  any real ``power`` -- written from the torchaudio 2.4.0 algorithms as recalled,
  **equally unpinned**; the Slaney filterbanks are cross-checked against
  ``transformers.audio_utils.mel_filter_bank(norm=..., mel_scale=...)``
  (tests/test_oracle.py), the rest has no second implementation here.
"""
