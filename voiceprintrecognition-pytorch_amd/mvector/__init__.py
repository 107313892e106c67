"""mvector -- MI355X-native speaker-embedding extraction behind the API of mvector 1.1.1.

Same import surface as yeyupiaoling/VoiceprintRecognition-Pytorch for the embedding path
(``mvector.predict.MVectorPredictor``, ``mvector.data_utils.featurizer.AudioFeaturizer``,
``mvector.models.build_model`` and the model classes with the reference's ``state_dict`` layout); the
device work is done by hand-written HIP kernels in ``lib/libmvector_hip.so`` (see ``_hip.py``).
"""
__version__ = "1.1.1"
