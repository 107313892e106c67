"""Enrolment gallery behind ``MVectorPredictor.register / recognition / remove_user``.

On disk it is the reference's layout, so galleries are interchangeable (mvector/predict.py:85-161, 281-318, 343-363):

    <root>/<user name>/<k>.wav          the enrolled audio
    <root>/audio_indexes.bin            pickle {"users_name": [...], "faces_feature": ndarray [N, D], "users_image_path": [...]}

In memory: one row per enrolled audio plus running per-user sums, from which the matrix of per-user mean embeddings that
recognition scores against is rebuilt lazily (the reference re-stacks numpy arrays on every change).
"""
import os
import pickle
import shutil

import numpy as np

INDEX_NAME = 'audio_indexes.bin'


class SpeakerGallery:
    def __init__(self, root):
        self.root = root
        self.index_path = os.path.join(root, INDEX_NAME)
        self.names = []        # user of every enrolled audio (with repeats, enrolment order)
        self.paths = []        # its file
        self.rows = []         # its embedding (1-D float32 arrays)
        self._means = None     # cache: (user names, [U, D] matrix)
        os.makedirs(root, exist_ok=True)
        self._read_index()

    # ---- persistence ----------------------------------------------------------------------------------------------
    def _read_index(self):
        if not os.path.exists(self.index_path):
            return
        with open(self.index_path, 'rb') as f:
            blob = pickle.load(f)
        for name, row, path in zip(blob['users_name'], blob['faces_feature'], blob['users_image_path']):
            if os.path.exists(path):  # entries whose audio has been deleted are dropped
                self._append(name, path, row)

    def save(self):
        matrix = np.stack(self.rows) if self.rows else None
        with open(self.index_path, 'wb') as f:
            pickle.dump({'users_name': list(self.names), 'faces_feature': matrix, 'users_image_path': list(self.paths)}, f)

    # ---- content --------------------------------------------------------------------------------------------------
    def _append(self, name, path, row):
        self.names.append(name)
        self.paths.append(path)
        self.rows.append(np.asarray(row, dtype=np.float32).reshape(-1))
        self._means = None

    def unindexed_audio(self):
        """Audio files lying in the user folders that the index does not know yet (sorted for a stable order)."""
        known, found = set(self.paths), []
        for user in sorted(os.listdir(self.root)):
            folder = os.path.join(self.root, user)
            if os.path.isdir(folder):
                found += [os.path.join(folder, f).replace('\\', '/') for f in sorted(os.listdir(folder))]
        return [p for p in found if p not in known]

    def add(self, name, path, row):
        self._append(name, path.replace('\\', '/'), row)

    def next_audio_path(self, name):
        folder = os.path.join(self.root, name)
        os.makedirs(folder, exist_ok=True)
        return os.path.join(folder, f'{len(os.listdir(folder))}.wav')

    def remove(self, name):
        if name not in self.names:
            return False
        keep = [i for i, n in enumerate(self.names) if n != name]
        self.names, self.paths, self.rows = ([seq[i] for i in keep] for seq in (self.names, self.paths, self.rows))
        self._means = None
        self.save()
        shutil.rmtree(os.path.join(self.root, name), ignore_errors=True)
        return True

    def user_means(self):
        """(user names in first-enrolment order, [U, D] float32 matrix of their mean embeddings)."""
        if self._means is None:
            order, sums, counts = [], {}, {}
            for name, row in zip(self.names, self.rows):
                if name not in sums:
                    order.append(name)
                    sums[name], counts[name] = np.zeros_like(row, dtype=np.float64), 0
                sums[name] += row
                counts[name] += 1
            matrix = np.stack([(sums[n] / counts[n]).astype(np.float32) for n in order]) if order else None
            self._means = (order, matrix)
        return self._means

    def __len__(self):
        return len(self.names)
