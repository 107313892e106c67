"""Enrolment gallery behind ``MVectorPredictor.register / recognition / remove_user``.

On disk it is the reference's layout, so galleries are interchangeable (mvector/predict.py:85-161, 281-318, 343-363):

    <root>/<user name>/<k>.wav          the enrolled audio
    <root>/audio_indexes.bin            pickle {"users_name": [...], "faces_feature": ndarray [N, D], "users_image_path": [...]}

In memory: one row per enrolled audio plus running per-user sums, from which the matrix of per-user mean embeddings that
recognition scores against is rebuilt lazily (the reference re-stacks numpy arrays on every change).

``DeviceGallery`` is the GPU-resident mirror the HIP predictor scores against: one fp32 row of per-user embedding SUMS per
user in a device matrix that grows geometrically; enrolling an audio adds one row in place (``index_add_``), so the
matrix is never re-uploaded.  Cosine similarity ignores the scale of a row, so scoring against the sums equals scoring
against the reference's per-user means (predict.py:142-151, 165-183).
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


class DeviceGallery:
    """Per-user embedding sums resident on the device (see the module docstring); mirrors a SpeakerGallery."""

    def __init__(self, device, dim, capacity=64):
        import torch
        self.device = device
        self.users = []                       # row index -> user name (first-enrolment order)
        self._row_of = {}
        self.sums = torch.zeros((capacity, dim), dtype=torch.float32, device=device)
        self.uploads = 0                      # host -> device row transfers (tests: recognition must not add any)

    @classmethod
    def from_gallery(cls, gallery, device, dim):
        dg = cls(device, dim, capacity=max(64, 2 * len(set(gallery.names))))
        import torch
        if len(gallery):
            rows = torch.from_numpy(np.stack(gallery.rows)).to(device)  # one upload of the whole index
            dg.uploads += 1
            for name in gallery.names:
                dg._row(name)
            idx = torch.tensor([dg._row_of[n] for n in gallery.names], dtype=torch.int64, device=device)
            dg.sums.index_add_(0, idx, rows)
        return dg

    def _row(self, name):
        import torch
        r = self._row_of.get(name)
        if r is None:
            r = len(self.users)
            if r == self.sums.shape[0]:   # grow geometrically, on the device
                bigger = torch.zeros((2 * r, self.sums.shape[1]), dtype=torch.float32, device=self.device)
                bigger[:r] = self.sums
                self.sums = bigger
            self.users.append(name)
            self._row_of[name] = r
        return r

    def add(self, name, embedding):
        """embedding: 1-D device tensor (stays on the device) or array (one row upload)"""
        import torch
        if not torch.is_tensor(embedding):
            embedding = torch.as_tensor(np.asarray(embedding, dtype=np.float32))
        if embedding.device != self.sums.device:
            embedding = embedding.to(self.device)
            self.uploads += 1
        r = self._row(name)  # first: a new user may replace self.sums by a bigger matrix
        self.sums[r] += embedding.reshape(-1).float()

    def remove(self, name):
        """drop a user: the rows behind it move up (device-side copy), order of the others is kept"""
        r = self._row_of.pop(name, None)
        if r is None:
            return False
        n = len(self.users)
        if r < n - 1:
            self.sums[r:n - 1] = self.sums[r + 1:n].clone()
        self.sums[n - 1].zero_()
        del self.users[r]
        self._row_of = {u: i for i, u in enumerate(self.users)}
        return True

    def matrix(self):
        return self.sums[:len(self.users)]
