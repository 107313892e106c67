"""Mixin that routes eval-mode CUDA forwards of a backbone to its native (HIP) handle.

The module keeps the reference's parameter tree (so ``model.pth`` loads unchanged); the native handle is a
derived cache built from ``state_dict()`` on first use and dropped whenever the parameters can have changed
(``load_state_dict``, ``.to()/.cuda()/.float()``, ``train()``).  On CUDA tensors in eval mode there is no
PyTorch fallback: if libmvector_hip.so is missing the forward raises.
"""
import torch


class NativeBackbone:
    _native_kind = None  # 'ecapa' | 'campp' | 'tdnn'

    def _native_cfg(self):
        raise NotImplementedError

    def _native_supported(self):
        return True, ''

    def invalidate_native(self):
        self.__dict__['_native_handles'] = {}

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_native()
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_native()
        return super()._load_from_state_dict(*args, **kwargs)

    def train(self, mode=True):
        self.invalidate_native()
        return super().train(mode)

    def _use_native(self, x):
        return x.is_cuda and not self.training

    def _native_forward(self, x):
        ok, why = self._native_supported()
        if not ok:
            raise NotImplementedError(f'{type(self).__name__}: {why} is not implemented on the MI355X path')
        from mvector import _hip
        handles = self.__dict__.setdefault('_native_handles', {})
        key = x.device.index if x.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(key):
            h = handles.get(key)
            if h is None:
                sd = {k: v for k, v in self.state_dict().items()}
                for v in sd.values():
                    if v.device != x.device:
                        raise RuntimeError(f'{type(self).__name__} parameters are on {v.device} but the input is on '
                                           f'{x.device}')
                h = _hip.Model(self._native_kind, self._native_cfg(), sd)
                handles[key] = h
            return h.forward(x if x.dtype == torch.float32 else x.float())
