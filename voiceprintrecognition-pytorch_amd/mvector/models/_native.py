"""Mixin that routes eval-mode CUDA forwards of a backbone to its native (HIP) handle.

The module keeps the reference's parameter tree (so ``model.pth`` loads unchanged); the native handle is a
derived cache built from ``state_dict()`` on first use and dropped whenever the parameters can have changed
(``load_state_dict``, ``.to()/.cuda()/.float()``, ``train()``, or an in-place edit that bumps a tensor's version counter:
``with torch.no_grad(): p.copy_(..)``, EMA swaps; edits through ``p.data`` bypass the counter -- call
``invalidate_native()`` after those).  On CUDA tensors in eval mode there is no PyTorch fallback: if libmvector_hip.so is
missing the forward raises.  The handle is never part of the module's pickled / deep-copied state, and a forward that
autograd has to differentiate with respect to its input (grad mode on, ``x.requires_grad``) runs the torch graph, as the
reference's modules do; fine-tuning needs ``train()`` mode, which is the torch graph as well.
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

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_native_handles', None)  # ctypes handles neither pickle nor survive a copy (each copy builds its own)
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__dict__.update({k: copy.deepcopy(v, memo) for k, v in self.__getstate__().items()})
        return new

    def _params_version(self):
        return sum(t._version for t in self.state_dict(keep_vars=True).values())

    def _use_native(self, x):
        if not x.is_cuda or self.training:
            return False
        if torch.is_grad_enabled() and x.requires_grad:
            return False  # gradients with respect to the input were asked for: only the torch graph carries a grad_fn
        return True  # (parameters require grad by default and the reference's predictor never enters no_grad, predict.py:228,262:
        #               that alone must not take inference off the HIP path)

    def _native_forward(self, x):
        ok, why = self._native_supported()
        if not ok:
            raise NotImplementedError(f'{type(self).__name__}: {why} is not implemented on the MI355X path')
        from mvector import _hip
        handles = self.__dict__.setdefault('_native_handles', {})
        key = x.device.index if x.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(key):
            version = self._params_version()
            h, built_at = handles.get(key, (None, None))
            if h is None or built_at != version:
                sd = {k: v for k, v in self.state_dict().items()}
                for v in sd.values():
                    if v.device != x.device:
                        raise RuntimeError(f'{type(self).__name__} parameters are on {v.device} but the input is on '
                                           f'{x.device}')
                h = _hip.Model(self._native_kind, self._native_cfg(), sd)
                handles[key] = (h, version)
            return h.forward(x if x.dtype == torch.float32 else x.float())
