"""Mixin that routes eval-mode CUDA forwards of a backbone to its native (HIP) handle.

The module keeps the reference's parameter tree (so ``model.pth`` loads unchanged); the native handle is a
derived cache built from ``state_dict()`` on first use and dropped whenever the parameters can have changed
(``load_state_dict``, ``.to()/.cuda()/.float()``, ``train()``, or an in-place edit that bumps a tensor's version counter:
``with torch.no_grad(): p.copy_(..)``, EMA swaps; edits through ``p.data`` bypass the counter and assigning a NEW Parameter
object to a submodule attribute leaves the cached tensor list pointing at the old one -- call ``invalidate_native()`` after
those).  The per-forward check is a sum of version counters over a tensor list cached with the handle (CAM++: 815 tensors,
~75 us), not a ``state_dict()`` walk (1.9 ms, more than the GPU time of a batch-1 forward).  On CUDA tensors in eval mode there is no PyTorch fallback: if libmvector_hip.so is
missing the forward raises.  The handle is never part of the module's pickled / deep-copied state, and a forward that
autograd has to differentiate with respect to its input (grad mode on, ``x.requires_grad``) runs the torch graph, as the
reference's modules do; fine-tuning needs ``train()`` mode, which is the torch graph as well.
"""
import operator

import torch

_version_of = operator.attrgetter('_version')


class NativeBackbone:
    _native_kind = None  # 'ecapa' | 'campp' | 'tdnn'

    def _native_cfg(self):
        raise NotImplementedError

    def _native_supported(self):
        return True, ''

    def _native_created(self, handle, build):
        """hook: a freshly built handle (``build()`` builds another one from the same parameters); returns the handle to keep"""
        return handle

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

    def _params_version(self, tensors=None):
        """sum of the autograd version counters of the parameters and buffers (``tensors``: the list cached with a handle)"""
        if tensors is None:
            tensors = self._forward_tensors(self.state_dict(keep_vars=True))
        return sum(map(_version_of, tensors))

    @staticmethod
    def _forward_tensors(live):
        """the tensors an eval forward reads (BatchNorm's num_batches_tracked counters only matter to training)"""
        return [v for k, v in live.items() if not k.endswith('num_batches_tracked')]

    def _use_native(self, x):
        if not x.is_cuda or self.training:
            return False
        if torch.is_grad_enabled() and x.requires_grad:
            return False  # gradients with respect to the input were asked for: only the torch graph carries a grad_fn
        return True  # (parameters require grad by default and the reference's predictor never enters no_grad, predict.py:228,262:
        #               that alone must not take inference off the HIP path)

    def _native_handle_on(self, key):
        """the native handle of CUDA device ``key`` (the caller has made it current), built from the live parameters when there is none or they
        have changed"""
        handles = self.__dict__.setdefault('_native_handles', {})
        h, built_at, tensors = handles.get(key, (None, None, None))
        if h is None or built_at != self._params_version(tensors):
            ok, why = self._native_supported()
            if not ok:
                raise NotImplementedError(f'{type(self).__name__}: {why} is not implemented on the MI355X path')
            from mvector import _hip
            live = self.state_dict(keep_vars=True)
            tensors = self._forward_tensors(live)
            for v in tensors:
                if v.device.type != 'cuda' or (v.device.index if v.device.index is not None else key) != key:
                    raise RuntimeError(f'{type(self).__name__} parameters are on {v.device} but the input is on cuda:{key}')
            build = lambda: _hip.Model(self._native_kind, self._native_cfg(), {k: v.detach() for k, v in live.items()})
            h = self._native_created(build(), build)
            handles[key] = (h, self._params_version(tensors), tensors)
        return h

    def _native_handle(self, device):
        key = device.index if device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(key):
            return self._native_handle_on(key)

    def _native_forward(self, x):
        key = x.device.index if x.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(key):
            return self._native_handle_on(key).forward(x if x.dtype == torch.float32 else x.float())
