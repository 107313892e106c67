"""ctypes binding of libmvector_hip.so (the C ABI declared in include/mvector_hip.h).

This is the only place the Python package touches native code.  There is NO fallback: if the library is
missing or a call fails, a RuntimeError carrying ``mv_last_error()`` is raised.  Tensors are passed as raw
``data_ptr()`` values together with the current HIP stream; PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libmvector_hip.so')

MV_ACT_NONE, MV_ACT_RELU, MV_ACT_TANH, MV_ACT_SIGMOID = 0, 1, 2, 3
MV_PAD_ZERO, MV_PAD_REFLECT = 0, 1
MV_DT_F32, MV_DT_F16 = 0, 1

c_i32, c_i64, c_f32, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class MvFbankCfg(ctypes.Structure):
    _fields_ = [('sample_frequency', c_f32), ('frame_length_ms', c_f32), ('frame_shift_ms', c_f32),
                ('num_mel_bins', c_i32), ('low_freq', c_f32), ('high_freq', c_f32),
                ('preemphasis_coefficient', c_f32), ('remove_dc_offset', c_i32), ('use_power', c_i32),
                ('use_log_fbank', c_i32), ('subtract_time_mean', c_i32), ('window_type', c_i32), ('blackman_coeff', c_f32),
                ('snip_edges', c_i32), ('subtract_mean', c_i32), ('min_duration', c_f32), ('vtln_warp', c_f32), ('vtln_low', c_f32),
                ('vtln_high', c_f32), ('kernel', c_i32), ('min_samples', c_i64), ('use_energy', c_i32), ('raw_energy', c_i32),
                ('energy_floor', c_f32), ('htk_compat', c_i32)]


class MvMelSpecCfg(ctypes.Structure):
    _fields_ = [('sample_rate', c_i32), ('n_fft', c_i32), ('win_length', c_i32), ('hop_length', c_i32),
                ('f_min', c_f32), ('f_max', c_f32), ('n_mels', c_i32), ('power', c_f32), ('center', c_i32),
                ('subtract_time_mean', c_i32), ('mel_scale', c_i32), ('norm', c_i32), ('normalized', c_i32), ('window', c_vp),
                ('pad', c_i32), ('pad_mode', c_i32)]


class MvTensorRef(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('data', c_vp), ('numel', c_i64)]


class MvEcapaCfg(ctypes.Structure):
    _fields_ = [('input_size', c_i32), ('embd_dim', c_i32), ('channels', c_i32 * 5), ('kernel_sizes', c_i32 * 5),
                ('dilations', c_i32 * 5), ('attention_channels', c_i32), ('res2net_scale', c_i32),
                ('se_channels', c_i32), ('global_context', c_i32)]


class MvCamppCfg(ctypes.Structure):
    _fields_ = [('input_size', c_i32), ('embd_dim', c_i32), ('growth_rate', c_i32), ('bn_size', c_i32),
                ('init_channels', c_i32), ('head_precision', c_i32), ('xvector_probe', c_i32)]


class MvEres2Cfg(ctypes.Structure):
    _fields_ = [('version', c_i32), ('input_size', c_i32), ('embd_dim', c_i32), ('num_blocks', c_i32 * 4),
                ('m_channels', c_i32), ('mul_channel', c_i32), ('expansion', c_i32), ('base_width', c_i32),
                ('scale', c_i32), ('two_emb_layer', c_i32)]


class MvConv2dsDesc(ctypes.Structure):
    _fields_ = [('x', c_vp), ('x2', c_vp), ('cin1', c_i32), ('ldx', c_i64), ('ldx2', c_i64), ('w', c_vp), ('bias', c_vp),
                ('oscale', c_f32), ('res', c_vp), ('res2', c_vp), ('ldres', c_i64), ('ldres2', c_i64), ('add', c_vp),
                ('ldadd', c_i64), ('y', c_vp), ('ldy', c_i64), ('y2', c_vp), ('ldy2', c_i64), ('B', c_i32), ('H', c_i32),
                ('W', c_i32), ('cin16', c_i32), ('cout16', c_i32), ('ks', c_i32), ('stride', c_i32), ('epi', c_i32),
                ('lo', c_f32), ('hi', c_f32), ('cin_alg', c_i32), ('cout_alg', c_i32), ('stride_w', c_i32), ('peak', c_vp), ('nbw_hint', c_i32),
                ('ct_hint', c_i32), ('rows_hint', c_i32), ('ring_hint', c_i32), ('wgs_hint', c_i32), ('spw_hint', c_i32), ('nprod_hint', c_i32)]


class MvTdnnCfg(ctypes.Structure):
    _fields_ = [('input_size', c_i32), ('channels', c_i32), ('embd_dim', c_i32)]


class MvConv1dDesc(ctypes.Structure):
    _fields_ = [('x', c_vp), ('x2', c_vp), ('x_dtype', c_i32), ('ldx', c_i64), ('ldx2', c_i64), ('in_scale', c_vp),
                ('in_shift', c_vp), ('w_packed', c_vp), ('bias', c_vp), ('row_bias', c_vp), ('pre_act', c_i32),
                ('scale', c_vp), ('shift', c_vp), ('post_act', c_i32), ('gate', c_vp), ('gate_seg_len', c_i32),
                ('y', c_vp), ('y_dtype', c_i32), ('ldy', c_i64), ('add_src', c_vp), ('sum_dst', c_vp), ('ld_add', c_i64),
                ('ld_sum', c_i64), ('B', c_i32), ('T_in', c_i32), ('T_out', c_i32),
                ('cin', c_i32), ('cout', c_i32), ('k', c_i32), ('dilation', c_i32), ('stride', c_i32), ('pad', c_i32),
                ('pad_mode', c_i32), ('tile', c_i32), ('stat_sum', c_vp), ('stat_sq', c_vp), ('in_stat_sum', c_vp), ('in_stat_sq', c_vp),
                ('persist_blocks_hint', c_i32), ('clock_probe', c_vp)]


_SIGNATURES = {
    'mv_last_error': (ctypes.c_char_p, []),
    'mv_abi_version': (c_i32, []),
    'mv_fbank_default_cfg': (None, [ctypes.POINTER(MvFbankCfg)]),
    'mv_fbank_create': (c_i32, [ctypes.POINTER(MvFbankCfg), ctypes.POINTER(c_vp)]),
    'mv_fbank_destroy': (c_i32, [c_vp]),
    'mv_fbank_info': (c_i32, [c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    'mv_fbank_num_frames': (c_i32, [c_vp, c_i64, ctypes.POINTER(c_i64)]),
    'mv_fbank_forward': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'mv_fbank_forward_varlen': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'mv_fbank_forward_varlen_ws': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    'mv_fbank_workspace_bytes': (c_i32, [c_vp, c_i32, c_i64, ctypes.POINTER(ctypes.c_size_t)]),
    'mv_fbank_forward_ws': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    'mv_melspec_default_cfg': (None, [ctypes.POINTER(MvMelSpecCfg)]),
    'mv_melspec_create': (c_i32, [ctypes.POINTER(MvMelSpecCfg), ctypes.POINTER(c_vp)]),
    'mv_melspec_info': (c_i32, [c_vp, ctypes.POINTER(c_i32)]),
    'mv_melspec_destroy': (c_i32, [c_vp]),
    'mv_melspec_num_frames': (c_i32, [c_vp, c_i64, ctypes.POINTER(c_i64)]),
    'mv_melspec_workspace_bytes': (c_sz, [c_vp, c_i32, c_i64]),
    'mv_melspec_forward': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    'mv_ecapa_create': (c_i32, [ctypes.POINTER(MvEcapaCfg), ctypes.POINTER(MvTensorRef), c_i32, ctypes.POINTER(c_vp)]),
    'mv_campp_create': (c_i32, [ctypes.POINTER(MvCamppCfg), ctypes.POINTER(MvTensorRef), c_i32, ctypes.POINTER(c_vp)]),
    'mv_tdnn_create': (c_i32, [ctypes.POINTER(MvTdnnCfg), ctypes.POINTER(MvTensorRef), c_i32, ctypes.POINTER(c_vp)]),
    'mv_eres2net_create': (c_i32, [ctypes.POINTER(MvEres2Cfg), ctypes.POINTER(MvTensorRef), c_i32, ctypes.POINTER(c_vp)]),
    'mv_conv2d_first': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'mv_tstp_f32': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'mv_conv2ds_packed_elems': (c_i64, [c_i32, c_i32, c_i32]),
    'mv_conv2ds_pack_weight': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, ctypes.POINTER(c_f32), c_vp]),
    'mv_conv2ds_forward': (c_i32, [ctypes.POINTER(MvConv2dsDesc), c_vp]),
    'mv_map_split_f32': (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    'mv_map_merge_f32': (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    'mv_conv2d_first_s16': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'mv_tstp_s16': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'mv_model_destroy': (c_i32, [c_vp]),
    'mv_model_embd_dim': (c_i32, [c_vp, ctypes.POINTER(c_i32)]),
    'mv_model_info': (c_i32, [c_vp, c_i32, ctypes.POINTER(c_f32)]),
    'mv_model_workspace_bytes': (c_i32, [c_vp, c_i32, c_i32, ctypes.POINTER(c_sz)]),
    'mv_model_forward': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
    'mv_cosine_f32': (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp]),
    'mv_l2_normalize_f32': (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    'mv_conv1d_packed_elems': (c_i64, [c_i32, c_i32, c_i32]),
    'mv_conv1d_pack_weight': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'mv_conv1d_forward': (c_i32, [ctypes.POINTER(MvConv1dDesc), c_vp]),
    'mv_conv1d_stats_elems': (c_i64, [c_i32, c_i32, c_i32]),
    'mv_conv1d_in_stats_elems': (c_i64, [c_i32, c_i32, c_i32]),
    'mv_conv1d_in_stats_finish': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_f32, c_vp]),
    'mv_conv1d_stats_finish': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_f32, c_vp]),
    'mv_res2net_chain_f16': (c_i32, [c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                             ctypes.POINTER(c_vp), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    'mv_linear_f32': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    'mv_linear_f32_workspace_floats': (c_sz, [c_i32, c_i32, c_i32]),
    'mv_linear_f32_ws': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    'mv_profile_enable': (c_i32, [c_i32]),
    'mv_profile_read': (c_i32, [c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_i32]),
    'mv_wave_prepare_i16': (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i64, c_i32, c_f32, c_f32, c_vp, c_i64, c_vp, c_vp]),
    'mv_asp_pool_f16': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    'mv_fcm_conv3x3_f16': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    'mv_fcm_block_f16': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    'mv_fcm_block_c1_f16': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    'mv_fcm_c1_pack': (c_i32, [c_vp, c_vp]),
    'mv_time_stats_f16': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_f32, c_vp]),
    'mv_bn_relu_rows_f16': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def bind(cdll):
    """Attach prototypes to a loaded library (also used by the test-suite for its emulator build)."""
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return cdll


def bind_partial(cdll):
    """Prototypes for whatever subset of the ABI a (probe) library exports; tools only."""
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(cdll, name, None)
        if fn is not None:
            fn.restype = restype
            fn.argtypes = argtypes
    return cdll


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """The bound library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the HIP library has not been built (run `python __graft_entry__.py` or '
                f'`python voiceprintrecognition-pytorch_amd/build_native.py`). There is no non-HIP device path.')
        cdll = bind(ctypes.CDLL(LIB_PATH))
        if cdll.mv_abi_version() != 5:
            raise RuntimeError('libmvector_hip.so ABI version mismatch')
        _lib = cdll
    return _lib


def check(rc, cdll=None):
    if rc != 0:
        cdll = cdll or lib()
        raise RuntimeError(f'libmvector_hip: {cdll.mv_last_error().decode(errors="replace")} (code {rc})')


def current_stream(tensor):
    if tensor.is_cuda:
        return torch.cuda.current_stream(tensor.device).cuda_stream
    return None


def _ptr(t):
    return None if t is None else t.data_ptr()


class Fbank:
    """Handle of the fused Fbank + CMN + mask kernel (mv_fbank_*)."""

    WINDOW_TYPES = {'povey': 0, 'hamming': 1, 'hanning': 2, 'rectangular': 3, 'blackman': 4}
    KERNELS = {'auto': 0, 'generic': 1, 'tile': 2}

    def __init__(self, method_args=None, subtract_time_mean=True, cdll=None, kernel='auto'):
        """``method_args``: the keyword arguments of torchaudio.compliance.kaldi.fbank (featurizer.py:128 forwards them).  ``kernel``:
        'auto' | 'generic' (fbank_kernel) | 'tile' (fbank_tile_kernel; refused when the geometry has no instantiation) -- tests / tools."""
        self._cdll = cdll or lib()
        cfg = MvFbankCfg()
        self._cdll.mv_fbank_default_cfg(ctypes.byref(cfg))
        args = dict(method_args or {})
        mapping = {'sample_frequency': 'sample_frequency', 'frame_length': 'frame_length_ms',
                   'frame_shift': 'frame_shift_ms', 'num_mel_bins': 'num_mel_bins', 'low_freq': 'low_freq',
                   'high_freq': 'high_freq', 'preemphasis_coefficient': 'preemphasis_coefficient',
                   'remove_dc_offset': 'remove_dc_offset', 'use_power': 'use_power', 'use_log_fbank': 'use_log_fbank',
                   'blackman_coeff': 'blackman_coeff', 'snip_edges': 'snip_edges', 'subtract_mean': 'subtract_mean',
                   'min_duration': 'min_duration', 'vtln_warp': 'vtln_warp', 'vtln_low': 'vtln_low', 'vtln_high': 'vtln_high',
                   'use_energy': 'use_energy', 'raw_energy': 'raw_energy', 'energy_floor': 'energy_floor', 'htk_compat': 'htk_compat'}
        # arguments whose other values are not implemented (dither draws random numbers; non-power-of-two FFT sizes).  use_energy (since round 6) adds
        # the log-energy column: the output has num_mel_bins + 1 columns, which AudioFeaturizer.feature_dim -- like the reference's,
        # featurizer.py:110-111 -- does not count
        fixed = {'dither': 0.0, 'round_to_power_of_two': True, 'channel': (-1, 0)}
        ignored = ()
        for k, v in args.items():
            if k in mapping:
                field = mapping[k]
                typ = dict(MvFbankCfg._fields_)[field]
                setattr(cfg, field, int(v) if typ is c_i32 else float(v))
            elif k == 'window_type':
                if v not in self.WINDOW_TYPES:
                    raise Exception('Invalid window type ' + str(v))   # torchaudio's own message
                cfg.window_type = self.WINDOW_TYPES[v]
            elif k in fixed:
                allowed = fixed[k] if isinstance(fixed[k], tuple) else (fixed[k],)
                if v not in allowed:
                    raise NotImplementedError(f'Fbank argument {k}={v!r} is not implemented by the HIP kernel')
            elif k not in ignored:
                raise TypeError(f"fbank() got an unexpected keyword argument '{k}'")
        # torchaudio compares `len(waveform) < min_duration * sample_frequency` in double: the sample threshold goes down as an integer (the struct's
        # float32 min_duration rounds 0.1 s to 1601 samples at 16 kHz)
        import math
        cfg.min_samples = int(math.ceil(float(args.get('min_duration', 0.0)) * float(args.get('sample_frequency', cfg.sample_frequency))))
        cfg.subtract_time_mean = 1 if subtract_time_mean else 0
        cfg.kernel = self.KERNELS[kernel]
        self.num_mel_bins = cfg.num_mel_bins
        self.num_columns = cfg.num_mel_bins + (1 if cfg.use_energy else 0)
        self._needs_ws = (not cfg.snip_edges) or bool(cfg.use_energy)   # forms that cannot run without the caller workspace
        self._h = c_vp()
        check(self._cdll.mv_fbank_create(ctypes.byref(cfg), ctypes.byref(self._h)), self._cdll)

    def num_frames(self, num_samples):
        t = c_i64()
        check(self._cdll.mv_fbank_num_frames(self._h, num_samples, ctypes.byref(t)), self._cdll)
        return t.value

    def info(self):
        """{'tile_kernel': bool, 'pass_steps': (s0, s1)} -- which kernel the handle launches (mv_fbank_info)"""
        tk, steps = c_i32(), (c_i32 * 2)()
        check(self._cdll.mv_fbank_info(self._h, ctypes.byref(tk), steps), self._cdll)
        return {'tile_kernel': bool(tk.value), 'pass_steps': (steps[0], steps[1])}

    def __call__(self, wav, lens_ratio=None, num_samples=None, workspace=True):
        """wav [B, L] fp32 -> [B, T(L), F].  ``lens_ratio``: the reference's batched semantics (mean over all T frames,
        then mask).  ``num_samples`` (int64 [B]): every row featurised on its own length, zero rows beyond it.
        ``workspace=False`` keeps one workgroup per utterance (mv_fbank_forward without scratch; same bits, see the header)."""
        assert wav.dim() == 2 and wav.dtype == torch.float32
        assert lens_ratio is None or num_samples is None
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, L = wav.shape
        T = self.num_frames(L)
        out = torch.empty((B, T, self.num_columns), dtype=torch.float32, device=wav.device)
        if B == 0 or T == 0:
            return out
        # per-call scratch (the several-workgroups-per-utterance form of long utterances / batches smaller than the chip; the mirrored signal
        # of snip_edges=False): from torch's stream-aware caching allocator, so concurrent forwards on other streams never share it (the handle
        # owns no mutable state)
        need = ctypes.c_size_t()
        check(self._cdll.mv_fbank_workspace_bytes(self._h, B, L, ctypes.byref(need)), self._cdll)
        ws = torch.empty(need.value, dtype=torch.uint8, device=wav.device) if need.value and (workspace or self._needs_ws) else None
        if num_samples is not None:
            num_samples = num_samples.to(device=wav.device, dtype=torch.int64).contiguous()
            if ws is None:
                check(self._cdll.mv_fbank_forward_varlen(self._h, wav.data_ptr(), B, L, wav.stride(0), num_samples.data_ptr(), out.data_ptr(),
                                                         current_stream(wav)), self._cdll)
            else:
                check(self._cdll.mv_fbank_forward_varlen_ws(self._h, wav.data_ptr(), B, L, wav.stride(0), num_samples.data_ptr(), out.data_ptr(),
                                                            _ptr(ws), need.value, current_stream(wav)), self._cdll)
            return out
        if lens_ratio is not None:
            lens_ratio = lens_ratio.to(device=wav.device, dtype=torch.float32).contiguous()
        check(self._cdll.mv_fbank_forward_ws(self._h, wav.data_ptr(), B, L, wav.stride(0), _ptr(lens_ratio), out.data_ptr(), _ptr(ws),
                                             need.value if ws is not None else 0, current_stream(wav)), self._cdll)
        return out

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._cdll.mv_fbank_destroy(self._h)
        except Exception:
            pass


class MelSpec:
    """Handle of the MelSpectrogram + CMN + mask path (mv_melspec_*)."""

    PAD_MODES = {'reflect': 0, 'constant': 1, 'replicate': 2, 'circular': 3}

    def __init__(self, method_args=None, subtract_time_mean=True, cdll=None):
        self._cdll = cdll or lib()
        cfg = MvMelSpecCfg()
        self._cdll.mv_melspec_default_cfg(ctypes.byref(cfg))
        a = dict(method_args or {})
        allowed = {'sample_rate', 'n_fft', 'win_length', 'hop_length', 'f_min', 'f_max', 'pad', 'n_mels', 'power',
                   'normalized', 'center', 'pad_mode', 'onesided', 'norm', 'mel_scale', 'window_fn', 'wkwargs'}
        for k in a:
            if k not in allowed:
                raise TypeError(f"MelSpectrogram got an unexpected keyword argument '{k}'")
        # not implemented: two-sided spectra (torchaudio's own MelScale refuses their bin count), power=None (a complex spectrogram has no mel scale)
        if a.get('onesided') not in (None, True) or a.get('power', 2.0) is None:
            raise NotImplementedError('MelSpectrogram option not implemented by the HIP kernel')
        if a.get('pad_mode', 'reflect') not in self.PAD_MODES:
            raise NotImplementedError(f"Unrecognised padding mode {a.get('pad_mode')}")   # (torch.nn.functional.pad's own error class)
        if a.get('norm') not in (None, 'slaney'):
            raise ValueError('norm must be one of None or "slaney"')          # torchaudio.functional.melscale_fbanks' own messages
        if a.get('mel_scale', 'htk') not in ('htk', 'slaney'):
            raise ValueError('mel_scale should be one of "htk" or "slaney".')
        if a.get('normalized', False) not in (False, True, 'window', 'frame_length'):
            raise ValueError(f"Invalid normalized parameter: {a.get('normalized')}")
        cfg.sample_rate = int(a.get('sample_rate', 16000))
        cfg.n_fft = int(a.get('n_fft', 400))
        win = a.get('win_length')
        cfg.win_length = int(win if win is not None else cfg.n_fft)
        hop = a.get('hop_length')
        cfg.hop_length = int(hop if hop is not None else cfg.win_length // 2)
        cfg.f_min = float(a.get('f_min', 0.0))
        f_max = a.get('f_max')
        cfg.f_max = float(f_max if f_max is not None else cfg.sample_rate // 2)
        cfg.n_mels = int(a.get('n_mels', 128))
        cfg.power = float(a.get('power', 2.0))
        cfg.center = 1 if a.get('center', True) else 0
        cfg.subtract_time_mean = 1 if subtract_time_mean else 0
        cfg.mel_scale = 1 if a.get('mel_scale', 'htk') == 'slaney' else 0
        cfg.norm = 1 if a.get('norm') == 'slaney' else 0
        cfg.normalized = {False: 0, True: 1, 'window': 1, 'frame_length': 2}[a.get('normalized', False)]
        cfg.pad = int(a.get('pad', 0))
        cfg.pad_mode = self.PAD_MODES[a.get('pad_mode', 'reflect')]
        win_host = None
        if a.get('window_fn') is not None:   # torchaudio evaluates window_fn(win_length, **wkwargs) once, at construction: so does this
            win_host = a['window_fn'](cfg.win_length, **(a.get('wkwargs') or {})).detach().to(device='cpu', dtype=torch.float32).contiguous()
            if win_host.shape != (cfg.win_length,):
                raise ValueError(f'window_fn returned {tuple(win_host.shape)}, expected ({cfg.win_length},)')
            cfg.window = win_host.data_ptr()
        self.n_mels = cfg.n_mels
        self._h = c_vp()
        check(self._cdll.mv_melspec_create(ctypes.byref(cfg), ctypes.byref(self._h)), self._cdll)

    def info(self):
        """{'tile_kernel': bool, 'kernel': name}: True when an FFT kernel runs -- melspec_tile_kernel (n_fft = 400) or melspec_pow2_kernel
        (n_fft 128 ... 1024, power of two) --, False for the dense-DFT kernels"""
        tk = c_i32()
        check(self._cdll.mv_melspec_info(self._h, ctypes.byref(tk)), self._cdll)
        return {'tile_kernel': bool(tk.value), 'kernel': {0: 'stft_power_kernel (dense DFT)', 1: 'melspec_tile_kernel', 2: 'melspec_pow2_kernel'}[tk.value]}

    def num_frames(self, num_samples):
        t = c_i64()
        check(self._cdll.mv_melspec_num_frames(self._h, num_samples, ctypes.byref(t)), self._cdll)
        return t.value

    def __call__(self, wav, lens_ratio=None):
        assert wav.dim() == 2 and wav.dtype == torch.float32
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, L = wav.shape
        T = self.num_frames(L)
        out = torch.empty((B, T, self.n_mels), dtype=torch.float32, device=wav.device)
        if B == 0 or T == 0:
            return out
        if lens_ratio is not None:
            lens_ratio = lens_ratio.to(device=wav.device, dtype=torch.float32).contiguous()
        nbytes = self._cdll.mv_melspec_workspace_bytes(self._h, B, L)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=wav.device)
        check(self._cdll.mv_melspec_forward(self._h, wav.data_ptr(), B, L, wav.stride(0), _ptr(lens_ratio), out.data_ptr(),
                                            ws.data_ptr(), nbytes, current_stream(wav)), self._cdll)
        return out

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._cdll.mv_melspec_destroy(self._h)
        except Exception:
            pass


class Model:
    """Handle of a native backbone (mv_*_create / mv_model_forward) built from a reference-layout state_dict."""

    def __init__(self, kind, cfg, state_dict, cdll=None):
        self._cdll = cdll or lib()
        names, tensors = [], []
        for k, v in state_dict.items():
            if not torch.is_floating_point(v):
                continue  # num_batches_tracked
            t = v.detach().to(torch.float32).contiguous()
            names.append(k.encode())
            tensors.append(t)
        refs = (MvTensorRef * len(tensors))()
        for i, (n, t) in enumerate(zip(names, tensors)):
            refs[i].name = n
            refs[i].data = t.data_ptr()
            refs[i].numel = t.numel()
        self._h = c_vp()
        create = {'ecapa': self._cdll.mv_ecapa_create, 'campp': self._cdll.mv_campp_create,
                  'tdnn': self._cdll.mv_tdnn_create, 'eres2net': self._cdll.mv_eres2net_create}[kind]
        if tensors and tensors[0].is_cuda:
            torch.cuda.current_stream(tensors[0].device).synchronize()  # weights fully written before create() reads them
        check(create(ctypes.byref(cfg), refs, len(tensors), ctypes.byref(self._h)), self._cdll)
        e = c_i32()
        check(self._cdll.mv_model_embd_dim(self._h, ctypes.byref(e)), self._cdll)
        self.embd_dim = e.value
        self._ws = {}   # one workspace per (device, stream): forwards of ONE handle on different streams run concurrently (the C ABI's contract)

    def workspace_bytes(self, B, T):
        n = c_sz()
        check(self._cdll.mv_model_workspace_bytes(self._h, B, T, ctypes.byref(n)), self._cdll)
        return n.value

    def info(self, key):
        """mv_model_info: 1 = CAM++ head on fp32 maps (1.0 / 0.0), 2 = its creation-time calibration 1 - cos, 3 + p = probe p's figure"""
        v = c_f32()
        check(self._cdll.mv_model_info(self._h, key, ctypes.byref(v)), self._cdll)
        return v.value

    XVEC_WARN = 2.5e-5   # MV_CAMPP_XVEC_WARN (include/mvector_hip.h)

    def campp_head(self, range=False):
        """CAM++ handles: {'head': 'f16' | 'f32', 'calibration': largest probe figure (-1 when pinned), 'probes': the three figures} -- the
        FCM head the handle chose at create (include/mvector_hip.h, MvCamppCfg.head_precision).  Host-side getters only: no device call, usable
        anywhere (also under stream capture).  range=True adds the exact head's peak / saturation on the caller's inputs since create, which WAITS
        FOR THE DEVICE (hipDeviceSynchronize + a blocking copy, csrc/campplus.hip): a diagnostic, never on a hot path."""
        d = {'head': 'f32' if self.info(1) == 1.0 else 'f16', 'calibration': self.info(2), 'probes': tuple(self.info(3 + p) for p in [0, 1, 2])}
        # what the head probes cannot see (since ABI 5): 1 - cos between the shipped (fp16-operand) and an exact-fp32 evaluation of the x-vector part
        # on the three probes; -1 = not measured.  Above XVEC_WARN the 1e-4 contract is at risk on this checkpoint whatever head runs.
        d.update(xvector_sensitivity=self.info(10), xvector_probes=tuple(self.info(11 + p) for p in [0, 1, 2]))
        d['xvector_warning'] = d['xvector_sensitivity'] > self.XVEC_WARN
        if d['head'] == 'f32':   # the exact head's range (include/mvector_hip.h, MV_INFO_CAMPP_HEAD_*): gain 2^k and the probes' peak, fixed at create
            d.update(gain_log2=int(self.info(6)), probe_peak=self.info(7))
            if range:
                d.update(peak=self.info(8), saturated=self.info(9) == 1.0)
        return d

    def forward(self, feats):
        assert feats.dim() == 3 and feats.dtype == torch.float32
        feats = feats.contiguous()
        B, T, _ = feats.shape
        emb = torch.empty((B, self.embd_dim), dtype=torch.float32, device=feats.device)
        if B == 0:
            return emb
        nbytes = self.workspace_bytes(B, T)
        stream = current_stream(feats)
        key = (str(feats.device), stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws.pop(key, None)   # (drop the smaller buffer before the larger one is allocated)
            ws = None
            if len(self._ws) >= 4:    # a caller that creates streams by the dozen must not pin a workspace for each
                self._ws.clear()
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=feats.device)
        check(self._cdll.mv_model_forward(self._h, feats.data_ptr(), B, T, emb.data_ptr(), ws.data_ptr(), ws.numel(), stream), self._cdll)
        return emb

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._cdll.mv_model_destroy(self._h)
        except Exception:
            pass


def wave_prepare(pcm, num_samples=None, target_db=None, max_gain_db=300.0, cdll=None):
    """int16 [B, L] on the device -> (float32 [B, L], too_quiet int32 [B]); ``target_db`` None = no dB normalisation."""
    cdll = cdll or lib()
    assert pcm.dim() == 2 and pcm.dtype == torch.int16
    if pcm.stride(1) != 1:
        pcm = pcm.contiguous()
    B, L = pcm.shape
    wav = torch.empty((B, L), dtype=torch.float32, device=pcm.device)
    flags = torch.zeros((B,), dtype=torch.int32, device=pcm.device)
    if num_samples is not None:
        num_samples = num_samples.to(device=pcm.device, dtype=torch.int64).contiguous()
    check(cdll.mv_wave_prepare_i16(pcm.data_ptr(), pcm.stride(0) if B else L, _ptr(num_samples), B, L, 0 if target_db is None else 1,
                                   float(target_db or 0.0), float(max_gain_db), wav.data_ptr(), L, flags.data_ptr(), current_stream(pcm)),
          cdll)
    return wav, flags


def cosine(a, b, cdll=None):
    """[N, D] x [M, D] -> [N, M] cosine similarity on the device of ``a``."""
    cdll = cdll or lib()
    a = a.to(torch.float32).contiguous()
    b = b.to(device=a.device, dtype=torch.float32).contiguous()
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1]
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if out.numel():
        check(cdll.mv_cosine_f32(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], a.shape[1], out.data_ptr(),
                                 current_stream(a)), cdll)
    return out
