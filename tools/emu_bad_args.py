"""Bad arguments at the C ABI (no GPU): every field of MvConv1dDesc / MvConv2dsDesc of a VALID layer call -- and every pointer / integer
argument of the positional entry points (linear, time statistics, ASP pooling, Res2Net chain, BN + ReLU rows, TSTP, wave preparation, cosine) -- is replaced, one at a
time, by values a caller can get wrong -- a null pointer, 0, -1, a huge size, an enum out of range, a leading dimension smaller than the row --
and the entry point is called on the emulator build.  The contract (include/mvector_hip.h): a call the library cannot run returns an error code
and a message; it never crashes and never touches memory outside the caller's buffers.

    python tools/emu_bad_args.py                 # plain emulator build
    python tools/emu_bad_args.py --mode asan     # AddressSanitizer build: an accepted bad value that reads or writes out of bounds is reported

Output: one line per (entry point, field, value): "rejected" (error code + message), "accepted" (returned MV_OK: fine for optional operands and
hints, listed so that a reader can judge) -- and, if the process dies, the last line printed names the call that killed it.
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests'), ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd'), os.path.join(ROOT, 'tools')]

POINTER_T = ctypes.c_void_p


def candidates(name, typ, value):
    if typ is POINTER_T:
        # only the REQUIRED tensors: dropping an optional operand (second input, residual, bias ...) is a different, valid call whose other arguments
        # -- the channel split of a concatenation, say -- the caller would have set differently
        return [None] if value and name in ('x', 'w', 'w_packed', 'y') else []
    if typ in (ctypes.c_int32, ctypes.c_int64):
        # (sizes LARGER than the buffers are the caller's statement about its own memory: the library has no way to see behind a pointer, so only
        # values that are wrong whatever the buffers are get tried)
        vals = [0, -1]
        if name.startswith('ld') or name in ('ld_add', 'ld_sum'):
            vals += [1, 7]                       # a row shorter than its channels / not a multiple of the vector width
        if name in ('cin16', 'cout16', 'cin1'):
            vals += [value + 1, 8]               # not padded to 16
        if name in ('ks', 'k'):
            vals += [2, 4, 9]
        if name in ('x_dtype', 'y_dtype', 'pad_mode', 'pre_act', 'post_act', 'epi', 'x2_mode', 'stride', 'stride_w', 'dilation', 'tile'):
            vals += [99, 2 ** 31 - 1]
        if name in ('x_dtype', 'y_dtype'):
            vals = [-1, 99, 2 ** 31 - 1]          # (the other VALID code would be a wrong statement about the buffer's element size, not a bad argument)
        return [v for v in dict.fromkeys(vals) if v != value]
    if typ is ctypes.c_float:
        return [float('nan'), 0.0, -1.0] if name == 'oscale' else []
    return []


class TamperArgs:
    """the same for entry points with positional arguments: every pointer argument that was given becomes null, every integer 0 and -1 -- the
    LAST argument (the stream) excepted.  A null pointer for an OPTIONAL operand is a valid call; it shows up under ACCEPTED."""

    def __init__(self, cdll, fn_name, log):
        from mvector import _hip
        self._cdll, self._fn, self._log = cdll, fn_name, log
        self._argtypes = _hip._SIGNATURES[fn_name][1]

    def __getattr__(self, name):
        real = getattr(self._cdll, name)
        if name != self._fn:
            return real

        def call(*args):
            for i, (typ, val) in enumerate(zip(self._argtypes[:-1], args[:-1])):
                if typ is POINTER_T:
                    bads = [None] if val else []
                elif typ in (ctypes.c_int32, ctypes.c_int64):
                    bads = [v for v in (0, -1) if v != val]
                else:
                    bads = []
                for bad in bads:
                    print(f'CALL {self._fn} arg{i} = {bad!r}', flush=True)
                    rc = real(*(args[:i] + (bad,) + args[i + 1:]))
                    msg = self._cdll.mv_last_error().decode() if rc != 0 else ''
                    self._log.append((self._fn, f'arg{i}', bad, rc, msg))
                    print(f'  -> {"accepted" if rc == 0 else "rejected: " + msg[:120]}', flush=True)
            return real(*args)
        return call


class Tamper:
    """stands in for the bound library: the named entry point is first called with every tampered copy of its descriptor"""

    def __init__(self, cdll, fn_name, desc_type, log):
        self._cdll, self._fn, self._type, self._log = cdll, fn_name, desc_type, log

    def __getattr__(self, name):
        real = getattr(self._cdll, name)
        if name != self._fn:
            return real

        def call(desc_ref, stream):
            d = desc_ref._obj
            for field, typ in self._type._fields_:
                saved = getattr(d, field)
                for bad in candidates(field, typ, saved):
                    print(f'CALL {self._fn} {field} = {bad!r}', flush=True)     # (the line that names a crash)
                    setattr(d, field, bad)
                    rc = real(ctypes.byref(d), stream)
                    msg = self._cdll.mv_last_error().decode() if rc != 0 else ''
                    self._log.append((self._fn, field, bad, rc, msg))
                    print(f'  -> {"accepted" if rc == 0 else "rejected: " + msg[:120]}', flush=True)
                    setattr(d, field, saved)
            return real(desc_ref, stream)
        return call


def worker():
    import layer_checks as lc
    from mvector import _hip
    from emu_lib import emu_cdll
    cdll = emu_cdll()
    log = []
    # one valid layer per entry point, small enough that an "accepted" variant costs little
    lc.conv1d_case(Tamper(cdll, 'mv_conv1d_forward', _hip.MvConv1dDesc, log), 'cpu', B=2, T=21, cin=24, cout=40, k=3, dil=2, with_x2=True, row_bias=True)
    lc.conv1d_case(Tamper(cdll, 'mv_conv1d_forward', _hip.MvConv1dDesc, log), 'cpu', B=5, T=90, cin=128, cout=256, k=1, dil=1, tile=256, stats=2)
    lc.conv2ds_case(Tamper(cdll, 'mv_conv2ds_forward', _hip.MvConv2dsDesc, log), 'cpu', B=1, H=5, W=9, cin=16, cout=32, ks=3, with_res=True, with_sum=True)
    lc.conv2ds_case(Tamper(cdll, 'mv_conv2ds_forward', _hip.MvConv2dsDesc, log), 'cpu', B=1, H=4, W=9, cin=64, cout=16, ks=1, concat=True, epi=1)
    import torch
    lc.linear_case(TamperArgs(cdll, 'mv_linear_f32', log), 'cpu', B=3, K=40, O=12)
    lc.time_stats_case(TamperArgs(cdll, 'mv_time_stats_f16', log), 'cpu', B=2, T=9, C=72, ld=80)
    lc.asp_pool_case(TamperArgs(cdll, 'mv_asp_pool_f16', log), 'cpu', B=2, T=20, C=72, A=64)
    lc.res2_chain_case(TamperArgs(cdll, 'mv_res2net_chain_f16', log), 'cpu', B=1, T=20, width=64, dil=2)
    lc.bn_relu_rows_case(TamperArgs(cdll, 'mv_bn_relu_rows_f16', log), 'cpu', rows=9, C=72, ldx=80, ldy=88)
    lc.tstp_case(TamperArgs(cdll, 'mv_tstp_f32', log), 'cpu', B=2, H=3, W=10, C=16)
    lc.wave_prepare_case(TamperArgs(cdll, 'mv_wave_prepare_i16', log), 'cpu', B=2, L=300)
    a, b = torch.randn(5, 16), torch.randn(7, 16)
    lc.cosine_case(TamperArgs(cdll, 'mv_cosine_f32', log), 'cpu', a.numpy(), b.numpy(),
                   (torch.nn.functional.normalize(a, dim=1) @ torch.nn.functional.normalize(b, dim=1).t()).numpy())
    rej = sum(1 for e in log if e[3] != 0)
    print(f'SUMMARY {len(log)} tampered calls: {rej} rejected, {len(log) - rej} accepted, 0 crashed', flush=True)
    acc = sorted({(e[0], e[1], e[2]) for e in log if e[3] == 0})
    print('ACCEPTED ' + '; '.join(f'{f}.{k}={v!r}' for f, k, v in acc), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='plain', choices=['plain', 'asan', 'ubsan'])
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.worker:
        worker()
        return
    import emu_fuzz
    env = emu_fuzz.mode_env(args.mode)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tests', 'emu', 'build_emu.py')], env={**env, 'LD_PRELOAD': ''}, stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker'], env=env, cwd=ROOT, capture_output=True, text=True)
    lines = r.stdout.splitlines()
    if r.returncode != 0 or not any(l.startswith('SUMMARY') for l in lines):
        calls = [l for l in lines if l.startswith('CALL')]
        print(f'{args.mode}: the process died (rc {r.returncode}) in: {calls[-1] if calls else "?"}')
        print('\n'.join(l for l in (r.stdout + r.stderr).splitlines() if 'ERROR' in l or ' #0 ' in l or ' #1 ' in l or 'located' in l or 'SUMMARY' in l))
        sys.exit(1)
    for l in lines:
        if l.startswith(('SUMMARY', 'ACCEPTED')):
            print(f'{args.mode}: {l}')


if __name__ == '__main__':
    main()
