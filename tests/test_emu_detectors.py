"""The emulator's checking modes detect what they are for (tests/emu/hip_emu.h header; tools/emu_check.py runs the whole emulator suite
under them): a missing barrier that the forward thread order hides shows in the reverse / random orders, and the AddressSanitizer build
stops at an index one element behind the dynamic LDS block or a global buffer."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, 'emu')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
SRC = os.path.join(EMU, 'selftest', 'detectors.cpp')


def build(tmp_path, asan):
    exe = str(tmp_path / ('detectors_asan' if asan else 'detectors'))
    cmd = [CLANG, '-x', 'c++', '-std=c++17', '-O1', '-I', EMU, '-include', os.path.join(EMU, 'hip_emu.h'), '-Wno-unused-value', SRC, '-o', exe]
    if asan:
        cmd[1:1] = ['-fsanitize=address', '-fno-omit-frame-pointer', '-g1']
    subprocess.check_call(cmd)
    return exe


def run(exe, *args, sched=None, poison=False, dma=None, lds=None):
    env = dict(os.environ)
    env.pop('MV_EMU_DMA', None)
    env.pop('MV_EMU_LDS', None)
    if lds:
        env['MV_EMU_LDS'] = lds
    if dma:
        env['MV_EMU_DMA'] = dma
    env.pop('MV_EMU_SCHED', None)
    env.pop('MV_EMU_POISON', None)
    if poison:
        env['MV_EMU_POISON'] = '1'
    if sched:
        env['MV_EMU_SCHED'] = sched
    env['ASAN_OPTIONS'] = 'detect_leaks=0'
    return subprocess.run([exe] + [str(a) for a in args], env=env, capture_output=True, text=True)


@pytest.fixture(scope='module')
def exe(tmp_path_factory):
    return build(tmp_path_factory.mktemp('emu_detectors'), asan=False)


@pytest.fixture(scope='module')
def exe_asan(tmp_path_factory):
    return build(tmp_path_factory.mktemp('emu_detectors_asan'), asan=True)


def test_missing_barrier_is_hidden_by_the_forward_order_and_shown_by_the_others(exe):
    assert run(exe, 'race', 0).stdout.strip() == 'wrong=0'              # the hidden race: producer threads happen to run first
    assert run(exe, 'race', 0, sched='forward').stdout.strip() == 'wrong=0'
    assert run(exe, 'race', 0, sched='reverse').stdout.strip() == 'wrong=192'
    assert run(exe, 'race', 0, sched='waves-reverse').stdout.strip() == 'wrong=192'
    wrong = [int(run(exe, 'race', 0, sched=f'random:{s}').stdout.strip().split('=')[1]) for s in (1, 2, 3)]
    assert all(0 < w < 192 for w in wrong), wrong
    for sched in (None, 'reverse', 'waves-reverse', 'random:1', 'random:2'):   # with the barrier every order is right
        assert run(exe, 'race', 1, sched=sched).stdout.strip() == 'wrong=0'


def test_unknown_schedule_is_refused(exe):
    r = run(exe, 'race', 1, sched='sideways')
    assert r.returncode != 0 and 'MV_EMU_SCHED' in r.stderr


@pytest.mark.parametrize('what', ['lds_oob', 'global_oob'])
def test_address_sanitizer_build_stops_one_element_behind_a_buffer(exe_asan, what):
    ok = run(exe_asan, what, 0)
    assert ok.returncode == 0 and ok.stdout.strip() == 'read=0', ok.stderr
    bad = run(exe_asan, what, 1)
    assert bad.returncode != 0 and 'heap-buffer-overflow' in bad.stderr, bad.stderr[-2000:]
    for sched in ('reverse', 'random:1'):   # the fibers' stack switches are announced in every order
        assert run(exe_asan, what, 0, sched=sched).returncode == 0


def test_poison_mode_fills_unwritten_lds_and_device_blocks(exe):
    assert run(exe, 'uninit', 0).stdout.strip() == 'lds=0 global=0'
    assert run(exe, 'uninit', 0, poison=True).stdout.strip() == 'lds=-1 global=-1'


def test_missing_counted_wait_is_hidden_by_eager_transfers_and_shown_by_lazy_ones(exe):
    assert run(exe, 'dma', 0).stdout.strip() == 'wrong=0'                       # the default: every transfer lands at issue
    assert run(exe, 'dma', 0, dma='lazy').stdout.strip() == 'wrong=256'         # as late as the hardware allows: nothing has landed
    assert run(exe, 'dma', 0, dma='lazy', sched='reverse').stdout.strip() == 'wrong=256'
    for dma in (None, 'lazy', 'lazy-sync'):
        assert run(exe, 'dma', 1, dma=dma).stdout.strip() == 'wrong=0'          # with wait_vm<0>() in front of the barrier
    r = run(exe, 'dma', 1, dma='sometimes')
    assert r.returncode != 0 and 'MV_EMU_DMA' in r.stderr


def test_missing_lgkmcnt_wait_is_hidden_by_eager_fragment_reads_and_shown_by_lazy_ones(exe):
    assert run(exe, 'ldsread', 0).stdout.strip() == 'wrong=0'                  # the default: a hand-issued LDS read delivers at once
    assert run(exe, 'ldsread', 0, lds='lazy').stdout.strip() == 'wrong=64'     # only a counted wait delivers: the register still holds NaNs
    assert run(exe, 'ldsread', 1, lds='lazy').stdout.strip() == 'wrong=0'
    assert run(exe, 'ldsread', 1, lds='lazy', sched='reverse', dma='lazy').stdout.strip() == 'wrong=0'


def test_fuzzer_runs_random_geometries_of_every_family_clean():
    """tools/emu_fuzz.py: a handful of seeded random launch geometries of five kernel families (the tool knows fifteen) against the layer checks (the long runs and the
    checking modes are a tool, profiles/r13a_emu_check.log); a refused geometry is fine, a wrong value is not"""
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'emu_fuzz.py'), 'conv2ds,conv1d,fbank,melspec,model', '4', '--jobs', '4', '--seed', '11'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if ' ok, ' in l]
    assert len(lines) == 5 and all(' 0 failed, 0 workers crashed' in l for l in lines), r.stdout


def test_tampered_descriptors_are_refused_or_harmless_never_a_crash():
    """tools/emu_bad_args.py: every field of a valid MvConv1dDesc / MvConv2dsDesc replaced by a value a caller can get wrong (null
    tensor, 0, -1, a row length that is no multiple of the vector width, an enum out of range): the entry point returns an error code with a
    message or runs a call that is valid in itself -- the process survives all of them (include/mvector_hip.h: fail loudly, never crash)"""
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'emu_bad_args.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    summary = [l for l in r.stdout.splitlines() if 'SUMMARY' in l][0]
    n, rejected = int(summary.split()[2]), int(summary.split()[5])
    assert n > 380 and rejected > 300 and ' 0 crashed' in summary, summary   # (round 5: one descriptor fewer -- the fp32 conv2d form left the library)
    accepted = [l for l in r.stdout.splitlines() if 'ACCEPTED' in l][0]
    for must_reject in ('pre_act=99', 'post_act=-1', 'pad_mode=99', 'mv_conv1d_forward.ldx=', 'mv_conv1d_forward.ldy=', 'x=None', 'y=None', 'oscale'):
        assert must_reject not in accepted, must_reject
