"""Race / memory-safety sweep of the kernels on the SIMT emulator (no GPU): runs tests/test_emu_kernels.py -- every kernel family, the
tiny models end to end -- once per checking mode of tests/emu/hip_emu.h and prints one summary line per mode.

    python tools/emu_check.py                       # all modes: asan, ubsan, poison, lazy-dma, lazy-dma+reverse, lazy-lds, lazy-all+random:1, reverse, waves-reverse, random:1, random:2
    python tools/emu_check.py asan random:7         # chosen modes
    python tools/emu_check.py -k conv2ds asan       # a subset of the cases (pytest -k)

  asan            the emulator build with -fsanitize=address (tests/emu/build_asan): every global buffer and every block's dynamic LDS is a
                  heap block of exactly its size, so an index that leaves its buffer is reported -- also one the GPU would absorb silently.
  ubsan           -fsanitize=signed-integer-overflow,shift,integer-divide-by-zero,bounds,null,float-cast-overflow (tests/emu/build_ubsan,
                  no recovery): index and size arithmetic that overflows, on the host side and in the kernels.
  poison          MV_EMU_POISON=1: dynamic LDS and hipMalloc blocks start as 0xFF bytes (NaN) instead of zeros -- a result that depends on
                  storage nobody wrote (the device leaves the previous owner's bytes there) turns into a NaN.
  lazy-dma[+<order>]
                  MV_EMU_DMA=lazy (+ poison, optionally with a thread order): the LDS-DMA transfers land as late as the kernels' counted s_waitcnt vmcnt allow (the N
                  youngest of a wait_vm<N>() stay in flight, only counted waits retire anything) -- a fragment read that no counted wait +
                  barrier covers sees NaNs.  The default emulator lets every transfer land at issue (the other extreme).
  lazy-lds[+<order>]
                  MV_EMU_LDS=lazy (+ poison): the hand-issued LDS fragment reads (inline-assembly ds_read_b128) deliver only when a counted
                  s_waitcnt lgkmcnt of the issuing thread (lds_wait<N>, the WAIT of the MFMA groups) retires them; until then the destination
                  register holds NaNs -- an MFMA group that starts on a fragment its wait count does not cover computes NaNs.
  lazy-all[+<order>]  lazy-dma and lazy-lds together.
  reverse | waves-reverse | random:<seed>
                  MV_EMU_SCHED: the order in which a block's threads run between barriers; a dependency no barrier orders gives a wrong result
                  in one of them and the tests' expected values catch it (tests/test_emu_detectors.py shows both detectors at work).
MV_EMU_CUS=<n> in the environment sets the emulated chip's compute units (default 8): 256 = every batch a sub-chip one, 1 = a chip-filling one.
What no mode models is time itself (how long a wait takes, MFMA-to-register hazards inside the assembly groups): that stays with the device-side
detectors, tools/stress_determinism.py and the bit-identity tests.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_RT = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
UBSAN_RT = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so')


def run_mode(mode, k, workers, slow):
    env = dict(os.environ)
    env.pop('MV_EMU_SCHED', None)
    env.pop('MV_EMU_SANITIZE', None)
    env.pop('MV_EMU_POISON', None)
    env.pop('MV_EMU_DMA', None)
    env.pop('MV_EMU_LDS', None)
    if slow:
        env['MV_SLOW_EMU'] = '1'
    logdir = None
    if mode == 'asan':
        assert ASAN_RT, 'no AddressSanitizer runtime under /opt/rocm/lib/llvm'
        logdir = tempfile.mkdtemp(prefix='emu_asan_')
        env.update(MV_EMU_SANITIZE='address', LD_PRELOAD=ASAN_RT[0], ASAN_SYMBOLIZER_PATH='/opt/rocm/lib/llvm/bin/llvm-symbolizer',
                   ASAN_OPTIONS=f'detect_leaks=0:verify_asan_link_order=0:halt_on_error=1:log_path={logdir}/log')
    elif mode == 'ubsan':
        assert UBSAN_RT, 'no UndefinedBehaviorSanitizer runtime under /opt/rocm/lib/llvm'
        logdir = tempfile.mkdtemp(prefix='emu_ubsan_')
        env.update(MV_EMU_SANITIZE='undefined', LD_PRELOAD=UBSAN_RT[0], UBSAN_SYMBOLIZER_PATH='/opt/rocm/lib/llvm/bin/llvm-symbolizer',
                   UBSAN_OPTIONS=f'print_stacktrace=1:halt_on_error=1:log_path={logdir}/log')
    elif mode == 'poison':
        env['MV_EMU_POISON'] = '1'
    elif mode.startswith('lazy-lds') or mode.startswith('lazy-all'):   # [+<order>]
        env['MV_EMU_POISON'] = '1'
        env['MV_EMU_LDS'] = 'lazy'
        if mode.startswith('lazy-all'):
            env['MV_EMU_DMA'] = 'lazy'
        if '+' in mode:
            env['MV_EMU_SCHED'] = mode.split('+', 1)[1]
    elif mode.startswith('lazy-dma'):   # lazy-dma | lazy-dma+reverse | lazy-dma+random:<seed>
        env['MV_EMU_POISON'] = '1'
        env['MV_EMU_DMA'] = 'lazy'
        if '+' in mode:
            env['MV_EMU_SCHED'] = mode.split('+', 1)[1]
    else:
        env['MV_EMU_SCHED'] = mode
    # build first, in this process' environment (the workers then find the stamp)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tests', 'emu', 'build_emu.py')], env={**env, 'LD_PRELOAD': ''}, stdout=subprocess.DEVNULL)
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_emu_kernels.py'), '-q', '-p', 'no:cacheprovider', '-n', str(workers),
           '--max-worker-restart=200']
    if k:
        cmd += ['-k', k]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    tail = [l for l in r.stdout.splitlines() if re.search(r'\d+ (passed|failed)', l)]
    print(f'{mode:14s} rc={r.returncode} {tail[-1].strip() if tail else r.stdout[-300:]!r} wall {time.time() - t0:.0f} s', flush=True)
    for l in r.stdout.splitlines():
        if l.startswith('FAILED'):
            print('   ', l)
    if logdir:
        sites = {}
        for f in glob.glob(os.path.join(logdir, 'log*')):
            for l in open(f):
                if l.startswith('SUMMARY') or 'runtime error' in l:
                    sites[l.strip()] = sites.get(l.strip(), 0) + 1
        for s, n in sorted(sites.items()):
            print(f'    {n} x {s}')
        if not sites:
            print('    no sanitizer report')
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('modes', nargs='*', default=['asan', 'ubsan', 'poison', 'lazy-dma', 'lazy-dma+reverse', 'lazy-lds', 'lazy-all+random:1', 'reverse', 'waves-reverse', 'random:1', 'random:2'])
    ap.add_argument('-k', default='')
    ap.add_argument('--slow', action='store_true', help='also the cases behind MV_SLOW_EMU=1 (CAM++ end to end, the largest conv2d cases)')
    ap.add_argument('-n', type=int, default=min(8, os.cpu_count() or 1))
    args = ap.parse_args()
    rcs = [run_mode(m, args.k, args.n, args.slow) for m in args.modes]
    sys.exit(1 if any(rcs) else 0)


if __name__ == '__main__':
    main()
