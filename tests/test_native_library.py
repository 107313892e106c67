"""CPU checks of the built product library: it exists, exports every symbol include/mvector_hip.h declares,
and the Python binding refuses to run without it (no fallback).  No compute calls are made here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, PKG


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'mvector_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mv_[a-z0-9_]+)\s*\(', text)))


def test_library_built_and_exports_every_declared_symbol():
    import __graft_entry__
    lib = __graft_entry__.build()
    cdll = ctypes.CDLL(lib)
    syms = _header_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(cdll, s)]
    assert not missing, missing
    from mvector import _hip
    assert set(_hip.EXPORTED_SYMBOLS) == set(syms)
    _hip.bind(cdll)
    assert cdll.mv_abi_version() == 5
    assert cdll.mv_conv1d_packed_elems(192, 80, 5) == 192 * 5 * 128


def test_library_exports_nothing_but_the_declared_entry_points():
    """`nm -D` listed 99 C++ internals (mv::conv1d_launch, MvModelBase::..., kernel handles, even global-namespace
    fbank_launch) beside the 60 mv_* names -- the library is dlopen'ed into a process that already holds torch + ROCm.  Now: -fvisibility=hidden,
    default visibility pushed around the header's declarations, and a linker version script (csrc/exports.map) that makes the kernel handles
    hipcc exports regardless local as well.  The dynamic symbol table IS include/mvector_hip.h."""
    import shutil
    import subprocess
    import __graft_entry__
    lib = __graft_entry__.build()
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    out = subprocess.run([nm, '-D', '--defined-only', lib], capture_output=True, text=True, check=True).stdout
    names = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert names == _header_symbols(), sorted(set(names) ^ set(_header_symbols()))


def test_binding_fails_loudly_without_library(monkeypatch):
    from mvector import _hip
    monkeypatch.setattr(_hip, '_lib', None)
    monkeypatch.setattr(_hip, 'LIB_PATH', os.path.join(PKG, 'mvector', 'lib', 'does_not_exist.so'))
    with pytest.raises(RuntimeError, match='has not been built'):
        _hip.lib()


def test_product_package_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                src = open(os.path.join(dirpath, f), errors='ignore').read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or 'oracle/' in src and f.endswith('.py') and 'import' in src and re.search(r'import.*oracle', src):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_kernel_sources_have_one_architecture_and_no_emulator_branches():
    """the SIMT emulator of the test-suite lives behind ONE header (<arch/gfx950.h>, twin under tests/emu/arch);
    no kernel source carries an emulator #ifdef, a CUDA / multi-platform guard or a hipify artefact."""
    import glob
    import re
    from conftest import PKG
    bad = re.compile(r'MV_EMU|__HIP_PLATFORM|__CUDACC__|__CUDA_ARCH__|cuda_runtime|hipify')
    srcs = [p for ext in ('*.hip', '*.cpp', '*.h') for p in glob.glob(os.path.join(PKG, 'csrc', '**', ext), recursive=True)]
    assert len(srcs) > 15
    for p in srcs:
        text = open(p).read()
        assert not bad.search(text), p
        assert 'getenv' not in text, f'{p}: the library reads no environment variable (round 5: the last test hook became MvConv1dDesc.persist_blocks_hint)'
    assert os.path.exists(os.path.join(PKG, 'csrc', 'arch', 'gfx950.h')) and os.path.exists(os.path.join(ROOT, 'tests', 'emu', 'arch', 'gfx950.h'))


def test_counted_waits_of_the_cam_block_kernel_match_the_generated_code():
    """cam_dense_block_kernel's layer entry waits with `s_waitcnt vmcnt(12 + y stores)`: the twelve youngest vector-memory operations of a wave must
    be the next layer's k = 3 weight loads (camblock.hip, the lazy layer entry).  That number is a property of the GENERATED code, so it is asserted on
    the gfx950 assembly hipcc produces here (no GPU needed): per layer 12 + 9 sixteen-byte parameter loads, 4 + 2 four-byte ones, 3 y stores from
    inline assembly, no FLAT and no scratch instruction (a pointer that lost its address space, a spill) -- round 5's ISA audit as a regression test."""
    import shutil
    import subprocess
    import sys
    import tempfile
    sys.path.insert(0, PKG)
    import build_native
    if not (shutil.which(build_native.HIPCC) or os.path.exists(build_native.HIPCC)):
        pytest.skip('hipcc not found')
    src = os.path.join(PKG, 'csrc', 'camblock.hip')
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 'camblock.s')
        subprocess.check_call([build_native.HIPCC] + build_native.FLAGS + build_native._file_flags(src) +
                              ['-Wno-inline-asm', '--cuda-device-only', '-S', '-x', 'hip', src, '-o', asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    body = text[text.index('_ZN2mv22cam_dense_block_kernelENS_12CamBlockArgsE:'):]
    body = body[:body.index('.Lfunc_end')]
    lines = [l.strip() for l in body.split('\n')]
    loop = lines[max(i for i, l in enumerate(lines) if 'Loop Header: Depth=1' in l):]   # the layer loop is the kernel's last top-level loop
    count = lambda seq, prefix: sum(1 for l in seq if l.startswith(prefix))
    assert count(lines, 'flat_') == 0 and count(lines, 'scratch_') == 0
    assert count(loop, 'global_load_dwordx4') == 21, 'context parameters (9) + k = 3 weights (12) per layer'
    assert count(loop, 'global_load_dword ') == 6, 'BN1 tables (4) + the two context biases per layer'
    assert count(loop, 'global_store_dwordx2') == 3, 'one y store per time tile of a wave'
    assert count(loop, 'v_mfma') == 20 + 36, 'the stage (20) and the k = 3 phase (36), each once'
    # the ORDER the count relies on.  Behind the tail's last LDS-DMA transfer the only vector-memory instructions of the layer loop are the
    # (up to) three y stores and then the twelve k = 3 weight loads -- nothing sunk below them, none merged or re-materialised -- and the loop
    # holds the four counted entry waits vmcnt(12 + n_st), n_st = 0..3.  (The Res2Net chain's counted wait behind its last stage, res2.hip
    # MV_VM_LOADS(6), needs no such guard: the tracked parameter loads it announces are OLDER than the four fragment loads the count names, and the
    # compiler protects their registers with its own waits; a different order can only make that wait stronger.)
    last_dma = max(i for i, l in enumerate(loop) if l.startswith('global_load_lds'))
    tail_vm = [l.split()[0] for l in loop[last_dma + 1:] if re.match(r'(global_|buffer_|flat_|scratch_)', l)]
    assert tail_vm == ['global_store_dwordx2'] * 3 + ['global_load_dwordx4'] * 12, tail_vm
    assert {l for l in loop if l.startswith('s_waitcnt vmcnt')} >= {f's_waitcnt vmcnt({12 + n})' for n in range(4)}   # (block layout puts three of them behind the stage loop)
    meta = dict(re.findall(r'\.(vgpr_spill_count|sgpr_count|vgpr_count):\s+(\d+)', text[text.index('.name:           _ZN2mv22cam_dense_block_kernel'):][:1500]))
    assert int(meta['vgpr_spill_count']) == 0


def test_no_kernel_on_a_model_path_has_flat_or_scratch_instructions():
    """tools/isa_audit.py over every csrc/*.hip (hipcc -S for gfx950, no GPU): round 5 found a CAM++ kernel whose parameter pointers had lost their address
    space (FLAT loads count on lgkmcnt: every LDS wait behind one waited for L2), a pointer select that had become a table in scratch memory, and a uniform
    branch per MFMA.  None of the three may come back: no FLAT instruction anywhere; scratch (spill) instructions only in the kernels listed here -- none of
    which a shipped configuration launches (the 16-group Fbank instantiations serve windows of no listed length, STATS = 2 of the double-buffer conv no model);
    no kernel with MFMAs that sit alone in their basic block."""
    import importlib.util
    import shutil
    import sys
    sys.path.insert(0, PKG)
    import build_native
    if not (shutil.which(build_native.HIPCC) or os.path.exists(build_native.HIPCC)):
        pytest.skip('hipcc not found')
    spec = importlib.util.spec_from_file_location('isa_audit', os.path.join(ROOT, 'tools', 'isa_audit.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    may_spill = re.compile(r'fbank_kernel<16,|fbank_tile_kernel<16,|conv1d_glds_persistent_kernel<true, 2>')
    seen = 0
    for fname, kernels in mod.audit().items():
        for name, k in kernels.items():
            seen += 1
            assert k['flat'] == 0, f'{fname}: {name}: {k["flat"]} FLAT instructions (a pointer without its address space: MV_GLOBAL_PTR)'
            assert k['lone'] <= 2, f'{fname}: {name}: {k["lone"]} MFMAs alone in a basic block (a branch per MFMA)'
            if not may_spill.search(name):
                assert k['scratch'] == 0 and k['vgpr_spill'] == 0, f'{fname}: {name}: scratch {k["scratch"]}, spilled VGPRs {k["vgpr_spill"]}'
    assert seen > 150
