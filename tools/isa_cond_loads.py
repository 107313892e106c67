"""Register loads inside blocks a forward branch may skip (s_cbranch_* .LBBx ... .LBBx:), per kernel, from the assembly tools/isa_audit.py leaves under
/tmp/isa_audit (run that first): at the join the compiler's wait-count insertion assumes such a load was NOT issued, so every later s_waitcnt vmcnt(N) counts it
out and comes out too strict (\"skippable blocks\"; the Res2Net chain's fragment requests, round 6).
usage: python tools/isa_audit.py >/dev/null; python tools/isa_cond_loads.py | sort -t= -k2 -n -r | head -40"""
import re, glob, subprocess
# register loads that sit in a block a forward branch may skip (s_cbranch_* .LBBx ... .LBBx:), i.e. loads the wait-count insertion must assume "not issued" at the join
for asm in sorted(glob.glob('/tmp/isa_audit/*.s')):
    kernel = None; lines = []
    ks = {}
    for line in open(asm):
        m = re.match(r'^(_Z\w+):', line)
        if m: kernel = m.group(1); ks[kernel] = []; continue
        if kernel: ks[kernel].append(line.strip())
    for k, L in ks.items():
        hits = 0; mf = 0
        for i, t in enumerate(L):
            m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', t)
            if not m: continue
            lbl = m.group(1) + ':'
            # forward label within 60 lines
            for j in range(i + 1, min(i + 60, len(L))):
                if L[j].startswith(lbl):
                    blk = L[i + 1:j]
                    nl = sum(1 for q in blk if re.match(r'(global|buffer)_load_dword', q) and 'lds' not in q)
                    if nl: hits += nl
                    break
        mf = sum(1 for q in L if q.startswith('v_mfma'))
        if hits:
            name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
            print(f'{asm.split("/")[-1]:14s} cond-loads={hits:4d} mfma={mf:5d}  {name[:110]}')
