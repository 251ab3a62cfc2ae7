#!/usr/bin/env python3
"""Static checks on the gfx950 ISA of the kernels (no GPU needed; hipcc cross-compiles).

  tools/isa_audit.py waits [file.hip ...]
      per kernel: the s_waitcnt vmcnt(...) the COMPILER put inside loops (not the ones written
      as inline asm).  Two defects of round 3 were visible this way and only this way:
      a vmcnt(0) in front of every stage's first ds_read_b64_tr_b16 of the bf16 attention
      (LDS reads behind an LDS-DMA in flight wait for it unless alias scopes separate them),
      and a load / wait / add chain per slice in the QKV kernel's prologue.

  tools/isa_audit.py diff <git-rev> [file.hip ...]
      every kernel of the working tree against the same kernel at <git-rev>, instruction by
      instruction (labels and comments normalised; a defaulted template parameter appended to
      a kernel's signature is folded back): which kernels are unchanged, changed, new, gone.
      Used to show that a prepared-but-unmeasured variant leaves the shipped kernels alone.

Assembly goes to $TMPDIR/wn_isa/{work,<rev>}/ and is reused when newer than the source.
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(os.environ.get('TMPDIR', tempfile.gettempdir()), 'wn_isa')


def _flags(root, f):
    sys.path.insert(0, root)
    try:
        for m in [k for k in sys.modules if k.startswith('wenet_amd')]:
            del sys.modules[m]
        from wenet_amd import build
        return [x for x in build.FLAGS + build.EXTRA_FLAGS.get(f, []) if x != '-fPIC']
    finally:
        sys.path.pop(0)


def assemble(root, tag, files):
    out = os.path.join(OUT, tag)
    os.makedirs(out, exist_ok=True)
    csrc = os.path.join(root, 'wenet_amd', 'csrc')
    files = files or sorted(f for f in os.listdir(csrc) if f.endswith('.hip'))
    hdr_time = max(os.path.getmtime(os.path.join(csrc, h)) for h in os.listdir(csrc)
                   if h.endswith('.h'))
    jobs = []
    for f in files:
        s = os.path.join(out, f[:-4] + '.s')
        src = os.path.join(csrc, f)
        if os.path.exists(s) and os.path.getmtime(s) > max(os.path.getmtime(src), hdr_time):
            continue
        jobs.append(['/opt/rocm/bin/hipcc'] + _flags(root, f) +
                    ['-S', '--cuda-device-only', src, '-o', s])
    with concurrent.futures.ThreadPoolExecutor(8) as ex:
        for cmd, r in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True,
                                                                 text=True), jobs)):
            if r.returncode != 0:
                raise SystemExit(f'hipcc failed: {" ".join(cmd)}\n{r.stderr[-2000:]}')
    return {f: os.path.join(out, f[:-4] + '.s') for f in files}


_DEFAULTED = re.compile(r'(I(?:L[ib]\d+E)+)(Lb0E)(EEv)')


def kernels(path, strip_defaulted=False):
    """{mangled name: [normalised instruction lines]}; with per-kernel wait statistics."""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r'^(_Z\S+):\s', line)
        if m and '@' in line:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith('.Lfunc_end'):
            cur = None
            continue
        out[cur].append(line.rstrip('\n'))
    return out


def _norm(lines):
    res = []
    for l in lines:
        l = re.sub(r'\.LBB\d+_', '.LBB_', l)
        l = re.sub(r';.*$', '', l).rstrip()
        if l.strip():
            res.append(l)
    return res


def demangle(name):
    r = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    return r.replace('wn::(anonymous namespace)::', '')


def cmd_waits(files):
    for f, path in assemble(ROOT, 'work', files).items():
        for name, lines in kernels(path).items():
            inasm = inloop = False
            n_loop = n_loop0 = n_asm = 0
            for l in lines:
                if l.startswith('.LBB'):
                    inloop = 'in Loop' in l or 'Loop Header' in l
                elif 'ASMSTART' in l:
                    inasm = True
                elif 'ASMEND' in l:
                    inasm = False
                elif 's_waitcnt' in l and 'vmcnt' in l:
                    if inasm:
                        n_asm += 1
                    elif inloop:
                        n_loop += 1
                        n_loop0 += 'vmcnt(0)' in l
            if n_loop:
                print(f'{f:22s} compiler vmcnt in loops {n_loop:3d} (vmcnt(0) {n_loop0:3d})  '
                      f'asm vmcnt {n_asm:3d}  {demangle(name)[:110]}')


def cmd_diff(rev, files):
    wt = os.path.join(OUT, 'tree_' + rev)
    if not os.path.isdir(wt):
        os.makedirs(wt)
        ar = subprocess.run(['git', '-C', ROOT, 'archive', rev, 'wenet_amd'], check=True,
                            capture_output=True).stdout
        subprocess.run(['tar', '-x', '-C', wt], input=ar, check=True)
    base = assemble(wt, rev, files)
    work = assemble(ROOT, 'work', files)
    tot = [0, 0, 0, 0]
    for f in sorted(work):
        if f not in base or not os.path.exists(base[f]):
            print(f'{f:22s} (new file)')
            continue
        A = {k: _norm(v) for k, v in kernels(base[f]).items()}
        B = {k: _norm(v) for k, v in kernels(work[f]).items()}
        # a defaulted trailing `bool = false` template parameter added since <rev>
        fold = {}
        for k in B:
            k2 = k
            while k2 not in A:
                k3 = _DEFAULTED.sub(lambda m: m.group(1) + m.group(3), k2, count=1)
                if k3 == k2:
                    break
                k2 = k3
            if k not in A and k2 in A:
                fold[k] = k2
        B2 = {}
        for k, v in B.items():
            k2 = fold.get(k, k)
            if k2 != k:
                v = [x.replace(k, k2) for x in v]
            B2[k2] = v
        same = [k for k in A if k in B2 and A[k] == B2[k]]
        diff = [k for k in A if k in B2 and A[k] != B2[k]]
        gone = [k for k in A if k not in B2]
        new = [k for k in B2 if k not in A]
        tot = [tot[0] + len(same), tot[1] + len(diff), tot[2] + len(gone), tot[3] + len(new)]
        if diff or gone or new:
            print(f'{f:22s} same {len(same)} changed {len(diff)} gone {len(gone)} new {len(new)}')
            for tag, ks in (('changed', diff), ('gone', gone), ('new', new)):
                for k in ks:
                    print(f'    {tag:8s}{demangle(k)[:120]}')
    print(f'total: {tot[0]} kernels unchanged, {tot[1]} changed, {tot[2]} gone, {tot[3]} new '
          f'(against {rev})')
    return 1 if tot[1] or tot[2] else 0


if __name__ == '__main__':
    if len(sys.argv) >= 2 and sys.argv[1] == 'waits':
        cmd_waits(sys.argv[2:])
    elif len(sys.argv) >= 3 and sys.argv[1] == 'diff':
        sys.exit(cmd_diff(sys.argv[2], sys.argv[3:]))
    else:
        raise SystemExit(__doc__)
