"""Build libwenet_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m wenet_amd.build          # incremental
    python -m wenet_amd.build --force

hipcc cross-compiles without a GPU; the .so lands next to this file so that it
travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '..', 'build', 'obj')
LIB = os.path.join(HERE, 'libwenet_amd.so')
SOURCES = ['gemm.hip', 'gemm_bf16.hip', 'gemm_bf16s.hip', 'gemm_bf16p.hip', 'ffn_fused.hip', 'gemm_rowln.hip', 'gemm_x6.hip', 'ffn_x6f.hip', 'gemm_x6r.hip', 'gemm_x6r512.hip', 'attn_search.hip', 'encoder_kernels.hip', 'attention_bf16.hip', 'attention_x6.hip', 'ctc.hip', 'fbank.hip',
           'logmel.hip', 'model.hip', 'cabi.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-fno-gpu-rdc', '-Wno-unused-result']
# WN_ABLATION=1 python -m wenet_amd.build: a MEASUREMENT build that also compiles the kernel
# variants which leave out parts of a kernel (wrong results by design) and superseded stage
# shapes; the product build has none of them (wn_tune_set refuses their keys / values)
if os.environ.get('WN_ABLATION') == '1':
    FLAGS.append('-DWN_ABLATION')
# WN_EXACT=1 python -m wenet_amd.build: a VALIDATION build whose SiLU / GLU gates and CTC
# log-softmax sums run on the exact expf / exp2f / IEEE division instead of v_exp_f32 /
# v_rcp_f32 (csrc/common.h); never the product library
if os.environ.get('WN_EXACT') == '1':
    FLAGS.append('-DWN_EXACT_TRANSCENDENTALS')
# per-source extras: the one-wave-per-SIMD kernel pins its VALU slices between MFMA pairs;
# SLP-packed f32 ops (v_pk_*) would undo the spacing (MI355X_MICROARCH.md: an anti-lever
# beside MFMAs)
EXTRA_FLAGS = {'ffn_x6f.hip': ['-fno-slp-vectorize'], 'gemm_x6r.hip': ['-fno-slp-vectorize'], 'gemm_x6r512.hip': ['-fno-slp-vectorize'],
               # the softmax of the bf16 attention kernels is VALU-bound: no v_pk_add + v_mov
               # packing of the row sums
               'attention_bf16.hip': ['-fno-slp-vectorize']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'),
              '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def build_host_ext(force: bool = False) -> str:
    """The CPython helper that turns the n-best arrays of a batch into the reference's list
    fields (cext/nbest_lists.c; host code, gcc).  Lands next to this file like the HIP
    library."""
    import sysconfig
    src = os.path.join(HERE, 'cext', 'nbest_lists.c')
    out = os.path.join(HERE, '_nbest_lists' + sysconfig.get_config_var('EXT_SUFFIX'))
    if (not force and os.path.exists(out)
            and os.path.getmtime(out) >= os.path.getmtime(src)):
        return out
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        raise RuntimeError('gcc not found')
    cmd = [cc, '-O2', '-shared', '-fPIC', '-I' + sysconfig.get_paths()['include'], src,
           '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'gcc failed for nbest_lists.c:\n{r.stderr}')
    return out


def _stamp(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, 'rb') as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    try:
        build_host_ext(force)
    except Exception as e:  # noqa: BLE001 -- host-side list building only: search.py then
        # uses its numpy pass (same lists); the HIP library below is what must build
        print(f'warning: host helper not built ({e}); wenet_amd.search falls back to numpy',
              file=sys.stderr)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
               if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'wenet_amd.h'))
    stamp_file = os.path.join(OBJ, 'stamp')
    stamp = _stamp(headers + [os.path.join(CSRC, s) for s in SOURCES])
    if (not force and os.path.exists(LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stderr}')
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB
           ] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr}')
    # every kernel stub must resolve: dlopen with RTLD_NOW in a child process
    r = subprocess.run([sys.executable, '-c',
                        f'import ctypes, os; ctypes.CDLL({LIB!r}, mode=os.RTLD_NOW)'],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'{LIB} does not load:\n{r.stderr[-2000:]}')
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    if verbose:
        print(f'built {LIB} ({os.path.getsize(LIB) // 1024} KiB)')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
