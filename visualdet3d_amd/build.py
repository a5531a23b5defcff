"""Build libvd3d_hip.so (every HIP kernel + the C-ABI) for gfx950 with hipcc, in-tree.

    python -m visualdet3d_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands next to this file (git-ignored, but it travels to the GPU
box with the repo snapshot)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libvd3d_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', os.path.join(REPO, 'include'), '-I', CSRC,
         '-Wno-unused-result']


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, tuning=False):
    """tuning=True: the -DVD3D_TUNING variant (experimental conv tiles + timing ablations) -> libvd3d_hip_tuning.so; the
    product library never contains those kernels."""
    global OBJ, LIB
    obj_dir, lib_path, flags = OBJ, LIB, FLAGS
    if tuning:
        obj_dir, lib_path, flags = OBJ + '_tuning', os.path.join(HERE, 'libvd3d_hip_tuning.so'), FLAGS + ['-DVD3D_TUNING']
    return _build(force, verbose, obj_dir, lib_path, flags)


def _build(force, verbose, OBJ, LIB, FLAGS):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(REPO, 'include', 'vd3d.h')]
    jobs = []
    objs = []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + '.o')
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if verbose:
            print('[vd3d build] compiled', os.path.basename(src))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    if force or jobs or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
        if verbose:
            print('[vd3d build] linked', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, tuning='--tuning' in sys.argv)
