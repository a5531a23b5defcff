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


# the files the implicit-GEMM convolution family is compiled from (profiles/*_pmc_traffic.json is stamped with their hash)
CONV_SOURCES = ('conv_igemm.hip', 'conv_resident.hip', 'conv_common.h', 'common.h', 'test_hooks.h')


def source_hash(files=None):
    """sha256 (first 16 hex digits) over the names and contents of the library's sources: every csrc/*.hip, csrc/*.h and include/vd3d.h,
    or only ``files`` (names inside csrc/).  The library carries the whole-tree value of the sources it was built from
    (``vd3d_source_hash``)."""
    import hashlib
    if files is None:
        paths = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))] + [os.path.join(REPO, 'include', 'vd3d.h')]
    else:
        paths = [os.path.join(CSRC, f) for f in sorted(files)]
    h = hashlib.sha256()
    for q in paths:
        h.update(os.path.basename(q).encode() + b'\0')
        h.update(open(q, 'rb').read())
        h.update(b'\0')
    return h.hexdigest()[:16]


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
    # common.hip carries the hash of ALL sources (vd3d_source_hash): recompiled whenever that hash moves
    sh = source_hash()
    stamp = os.path.join(OBJ, 'source_hash.txt')
    hash_moved = (not os.path.exists(stamp)) or open(stamp).read().strip() != sh
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + '.o')
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs) or (s == 'common.hip' and hash_moved):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        extra = ['-DVD3D_SRC_HASH="%s"' % sh] if os.path.basename(src) == 'common.hip' else []
        cmd = [HIPCC] + FLAGS + extra + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if verbose:
            print('[vd3d build] compiled', os.path.basename(src))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    open(stamp, 'w').write(sh)
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
