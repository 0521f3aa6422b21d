"""Build libaodhip.so in-tree: `python -m aod_meh_hua_amd.build` (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libaodhip.so')
# -ffp-contract=off: geometry / loss kernels must reproduce the reference's fp32 op order bit for bit
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def source_digest():
    """sha256[:16] over the kernel sources + the C-ABI header: the tag that ties a committed PMC summary (profiles/pmc_*.json, written by
    tools/profile/pmc_passes.sh) to the build it was measured on -- bench.py merges such a file into its line only when the tags agree."""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'pointwise.h'), os.path.join(CSRC, 'conv_x3p.h'), os.path.join(HERE, '..', 'include', 'aod_hip.h')]:
        if os.path.exists(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'pointwise.h'), os.path.join(CSRC, 'conv_x3p.h'),
                        os.path.join(HERE, '..', 'include', 'aod_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(HERE, 'lib', 'obj')
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    procs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [hipcc] + FLAGS + (['-x', 'hip'] if src.endswith('.hip') else []) + ['-c', src, '-o', obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
