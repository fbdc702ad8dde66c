"""Build a VARIANT of the library for same-box A/B runs:   python scripts/build_variant_lib.py <tag> [<git rev>] [-DFLAG ...]
-> simpleaicv_pytorch_training_examples_amd/libsaicv_hip_<tag>.so, from the csrc/ of <git rev> (default: the working tree) compiled
with the extra flags.  Built HERE before the GPU call (built .so files travel with the snapshot); the product library is untouched.
Use with scripts/with_lib.py:   python scripts/with_lib.py <tag> bench.py --model resnet50 ..."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleaicv_pytorch_training_examples_amd import build as b  # noqa: E402


def main():
    tag = sys.argv[1]
    rev = next((a for a in sys.argv[2:] if not a.startswith('-')), None)
    flags = [a for a in sys.argv[2:] if a.startswith('-')]
    with tempfile.TemporaryDirectory() as tmp:
        src = b.CSRC
        if rev:
            src = os.path.join(tmp, 'pkg', 'csrc')
            os.makedirs(src)
            rel = os.path.relpath(b.CSRC, ROOT)
            names = subprocess.check_output(['git', 'ls-tree', '--name-only', rev, rel + '/'], cwd=ROOT).decode().split()
            for n in names:
                if n.endswith(('.hip', '.h')):
                    open(os.path.join(src, os.path.basename(n)), 'wb').write(subprocess.check_output(['git', 'show', f'{rev}:{n}'], cwd=ROOT))
            inc = os.path.join(tmp, 'include')
            os.makedirs(inc)
            open(os.path.join(inc, 'saicv_hip.h'), 'wb').write(subprocess.check_output(['git', 'show', f'{rev}:include/saicv_hip.h'], cwd=ROOT))
        objs, procs = [], []
        for s in b.SOURCES:
            f = os.path.join(src, s)
            if not os.path.exists(f):
                continue
            o = os.path.join(tmp, s.replace('.hip', '.o'))
            objs.append(o)
            procs.append(subprocess.Popen([b._hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get(s, []) + flags + ['-c', f, '-o', o]))
        if any(p.wait() for p in procs):
            raise SystemExit('hipcc failed')
        out = os.path.join(b.HERE, f'libsaicv_hip_{tag}.so')
        subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + ['-ldl'])
        print(out)


if __name__ == '__main__':
    main()
