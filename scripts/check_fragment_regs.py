"""Static check of the hand-scheduled LDS fragment reads (csrc/igemm.hip lds_rd128 / lgkm_release, csrc/pwstream.hip): the
`ds_read_b128` instructions sit in inline asm whose outputs the compiler believes valid at once, and the matching `s_waitcnt
lgkmcnt(N)` are asm statements with hand-counted N.  Nothing at compile time tells whether the register allocator copied or
spilled a fragment register inside that window, or whether a compiler-generated LDS / scalar-memory operation slipped in and made
the count wrong.  This walks the generated assembly of every kernel and keeps the queue of outstanding LGKM operations
the way the hardware counter does:

  * an LGKM operation (ds_*, s_load*, s_buffer_load*) is appended when it issues; `s_waitcnt ... lgkmcnt(n)` retires all but the
    youngest n (LDS operations return in order; while a scalar load is outstanding only lgkmcnt(0) retires anything);
  * an instruction that READS a destination register of a still-outstanding asm ds_read is a violation (used before its wait);
  * an instruction that WRITES one is a violation too (the returning data would land on top of it) -- including the compiler's
    own moves and any spill;
  * kernels that contain asm fragment reads must use no scratch memory.

The walk is linear per kernel (fall-through order); after an unconditional branch or s_endpgm the queue restarts empty, i.e. the
software-pipelined loops are checked as the straight-line runs they are, with the prologue standing in for the back edge.
Usage: python scripts/check_fragment_regs.py [file.s ...]   (without arguments pwstream.hip and igemm.hip are compiled first)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'simpleaicv_pytorch_training_examples_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-Wno-unused-result', '-Wno-unused-value', '-S', '--cuda-device-only']

REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])')
NO_DEST = ('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'ds_write', 'ds_store', 's_waitcnt', 'v_cmp', 'v_cmpx', 's_nop',
           'buffer_atomic', 'global_atomic', 'ds_add', 'ds_max', 'ds_min', 'exp', 's_barrier')


def assembly(source):
    src = os.path.join(CSRC, source)
    out = os.path.join(tempfile.gettempdir(), 'saicv_frag_' + source.replace('.hip', '.s'))
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + [src, '-o', out], stderr=subprocess.DEVNULL)
    return open(out).read()


def regs(tok_text):
    out = set()
    for kind, single, lo, hi in REG.findall(tok_text):
        if single:
            out.add((kind, int(single)))
        else:
            out.update((kind, r) for r in range(int(lo), int(hi) + 1))
    return out


def split_operands(ins):
    body = ins.split(';')[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else '', []
    return parts[0], [o.strip() for o in re.split(r',(?![^\[]*\])', parts[1])]


def check_kernel(name, lines):
    """-> (asm fragment reads, counted waits, violations)"""
    outstanding = []            # (is_smem, dest regs or empty set, line number, from_asm)
    in_asm = False
    n_reads = n_waits = 0
    bad = []
    for no, raw in lines:
        line = raw.strip()
        if line.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if line.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not line or line.startswith(';') or line.startswith('.') or line.endswith(':'):
            continue
        mnem, ops = split_operands(line)
        if mnem in ('s_branch', 's_endpgm', 's_setpc_b64'):
            outstanding = []
            continue
        if mnem == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', line)
            if m:
                n = int(m.group(1))
                if in_asm:
                    n_waits += 1
                if n == 0:
                    outstanding = []
                elif not any(o[0] for o in outstanding):
                    outstanding = outstanding[len(outstanding) - n:] if n < len(outstanding) else outstanding
            continue
        dest = set()
        src_ops = ops
        if ops and not mnem.startswith(NO_DEST):
            dest = regs(ops[0])
            src_ops = ops[1:]
        srcs = set()
        for o in src_ops:
            srcs |= regs(o)
        if mnem.startswith('v_mfma') or mnem.startswith('v_fma') or mnem.startswith('v_mac') or mnem.startswith('v_dot'):
            pass                                # (a destination that is also the accumulator source is listed among the sources already)
        live = {}
        for is_smem, d, at, from_asm in outstanding:
            if from_asm:
                for r in d:
                    live[r] = at
        for r in srcs & set(live):
            bad.append((name, no, f'reads {r[0]}{r[1]} of the asm ds_read at line {live[r]} before its wait: {line}'))
        for r in dest & set(live):
            bad.append((name, no, f'writes {r[0]}{r[1]} while the asm ds_read at line {live[r]} is in flight: {line}'))
        if mnem.startswith('ds_') or mnem.startswith('s_load') or mnem.startswith('s_buffer_load'):
            is_read = mnem.startswith('ds_read') or mnem.startswith('ds_load')
            if in_asm and is_read:
                n_reads += 1
            outstanding.append((not mnem.startswith('ds_'), dest if is_read else set(), no, in_asm and is_read))
    return n_reads, n_waits, bad


def check(asm):
    """-> list of (kernel, asm fragment reads, asm counted waits, scratch bytes, violations) for every kernel with asm fragment reads"""
    report = []
    lines = asm.split('\n')
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r'^(_Z\S+):\s*(;.*)?$', l)] if m]
    scratch = {n: int(s) for n, s in re.findall(r'\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size\s+(\d+)', asm, re.S)}
    for k, (i, name) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = [(j + 1, lines[j]) for j in range(i + 1, end)]
        for j, (_, l) in enumerate(body):
            if l.strip().startswith('.section') or l.strip().startswith('.rodata') or '.amdhsa_kernel' in l:
                body = body[:j]
                break
        n_reads, n_waits, bad = check_kernel(name, body)
        if n_reads:
            report.append((name, n_reads, n_waits, scratch.get(name, -1), bad))
    return report


if __name__ == '__main__':
    texts = [open(p).read() for p in sys.argv[1:]] or [assembly('pwstream.hip'), assembly('igemm.hip')]
    worst = 0
    for t in texts:
        for name, n_reads, n_waits, scr, bad in check(t):
            print(f'{name[:110]:110s} asm ds_reads {n_reads:4d}  counted waits {n_waits:4d}  scratch {scr}  violations {len(bad)}')
            for b in bad[:5]:
                print('     ', b[1], b[2])
            worst = max(worst, len(bad), 1 if scr else 0)
    sys.exit(1 if worst else 0)
