"""Static check of the persistent NT kernels' asynchronous ticket (csrc/igemm.hip, draw_ticket): the returning atomic is
hidden from the compiler inside inline asm, so nothing in the generated code may read, copy or overwrite its destination
register between the draw and the mailbox write two K steps later.  For every igemm_nt_kernel instantiation:
  * every `saicv ticket` atomic of the kernel writes the SAME VGPR (the loop-carried value was coalesced, no copy);
  * that VGPR is read only by the instruction(s) feeding a `saicv mailbox` ds_write, and written by nothing else;
  * neither the mailbox access nor the ticket draw is followed by a compiler-inserted `s_waitcnt vmcnt(0)` in its block.
Usage: python scripts/check_ticket_regs.py [igemm.s]   (without an argument the source is compiled to assembly first)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'simpleaicv_pytorch_training_examples_amd', 'csrc', 'igemm.hip')


def assembly(path=None):
    if path:
        return open(path).read()
    from simpleaicv_pytorch_training_examples_amd import build
    out = os.path.join(tempfile.gettempdir(), 'saicv_igemm_check.s')
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(SRC):
        cmd = [build._hipcc()] + build.FLAGS + ['-S', '--cuda-device-only', SRC, '-o', out]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(asm):
    for m in re.finditer(r'^(_ZN\S*igemm_nt_kernel\S*?):[^\n]*\n', asm, re.M):
        end = asm.index('.Lfunc_end', m.end())
        yield m.group(1), asm[m.end():end].split('\n')


def uses(line, reg):
    """(reads, writes) of VGPR `reg` (e.g. 'v233') by one instruction line; register ranges v[a:b] included."""
    code = line.split(';')[0].strip()
    if not code or code.endswith(':') or code.startswith('.'):
        return False, False
    n = int(reg[1:])
    parts = code.split(None, 1)
    if len(parts) < 2:
        return False, False
    ops = [o.strip() for o in parts[1].split(',')]

    def hit(o):
        if re.fullmatch(r'v%d' % n, o):
            return True
        r = re.fullmatch(r'v\[(\d+):(\d+)\]', o)
        return bool(r and int(r.group(1)) <= n <= int(r.group(2)))
    if parts[0].startswith('v_pk_'):
        # packed-math sources are register PAIRS; with op_sel = 0 and op_sel_hi = 0 for a source both halves of the result
        # read its LOW register only (a broadcast scalar): the upper register of that pair is not an operand
        sel = re.search(r'op_sel:\[([01,]+)\]', code)
        hi = re.search(r'op_sel_hi:\[([01,]+)\]', code)
        sel = [int(x) for x in sel.group(1).split(',')] if sel else [0, 0, 0]
        hi = [int(x) for x in hi.group(1).split(',')] if hi else [1, 1, 1]
        ops = [o.split(' op_sel')[0].strip() for o in ops]
        for k in range(1, min(4, len(ops))):
            r = re.fullmatch(r'v\[(\d+):(\d+)\]', ops[k])
            if r and int(r.group(2)) == int(r.group(1)) + 1 and k - 1 < len(sel) and sel[k - 1] == 0 and hi[k - 1] == 0:
                ops[k] = 'v%s' % r.group(1)
    if parts[0].startswith('v_mad_u64_u32') and len(ops) == 5:
        # hipcc's 32-bit a * b + c: the addend is passed as a 64-bit pair whose upper register is whatever happens to sit
        # there; only the low half of the result is used.  The upper register is read by the hardware but not by the program.
        r = re.fullmatch(r'v\[(\d+):(\d+)\]', ops[4])
        if r:
            ops[4] = 'v%s' % r.group(1)
    stores = parts[0].startswith(('ds_write', 'global_store', 'buffer_store', 'scratch_store', 'flat_store', 's_'))
    w = (not stores) and hit(ops[0])
    r = any(hit(o) for o in (ops if stores else ops[1:]))
    return r, w


def check(asm):
    report = []
    for name, lines in kernels(asm):
        draws = [i for i, l in enumerate(lines) if 'saicv ticket' in l]
        boxes = [i for i, l in enumerate(lines) if 'saicv mailbox' in l]
        if not draws:
            raise AssertionError(f'{name}: a streaming-kernel instantiation without a ticket draw')
        spills = [l for l in lines if 'scratch_' in l]
        if spills:
            raise AssertionError(f'{name}: a persistent instantiation must not touch scratch memory: {spills[0].strip()}')
        regs = {lines[i].split()[1].rstrip(',') for i in draws}
        if len(regs) != 1:
            raise AssertionError(f'{name}: ticket lands in several registers {regs}: a copy sits between draw and use')
        reg = regs.pop()
        allowed = set(draws)
        for b in boxes:                     # the few instructions that turn the ticket into a tile index
            allowed.update(range(max(0, b - 24), b + 1))
        for i, l in enumerate(lines):
            if i < draws[0]:
                continue                    # straight-line set-up in front of the first draw (tk = 0)
            r, w = uses(l, reg)
            if w and not r and re.fullmatch(r'v_mov_b32(_e32)? %s, 0' % reg, l.split(';')[0].strip()):
                continue                    # `tk = 0` on a path that never drew (a workgroup without a tile)
            if (r or w) and i not in allowed:
                raise AssertionError(f'{name}: line {i} touches the ticket register {reg} outside draw / mailbox: {l.strip()}')
        consumed = any(uses(lines[i], reg)[0] for b in boxes for i in range(max(0, b - 24), b + 1))
        if not consumed:
            raise AssertionError(f'{name}: the mailbox write does not read {reg}')
        report.append((name, reg, len(draws), len(boxes)))
    return report


if __name__ == '__main__':
    sys.path.insert(0, ROOT)
    rep = check(assembly(sys.argv[1] if len(sys.argv) > 1 else None))
    for name, reg, nd, nb in rep:
        print(f'{reg:>5} draws {nd} mailbox writes {nb}  {name}')
    print(f'{len(rep)} igemm_nt_kernel instantiations OK')
