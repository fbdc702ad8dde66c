"""Generated-code check of the persistent implicit-GEMM kernels (no GPU needed: hipcc cross-compiles gfx950 assembly).

csrc/igemm.hip draws a workgroup's next tile with a returning atomic that is hidden from the compiler inside inline asm
(the compiler would wait vmcnt(0) right behind it -- the whole LDS-DMA ring plus an L2 round trip per tile).  The price is
that nothing tells the compiler the destination register is still in flight; scripts/check_ticket_regs.py verifies in the
assembly of EVERY persistent-capable instantiation that the register is written by the draws only, read by the mailbox
write only, never copied or spilled, and that such kernels use no scratch memory at all."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))


@pytest.mark.timeout(900)
def test_ticket_register_of_persistent_kernels_is_untouched_between_draw_and_use():
    import check_ticket_regs as C
    report = C.check(C.assembly())
    assert len(report) == 32                                  # 4 geometries x fwd / dgrad x pointwise / gather x plain / fused epilogue
    for name, reg, draws, boxes in report:
        assert draws >= 2 and boxes == 1, (name, reg, draws, boxes)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('source', ['elemwise.hip', 'detloss.hip', 'groupnorm.hip'])
def test_streaming_kernels_of_the_late_additions_keep_their_registers(source, tmp_path):
    """The HBM-bound kernels added for the convolutional backbones and the dense detectors (csrc/elemwise.hip, detloss.hip,
    groupnorm.hip) must not spill (no scratch memory) and must leave room for at least four wavefronts per SIMD (<= 128 vector
    registers): their only way to hide HBM latency is occupancy."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, 'simpleaicv_pytorch_training_examples_amd', 'csrc')
    out = str(tmp_path / 'k.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-S', '--cuda-device-only',
                    '-o', out, os.path.join(csrc, source)], check=True, capture_output=True)
    text = open(out).read()
    kernels = re.findall(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S)
    assert len(kernels) >= 6, source
    for name, body in kernels:
        scratch = int(re.search(r'\.amdhsa_private_segment_fixed_size\s+(\d+)', body).group(1))
        vgprs = int(re.search(r'\.amdhsa_next_free_vgpr\s+(\d+)', body).group(1))
        assert scratch == 0, (name, scratch)
        assert vgprs <= 128, (name, vgprs)


@pytest.mark.timeout(1200)
def test_asm_fragment_reads_are_not_touched_before_their_counted_wait():
    """the `ds_read_b128` fragment reads of csrc/igemm.hip and csrc/pwstream.hip are inline asm with `=v` outputs the
    compiler treats as valid at once, released by hand-counted `s_waitcnt lgkmcnt(N)` asm statements.  scripts/check_fragment_regs.py
    replays the LGKM counter over the generated assembly of every kernel: no instruction reads or writes a fragment register while
    its read is outstanding (a copy or spill the register allocator slipped in, or a compiler-generated LDS / scalar-memory
    operation that made a count wrong, would show), and those kernels use no scratch.  The checker itself is checked by mutation:
    a wait count raised by two, and a move inserted behind a read, must both be reported."""
    import re
    import check_fragment_regs as F
    import check_ticket_regs as C
    pw = F.assembly('pwstream.hip')
    reports = F.check(pw) + F.check(C.assembly())
    assert len([r for r in reports if 'pw_stream_kernel' in r[0]]) == 3           # the nine-tap forms: plain, + statistics, + data-gradient extras
    assert len(reports) >= 60, len(reports)
    for name, n_reads, n_waits, scratch, bad in reports:
        assert n_reads >= 1 and n_waits >= 1 and scratch == 0 and not bad, (name, n_reads, n_waits, scratch, bad[:3])
    # mutation 1: the first three counted waits release two reads too few
    lines, in_asm, changed = pw.split('\n'), False, 0
    for i, l in enumerate(lines):
        if 'ASMSTART' in l:
            in_asm = True
        elif 'ASMEND' in l:
            in_asm = False
        elif in_asm and 's_waitcnt lgkmcnt(' in l and changed < 3:
            n = int(re.search(r'lgkmcnt\((\d+)\)', l).group(1))
            lines[i] = l.replace(f'lgkmcnt({n})', f'lgkmcnt({n + 2})')
            changed += 1
    assert changed == 3 and sum(len(r[4]) for r in F.check('\n'.join(lines))) > 0
    # mutation 2: a copy of a fragment register right behind its read
    lines, in_asm = pw.split('\n'), False
    for i, l in enumerate(lines):
        if 'ASMSTART' in l:
            in_asm = True
        elif 'ASMEND' in l:
            in_asm = False
        elif in_asm and 'ds_read_b128' in l:
            lines.insert(i + 2, '\tv_mov_b32_e32 v200, v' + re.search(r'v\[(\d+):', l).group(1))
            break
    assert sum(len(r[4]) for r in F.check('\n'.join(lines))) == 1
