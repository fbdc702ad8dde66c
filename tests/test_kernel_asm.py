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
