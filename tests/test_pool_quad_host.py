"""Host emulation of the index logic of csrc/pool.hip bn_relu_maxpool_bwd_apply_k3s2_kernel (late r06): for MaxPool(3, 2, 1) one thread
owns the 2 x 2 block of input pixels (2a .. 2a+1, 2b .. 2b+1) and the four windows (a .. a+1, b .. b+1).  For every pixel the set of
(window, position inside the window) pairs the block form visits must be exactly what the per-pixel form (pooled_grad: the windows
oh_lo .. oh_hi x ow_lo .. ow_hi and `want = (h - (oh * 2 - 1)) * 3 + (w - (ow * 2 - 1))`) visits, in the same order."""
import pytest


def per_pixel(h, w, H, W, OH, OW, K=3, stride=2, pad=1):
    oh_lo, oh_hi = max(0, (h + pad - (K - 1) + stride - 1) // stride), min(OH - 1, (h + pad) // stride)
    ow_lo, ow_hi = max(0, (w + pad - (K - 1) + stride - 1) // stride), min(OW - 1, (w + pad) // stride)
    out = []
    for q in range(4):
        oh, ow = oh_lo + (q >> 1), ow_lo + (q & 1)
        if oh <= oh_hi and ow <= ow_hi:
            out.append((oh, ow, (h - (oh * stride - pad)) * K + (w - (ow * stride - pad))))
    return out


def block_form(H, W, OH, OW):
    seen = {}
    for a in range((H + 1) // 2):
        for b in range((W + 1) // 2):
            for p in range(4):
                r, c = p >> 1, p & 1
                h, w = 2 * a + r, 2 * b + c
                if h >= H or w >= W:
                    continue
                out = []
                for q in range(4):
                    dy, dx = q >> 1, q & 1
                    if (r == 0 and dy == 1) or (c == 0 and dx == 1):
                        continue
                    oh, ow = a + dy, b + dx
                    if not (oh < OH and ow < OW):
                        continue
                    kh = 1 if r == 0 else (2 if dy == 0 else 0)
                    kw = 1 if c == 0 else (2 if dx == 0 else 0)
                    out.append((oh, ow, kh * 3 + kw))
                assert (h, w) not in seen
                seen[(h, w)] = out
    return seen


@pytest.mark.parametrize('hw', [(1, 1), (2, 3), (5, 7), (6, 6), (33, 47), (112, 112)])
def test_block_form_visits_the_windows_of_the_per_pixel_form(hw):
    H, W = hw
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    assert (OH, OW) == ((H + 1) // 2, (W + 1) // 2)              # the condition under which the host selects the block form
    seen = block_form(H, W, OH, OW)
    assert len(seen) == H * W
    for h in range(H):
        for w in range(W):
            assert seen[(h, w)] == per_pixel(h, w, H, W, OH, OW), (h, w)
            assert all(0 <= pos <= 8 for _, _, pos in seen[(h, w)])
