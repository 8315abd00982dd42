"""Static checks of the built gfx950 code (no GPU): instructions issued from inline assembly sit outside the compiler's hazard
recogniser, so the wait states they need are verified in the machine code of the library that ships.
v_permlane32_swap_b32 (kernels_match_16bit.hip, the count-tile kernels' one-list-per-query epilogue): two wait states between a VALU
write of an operand and the swap."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("lib", ["libr3dm.so", "libr3dm_dev.so"])
def test_lane_half_swaps_keep_their_wait_states(lib):
    import check_swap_hazard
    path = os.path.join(ROOT, "regard3d_amd", lib)
    if not os.path.exists(path):
        pytest.skip(f"{lib} is not built")
    n, bad = check_swap_hazard.check(path)
    assert n >= 100, f"expected the count-tile kernels' swaps in {lib}, found {n}"
    assert not bad, bad[:5]
