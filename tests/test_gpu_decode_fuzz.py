"""(GPU) A short run of tests/tools/gpu_decode_fuzz.py inside the suite: adversarial map outputs written by the oracle (LZ4Block,
SnappyOutputStream, LZF chunks) through the reduce side on the real machine — both decode variants, the batched entry point, damaged
copies with the checksums off.  The batch decoder's parse blocks and exact-length stores are hand-written since round 6; the interpreter
runs the same generators on the CPU, this is the hardware's word (long campaigns: profiles/r06u_*)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


@pytest.mark.parametrize("codec", ["lz4", "snappy", "lzf"])
def test_adversarial_images_round_trip_on_the_gpu(codec):
    import gpu_decode_fuzz as g

    rounds, mb, refused, bad = g.run(seed=606, seconds=6.0, parts_n=32, codec_sel=codec)
    assert bad == 0 and rounds >= 3 and refused >= 1, (rounds, mb, refused, bad)
