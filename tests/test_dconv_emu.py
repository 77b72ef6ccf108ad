"""CPU tests of the generated gfx950 assembly kernels (sota_imagenet_amd/csrc/asm/dconv_gen.py, pw_gen.py): the text the
generators print is executed by the functional emulator tools/gcn_emu.py — one workgroup at a time, exact small-integer
data — and compared with a numpy convolution (the same quantity oracle/ops_ref.conv2d_fwd restates; integer data makes
every summation order exact).  The emulator also enforces the LDS-DMA protocol the kernels rely on (counted vmcnt before
the issuing wave reads, + a barrier before any other wave reads, no rewrite of bytes another wave read in the same barrier
epoch), so a wrong wait count fails here and not as a rare wrong tile on the GPU.  No GPU needed."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import dconv_emu_check as D  # noqa: E402
import gcn_emu  # noqa: E402
import pw_emu_check as P  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.parametrize("name,kw", [
    ("dconv_l3_s1", dict(Cin=128, tiles=(0, 1))),                                      # forward + BN statistics, 2 chunks
    ("dconv_l3_s2", dict(Cin=64, NCOLS=512, ntile=1, tiles=(1,), dgrad_taps=True)),    # dgrad taps, BN-backward sums, 2nd column tile
    ("dconv_l3_s0", dict(Cin=192)),                                                    # odd chunk count (loop exit in the middle)
    ("dconv_l4_s1", dict(Cin=64, ntile=1)),                                            # 2 images per tile, 3-stage weight ring
    ("dconv_l4_s2", dict(Cin=128, tiles=(1,))),
    ("dconv_l2_s1", dict(Cin=128, tiles=(0, 1))),                                      # half-image tiles: both parities, 4 x 1 waves
    ("dconv_l2_s2", dict(Cin=64, tiles=(3,), dgrad_taps=True)),
    ("dconv_l1_s1", dict(tiles=(0, 1, 13))),                                           # 14 row tiles per image: first / middle / last, one chunk
    ("dconv_l1_s2", dict(tiles=(14, 27), dgrad_taps=True)),
    # the other sizes of the progressive-resize recipe: a = 160 px (40 / 20 / 10 / 5), b = 320 px (80 / 40 / 20 / 10)
    ("dconv_l3a_s1", dict(Cin=128, tiles=(0, 1))), ("dconv_l4a_s2", dict(Cin=128, tiles=(1,))),                 # 10 x 10 images; four 5 x 5 images per tile
    ("dconv_l2a_s2", dict(Cin=64, tiles=(3,), dgrad_taps=True)), ("dconv_l1a_s1", dict(tiles=(0, 1, 4))),      # 10-row / 8-row tiles at pitch 32 / 48
    ("dconv_l4b_s1", dict(Cin=64, ntile=1)), ("dconv_l3b_s2", dict(Cin=64, tiles=(4, 6), dgrad_taps=True)),    # 5-row tiles of 20 x 20 images
    ("dconv_l2b_s1", dict(Cin=64, tiles=(0, 2, 4))), ("dconv_l1b_s2", dict(tiles=(20, 39), dgrad_taps=True)),  # pitch 48 / 96: three / six fragments per row
    # BASELINE configs[3] (BResNet-50): the deep stem's 3x3s at 112 x 112 (pitch 128), conv2 of the striding blocks at their input resolution
    ("dconv_v0_s1", dict(tiles=(0, 27, 29))), ("dconv_v0_s0", dict(tiles=(30,), dgrad_taps=True)),
    ("dconv_v2_s1", dict(Cin=64, tiles=(0, 13, 14))), ("dconv_v3_s1", dict(Cin=64, tiles=(0, 3, 5))), ("dconv_v4_s0", dict(Cin=64, tiles=(1,), ntile=1, dgrad_taps=True)),
])
def test_direct_conv_kernels_are_exact_in_the_emulator(name, kw):
    r = D.run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"], r
    if "stat_err" in r:
        assert r["stat_err"] < 1e-6, r


@pytest.mark.parametrize("name,kw", [
    ("dconv_l3_s1_q", dict(Cin=384, tiles=(1,))),                                      # three 128-channel chunks, BN statistics
    ("dconv_l3_s0_q", dict(Cin=256, over_scales=False)),                               # the output scale alone (no input / weight scale pointers)
    ("dconv_l3_s2_q", dict(Cin=128, NCOLS=512, ntile=1, dgrad_taps=True)),             # one chunk, dgrad taps, BN-backward sums from the late register pair
    ("dconv_l2_s2_q", dict(tiles=(3,), dgrad_taps=True)),                              # half-image tiles
    ("dconv_l4_s1_q", dict(Cin=256, ntile=1)),                                         # two images per tile, 2nd column tile
])
def test_e4m3_direct_conv_kernels_are_exact_in_the_emulator(name, kw):
    """the dconv_*_q variants (Cfg.fp8: e4m3 operands, 128-channel chunks, v_mfma_f32_16x16x128_f8f6f4, output x oscale / (sc_in * sc_wt))
    against the fp32 convolution of the decoded bytes"""
    r = D.run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"], r
    if "stat_err" in r:
        assert r["stat_err"] < 1e-6, r


@pytest.mark.parametrize("mod,name,kw", [
    ("dconv", "dconv_l3_s3", dict(Cin=64, NCOLS=512, ntile=1, tiles=(1,), dgrad_taps=True)),
    ("dconv", "dconv_l1_s3", dict(tiles=(14, 27), dgrad_taps=True)),
    ("dconv", "dconv_v3_s3", dict(Cin=64, tiles=(0, 3), dgrad_taps=True)),
    ("po", "po_k256_b64_s3_a0", dict(M=300, N=64, tpg=2, groups=((1, 0), (0, 0)))),
    ("po", "po_k512_b128_s3_a0", dict(M=150, N=256, tpg=2, groups=((1, 1), (0, 0)))),
    ("pk", "pk_k1024_n256_w196_s3", dict(tiles=(0, 2), Cin=192)),
    ("pk", "pk_k2048_n512_w98_s3", dict(tiles=(0,), ntile=1, Cin=192)),
])
def test_bn_backward_sums_under_a_leaky_mask_are_exact_in_the_emulator(mod, name, kw):
    """the stats == 3 epilogues (BASELINE configs[3]'s leaky ReLU, slope 0.01: dz = dx where the mask bit is set, fp32(dx) * fp32(0.01) elsewhere) of
    the three generator families: outputs exact, sums against fp64"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import importlib

    r = importlib.import_module(mod + "_emu_check").run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"] and r["stat_err"] < 1e-6, r


@pytest.mark.parametrize("name,kw", [
    ("dconv_l3_d2_s2", dict(Cin=128, tiles=(1,))),                                 # whole-image tiles, 2 chunks, BN-backward sums, all four classes
    ("dconv_l3_d2_s0", dict(Cin=192, classes=(0, 3))),                             # odd chunk count: the 1-tap class requests weights two chunks ahead
    ("dconv_l3_d2_s0", dict(Cin=64, NCOLS=512, ntile=1, classes=(1, 2))),          # one chunk; class and column tile from workgroup id y
    ("dconv_l2_d2_s2", dict(Cin=128, tiles=(0, 3))),                               # half-image tiles: first / last rows of an image, 4 x 1 waves
    ("dconv_l4_d2_s2", dict(Cin=128, tiles=(1,), ntile=1)),                        # two 7 x 7 images per tile (two rows per fragment), 2nd column tile
])
def test_stride2_data_gradient_kernels_are_exact_in_the_emulator(name, kw):
    """the data gradient of the stride-2 3x3 convolutions by output-parity classes (Cfg.s2d) against the transposed convolution in numpy"""
    r = D.run_s2d(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"], r
    assert r["stat_err"] < 1e-6, r


@pytest.mark.parametrize("name,kw", [
    ("dconv_l3_s1_bn", dict(Cin=192)),                          # 3 chunks: the prologue's transform, two in-loop ones, the skipped look-ahead tile
    ("dconv_l3_s1_bn", dict(Cin=64, NCOLS=512, ntile=1)),       # one chunk, second column tile
    ("dconv_l2_s1_bn", dict(Cin=128, tiles=(0, 1, 3))),         # two blocks per substep; first / last half-image tiles (halo rows skipped)
    ("dconv_l1_s1_bn", dict(tiles=(0, 1, 13, 14))),             # one chunk (transform in the prologue only), three tile classes, slots without a block
    ("dconv_l4_s1_bn", dict(Cin=128, tiles=(1,), ntile=1)),     # two images per tile, 3-stage weight ring
])
def test_forward_kernels_with_the_inputs_batchnorm_in_the_operand_path_are_exact_in_the_emulator(name, kw):
    """Cfg.bnin: out = conv(relu(y * scale + shift)) with the activation and its ReLU bits left in memory, on dyadic data (every step exact)"""
    r = D.run_bn(name, **kw)
    assert r["max_err"] == 0.0 and r["a_ok"] and r["bits_ok"], r
    assert r["stat_err"] < 1e-6, r


@pytest.mark.parametrize("name,kw", [
    ("po_k64_b256_s1_a0_bn", dict(M=256)),                                                         # four tiles: prologue transform, three in the loop, the skipped look-ahead
    ("po_k64_b256_s1_a0_bn", dict(M=64)),                                                          # a single tile
    ("po_k128_b256_s1_a0_bn", dict(M=320, N=512, tpg=3, groups=((0, 0), (1, 1), (1, 0)))),           # two runs x two column tiles: only column tile 0 stores a / bits
    ("po_k256_b256_s1_a0_bn", dict(M=192, N=1024, groups=((0, 3), (0, 0)))),                         # four planes, eight pieces per wave and tile
])
def test_pointwise_kernels_with_the_inputs_batchnorm_in_the_operand_path_are_exact_in_the_emulator(name, kw):
    """PoCfg.bnin: out = relu(y * scale + shift) @ w^T with the activation and its ReLU bits left in memory by the workgroups of column tile 0"""
    import po_emu_check as PO

    r = PO.run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"] and r["a_ok"] and r["bits_ok"], r
    assert r["stat_err"] < 1e-6, r


@pytest.mark.parametrize("name,kw", [
    ("pw_k256_n1024_s1", dict(mtiles=2, grid=2, N=512)),   # pixel-tile change inside a workgroup's range, statistics rows
    ("pw_k256_n1024_s1", dict(mtiles=3, grid=2)),          # 6 units per workgroup: both accumulator sets, refill path and not
    ("pw_k256_n1024_s0", dict(mtiles=2, grid=3, N=256)),   # one column tile: every unit refills
    ("pw_k256_n1024_s1", dict(mtiles=2, grid=2, K=128, N=512)),  # 2 planes: a weight stage two units ahead
])
def test_pointwise_kernels_are_exact_in_the_emulator(name, kw):
    r = P.run(name, **kw)
    assert r["max_err"] == 0.0, r
    if "stat_err" in r:
        assert r["stat_err"] == 0.0, r


def _tiny(body):
    return "\n".join(["t:"] + ["\t" + l for l in body] + ["\ts_endpgm"])


def test_the_emulator_rejects_an_lds_read_before_the_dma_wait():
    import numpy as np

    mem = gcn_emu.Memory()
    src = mem.alloc(np.arange(1024, dtype=np.uint8))
    ka = mem.alloc(np.frombuffer(gcn_emu.pack_kernarg([("q", src)]), dtype=np.uint8))
    head = ["s_load_dwordx2 s[4:5], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)", "s_and_b32 s5, s5, 0xffff", "s_mov_b32 s6, 1024",
            "s_mov_b32 s7, 0x00020000", "v_lshlrev_b32 v1, 4, v0", "s_mov_b32 m0, 0", "s_nop 0",
            "buffer_load_dwordx4 v1, s[4:7], 0 offen lds"]
    bad = _tiny(head + ["ds_read_b128 v[4:7], v1", "s_waitcnt lgkmcnt(0)"])
    with pytest.raises(gcn_emu.EmuError, match="not been waited for"):
        gcn_emu.Emulator(bad, mem, lds_bytes=4096).run_workgroup(1, ka)
    good = _tiny(head + ["s_waitcnt vmcnt(0)", "ds_read_b128 v[4:7], v1", "s_waitcnt lgkmcnt(0)"])
    gcn_emu.Emulator(good, mem, lds_bytes=4096).run_workgroup(1, ka)


@pytest.mark.parametrize("name,kw", [("wg3_l3", dict(splits=2, tps=3, pairs=((3, 2),))), ("wg3_l3", dict(splits=3, tps=1, pairs=((0, 1),))),
                                     ("wg3_l2", dict(splits=3, tps=2, pairs=((1, 0),))), ("wg3_l4", dict(splits=2, tps=2, pairs=((5, 7),))),
                                     ("wg3_l1", dict(splits=1, tps=29)), ("wg3_s112", dict(splits=2, tps=3)),
                                     # 160 / 320 px: pitch 48 (six 8-position blocks per row: pieces dealt rows x blocks as 2 x 2), 5-row tiles, 10 x 10 images
                                     ("wg3_l1a", dict(splits=2, tps=6)), ("wg3_l2a", dict(splits=3, tps=3, pairs=((1, 0),))), ("wg3_l3a", dict(splits=2, tps=2, pairs=((3, 2),))),
                                     ("wg3_l1b", dict(splits=1, tps=5)), ("wg3_l2b", dict(splits=2, tps=3, pairs=((0, 1),))), ("wg3_l3b", dict(splits=3, tps=3, pairs=((2, 3),))),
                                     ("wg3_l4b", dict(splits=2, tps=1, pairs=((5, 7),))),
                                     # BResNet-50's striding blocks (conv2 at the input resolution)
                                     ("wg3_v2", dict(splits=2, tps=3, pairs=((1, 0),))), ("wg3_v3", dict(splits=1, tps=5, pairs=((3, 2),))), ("wg3_v4", dict(splits=2, tps=2, pairs=((7, 1),)))])
def test_weight_gradient_kernels_are_exact_in_the_emulator(name, kw):
    """csrc/asm/wg_gen.py: odd and single tile counts per split, splits that end inside an image (row tiles), last channel tiles;
    every slab element of the run workgroups exact, nothing else written, no LDS-DMA protocol violation"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import wg_emu_check

    r = wg_emu_check.run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"]


@pytest.mark.parametrize("name,kw", [("wg1_c256_o1024", dict(splits=2, tps=4, pairs=((1, 3),))), ("wg1_c1024_o256", dict(splits=1, tps=5, npix=300, pairs=((7, 0),))),
                                     ("wg1_c512_o2048", dict(splits=2, tps=1, pairs=((3, 7),))), ("wg1_c2048_o512", dict(splits=1, tps=7, npix=400, pairs=((15, 1),))),
                                     ("wg1_c512_o128", dict(splits=1, tps=3, npix=150, pairs=((1, 0),))), ("wg1_c64_o256", dict(splits=2, tps=2, pairs=((0, 0),))),
                                     ("wg1_c256_o64", dict(splits=1, tps=4, pairs=((0, 0),)))])
def test_pointwise_weight_gradient_kernels_are_exact_in_the_emulator(name, kw):
    """csrc/asm/wg1_gen.py: tile counts of 1, 4, 5 and 7 per split (every exit of the three-buffer loop), a ragged last tile (300 and
    400 pixels), first and last channel tiles"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import wg1_emu_check

    r = wg1_emu_check.run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"]


@pytest.mark.parametrize("name,kw", [("pk_k1024_n256_w196_s1", dict(tiles=(1,), Cin=256)), ("pk_k1024_n256_w196_s2", dict(tiles=(0, 2), Cin=448)),
                                     ("pk_k2048_n512_w98_s1", dict(tiles=(1,), ntile=1, Cin=320)), ("pk_k2048_n512_w98_s2", dict(tiles=(0,), ntile=1, Cin=192)),
                                     ("pk_k1024_n256_w196_s0", dict(tiles=(0,), Cin=64)), ("pk_k2048_n512_w98_s0", dict(tiles=(2,), Cin=384)),
                                     ("pk_k512_n128_w196_s1", dict(tiles=(1,), Cin=192)), ("pk_k512_n128_w196_s2", dict(tiles=(0,), Cin=128)),
                                     # tiles of 200 / 100 pixels (the 160 px and 320 px stages)
                                     ("pk_k1024_n256_w200_s2", dict(tiles=(1,), Cin=192)), ("pk_k2048_n512_w100_s1", dict(tiles=(0, 2), ntile=1, Cin=128)),
                                     ("pk_k512_n128_w200_s2", dict(tiles=(1,), Cin=128))])
def test_long_reduction_pointwise_kernels_are_exact_in_the_emulator(name, kw):
    """csrc/asm/pk_gen.py with 1 .. 7 chunks of 64 channels (every exit of the unrolled buffer rotation), both column tiles, the three
    epilogues (none / BN statistics / BN-backward sums, the latter with the late second register set at 13 fragments)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pk_emu_check

    r = pk_emu_check.run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"] and r.get("stat_err", 0.0) <= 1e-6


@pytest.mark.parametrize("name,kw", [
    ("po_k64_b256_s1_a0", dict(M=200, groups=((0, 0),))),                                   # forward + BN statistics, ragged last tile
    ("po_k64_b256_s2_a2", dict(M=300, N=512, tpg=3, groups=((1, 1), (0, 0)))),              # two runs, second column tile, masked addend, BN-backward sums
    ("po_k256_b256_s2_a2", dict(M=130, N=1024, groups=((0, 3),))),                          # 4 column tiles (layer 3's conv1 data gradient)
    ("po_k256_b256_s2_a1", dict(M=640, N=512, tpg=1, groups=((9, 1), (3, 0)))),             # one tile per run, run index above 8 (second workgroup row of an XCD)
    ("po_k256_b256_s1_a0", dict(M=192, N=256, tpg=2, groups=((1, 0),))),
    ("po_k256_b256_s0_a0", dict(M=64, N=256, groups=((0, 0),))),
    ("po_k512_b128_s2_a2", dict(M=150, N=1024, tpg=2, groups=((1, 7), (0, 2)))),            # 128-column tiles: one tile pair per wave
    ("po_k512_b128_s1_a0", dict(M=128, N=256, groups=((0, 1),))),
    ("po_k256_b128_s1_a0", dict(M=200, N=128, tpg=2, groups=((1, 0),))),                    # 256 -> 128 (layer 2's first conv1): ragged last tile
    ("po_k128_b256_s2_a2", dict(M=330, N=512, tpg=4, groups=((1, 0), (0, 1)))),
    ("po_k128_b256_s0_a2", dict(M=100, N=256, groups=((0, 0),))),
    ("po_k64_b256_s0_a1", dict(M=70, N=256, groups=((0, 0),))),
    ("po_k64_b256_s2_a0", dict(M=640, N=256, tpg=1, groups=((8, 0), (9, 0)))),
    # the addend at half resolution (the downsample branch's data gradient: even pixels only), 10 x 20 / 4 x 6 / 6 x 10 pixel images
    ("po_k256_b256_s2_a3", dict(M=400, N=512, HW=(10, 20), tpg=4, groups=((1, 1), (0, 0)))),
    ("po_k128_b256_s0_a3", dict(M=192, N=256, HW=(4, 6), groups=((0, 0),))),
    ("po_k512_b128_s2_a3", dict(M=120, N=256, HW=(6, 10), groups=((0, 1),))),
    # 64 columns (layer 1's launches into 64 channels): waves 2 x 2, 128-pixel tiles, two statistics rows per workgroup
    ("po_k256_b64_s1_a0", dict(M=300, N=64, tpg=2, groups=((1, 0), (0, 0)))),               # ragged last tile in the second wave pair's half
    ("po_k256_b64_s2_a0", dict(M=200, N=64, groups=((0, 0),))),                             # conv3's data gradient + bn2's sums
    ("po_k256_b64_s0_a0", dict(M=129, N=64, groups=((0, 0),))),                             # the downsample branch's data gradient
    ("po_k64_b64_s2_a2", dict(M=260, N=128, tpg=1, groups=((2, 1), (0, 0)))),               # two column tiles, masked addend
    ("po_k64_b64_s0_a1", dict(M=70, N=64, groups=((0, 0),))),                               # layer1.0's conv1 data gradient + the shortcut gradient
])
def test_output_heavy_pointwise_kernels_are_exact_in_the_emulator(name, kw):
    """csrc/asm/po_gen.py (weights resident in AGPRs, rolling refill of the epilogue operands): every (K, BN) family with each epilogue —
    none / BN statistics / BN-backward sums x no addend / addend / addend under its ReLU bits; pixel counts that are not multiples of the
    64-pixel tile; 1 .. 5 tiles per workgroup (both exits of the two-buffer loop); the statistics row of a run; nothing else written; no
    register consumed before its load was waited for and no LDS-DMA protocol violation"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import po_emu_check

    r = po_emu_check.run(name, **kw)
    assert r["max_err"] == 0.0 and r["untouched_ok"] and r.get("stat_err", 0.0) <= 1e-6, r


def test_a_workgroup_beyond_the_last_run_of_a_launch_exits_without_touching_memory():
    """the grid is rounded up to whole XCD rows: workgroups whose run index is >= the run count end at once"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import po_emu_check

    r = po_emu_check.run("po_k64_b256_s1_a0", M=128, N=256, tpg=1, groups=(), extra_wgs=(2, 7))
    assert r["untouched_ok"] and r["insts"] < 400, r


def test_the_launch_plan_covers_every_tile_once_at_the_shapes_of_the_step():
    """dconv.cpp plan_po restated (tools/po_emu_check.plan): runs x tiles per run cover the tile count exactly once, every run has a
    tile, and the grid holds whole XCD rows of (run, column tile) pairs"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import po_emu_check

    for (M, N, BN) in ((256 * 3136, 256, 256), (256 * 784, 512, 256), (256 * 196, 1024, 256), (256 * 49, 2048, 128), (256 * 3136, 256, 256), (8 * 49, 2048, 128),
                       (512 * 1600, 256, 256), (3 * 196, 1024, 256)):
        T, nct, tpg, G, grid = po_emu_check.plan(M, N, BN, 64)
        assert (G - 1) * tpg < T <= G * tpg and grid % (8 * nct) == 0 and grid >= G * nct
        seen = set()
        for x in range(grid):
            xcd, l = x % 8, x // 8
            g, ct = (l // nct) * 8 + xcd, l % nct
            if g < G:
                assert (g, ct) not in seen
                seen.add((g, ct))
                assert po_emu_check.wg_of(g, ct, nct) == x
        assert len(seen) == G * nct


def test_the_transposed_lds_read_of_the_emulator_follows_the_documented_lane_map():
    """ds_read_b64_tr_b16 (guide T10): per 16 lanes a 4 x 16 block; lane 4q + p addresses row q, columns 4p .. 4p + 3; lane i receives
    column i, row q in element q"""
    import numpy as np

    mem = gcn_emu.Memory()
    ka = mem.alloc(np.zeros(8, dtype=np.uint8))
    # LDS image: 16 rows of 128 bytes, element (row, col) = row * 64 + col as u16; every group reads rows 4G .. 4G + 3, columns 16 .. 31
    prog = ["v_and_b32 v1, 63, v0", "v_lshlrev_b32 v2, 1, v1", "v_lshlrev_b32 v3, 2, v1"]
    prog += ["v_mov_b32 v4, %d" % 0]
    # fill: lane l writes dwords: element pairs (2k, 2k + 1) of row r: use ds_write_b32 in a loop over 16 rows x 32 dwords = 512 dwords / 64 lanes
    for k in range(8):
        prog += ["v_add_u32 v5, %d, v1" % (64 * k),            # dword index d = 64k + lane -> row = d >> 5, col pair = d & 31
                 "v_lshrrev_b32 v6, 5, v5", "v_and_b32 v7, 31, v5", "v_lshlrev_b32 v8, 6, v6", "v_lshl_add_u32 v8, v7, 1, v8",   # first element value
                 "v_add_u32 v9, 1, v8", "v_lshl_add_u32 v9, v9, 16, v8", "v_lshlrev_b32 v10, 2, v5", "ds_write_b32 v10, v9"]
    prog += ["s_waitcnt lgkmcnt(0)",
             "v_lshrrev_b32 v11, 4, v1", "v_and_b32 v12, 15, v1", "v_lshrrev_b32 v13, 2, v12", "v_and_b32 v14, 3, v12",
             "v_lshl_add_u32 v13, v11, 2, v13",                  # row = 4G + q
             "v_lshlrev_b32 v15, 7, v13", "v_lshl_add_u32 v15, v14, 3, v15", "v_add_u32 v15, 32, v15",   # + column 16 (32 bytes) + 8p
             "ds_read_b64_tr_b16 v[16:17], v15", "s_waitcnt lgkmcnt(0)"]
    emu = gcn_emu.Emulator(_tiny(prog), mem, lds_bytes=4096)
    emu.run_workgroup(1, ka)
    wv = emu.waves[0]
    for lane in range(64):
        G, i = lane >> 4, lane & 15
        want = [(4 * G + q) * 64 + 16 + i for q in range(4)]
        got = [int(wv.v[16][lane]) & 0xFFFF, int(wv.v[16][lane]) >> 16, int(wv.v[17][lane]) & 0xFFFF, int(wv.v[17][lane]) >> 16]
        assert got == want, (lane, got, want)


def test_the_emulator_rejects_a_read_of_another_waves_dma_without_a_barrier():
    import numpy as np

    mem = gcn_emu.Memory()
    src = mem.alloc(np.arange(2048, dtype=np.uint8))
    ka = mem.alloc(np.frombuffer(gcn_emu.pack_kernarg([("q", src)]), dtype=np.uint8))
    head = ["s_load_dwordx2 s[4:5], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)", "s_and_b32 s5, s5, 0xffff", "s_mov_b32 s6, 2048",
            "s_mov_b32 s7, 0x00020000", "v_lshlrev_b32 v1, 4, v0",                      # tid*16: wave w fills bytes 1024w ..
            "v_and_b32 v2, 63, v0", "v_lshlrev_b32 v2, 4, v2",
            "v_lshrrev_b32 v3, 6, v0", "s_nop 0", "v_readfirstlane_b32 s8, v3", "s_nop 3", "s_lshl_b32 s8, s8, 10", "s_mov_b32 m0, s8", "s_nop 0",
            "buffer_load_dwordx4 v1, s[4:7], 0 offen lds", "s_waitcnt vmcnt(0)",
            "v_mov_b32 v4, v2"]                                                          # wave 0's bytes, read by both waves
    bad = _tiny(head + ["ds_read_b128 v[8:11], v4", "s_waitcnt lgkmcnt(0)"])
    with pytest.raises(gcn_emu.EmuError, match="not yet published by a barrier"):
        gcn_emu.Emulator(bad, mem, lds_bytes=4096).run_workgroup(2, ka)
    good = _tiny(head + ["s_barrier", "ds_read_b128 v[8:11], v4", "s_waitcnt lgkmcnt(0)"])
    gcn_emu.Emulator(good, mem, lds_bytes=4096).run_workgroup(2, ka)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not installed")
def test_every_shipped_variant_assembles_for_gfx950_within_the_register_and_lds_budget(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "sota_imagenet_amd", "csrc", "asm"))
    import dconv_gen
    import pw_gen
    import pk_gen
    import po_gen
    import wg1_gen
    import wg_gen

    for mod in (dconv_gen, pw_gen, pk_gen, po_gen, wg_gen, wg1_gen):
        for name in mod.VARIANTS:
            c, g, text = mod.generate(name)
            assert g.accum_offset + g.nagpr <= 512
            lds = g.lds_bytes if hasattr(g, "lds_bytes") else c.LDS
            assert lds <= 160 * 1024
            f = tmp_path / (name + ".s")
            f.write_text(text)
            subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(f), "-o", str(tmp_path / (name + ".o"))],
                           check=True, capture_output=True, timeout=120)
