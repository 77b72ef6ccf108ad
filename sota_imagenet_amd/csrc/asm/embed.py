#!/usr/bin/env python3
"""embed.py <out dir> <linked .hsaco> — writes the include files dconv.cpp embeds: the code object as a byte array
(dconv_blob.inc) and one initialiser per generated kernel (dconv_meta.inc: direct 3x3 kernels, pw_meta.inc: pointwise, wg_meta.inc / wg1_meta.inc: 3x3 / 1x1 weight gradient,
pk_meta.inc: long-reduction pointwise, po_meta.inc: output-heavy pointwise with resident weights)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dconv_gen  # noqa: E402
import pw_gen  # noqa: E402
import wg_gen  # noqa: E402
import wg1_gen  # noqa: E402
import pk_gen  # noqa: E402
import po_gen  # noqa: E402


def main():
    out_dir, hsaco = sys.argv[1], sys.argv[2]
    blob = open(hsaco, "rb").read()
    with open(os.path.join(out_dir, "dconv_blob.inc"), "w") as f:
        for i in range(0, len(blob), 32):
            f.write(",".join(str(b) for b in blob[i:i + 32]) + ",\n")
    with open(os.path.join(out_dir, "dconv_meta.inc"), "w") as f:
        for name in dconv_gen.VARIANTS:
            c, g, _ = dconv_gen.generate(name)
            words = ",".join("%du" % w for par in dconv_gen.tables(c) for row in par for w in row)
            f.write('{"%s", %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, {%s}},\n' % (name, c.H, c.W, c.IPT, c.TPI, c.BN, c.Cin, c.NCOLS, c.stats, c.s2d, c.bnin, c.fp8, g.lds_bytes, g.ka_size, words))
    with open(os.path.join(out_dir, "dconv_tt.inc"), "w") as f:   # the transform tables of the kernels with the input's BatchNorm in their operand path
        for name in dconv_gen.VARIANTS:
            c, g, _ = dconv_gen.generate(name)
            if c.bnin:
                words = ",".join("%du" % w for par in dconv_gen.ttables(c) for row in par for w in row)
                f.write('{"%s", {%s}},\n' % (name, words))
    with open(os.path.join(out_dir, "pw_meta.inc"), "w") as f:
        for name in pw_gen.VARIANTS:
            c, g, _ = pw_gen.generate(name)
            words = ",".join("%du" % w for row in pw_gen.tables(c) for w in row)
            f.write('{"%s", %d, %d, %d, %d, %d, %d, {%s}},\n' % (name, c.K, c.N, c.stats, c.ROWS, c.LDS, pw_gen.Gen.KA["size"], words))
    with open(os.path.join(out_dir, "wg_meta.inc"), "w") as f:
        for name in wg_gen.VARIANTS:
            c, g, _ = wg_gen.generate(name)
            tn, ti = c.TPI_NUM
            f.write('{"%s", %d, %d, %d, %d, %d, %d, %d, %d},\n' % (name, c.H, c.W, c.C, c.CO, tn, ti, g.lds_bytes, wg_gen.Gen.KA["size"]))
    with open(os.path.join(out_dir, "wg1_meta.inc"), "w") as f:
        for name in wg1_gen.VARIANTS:
            c, g, _ = wg1_gen.generate(name)
            f.write('{"%s", %d, %d, %d, %d, %d, %d},\n' % (name, c.C, c.CO, c.XP, c.DP, g.lds_bytes, wg1_gen.Gen.KA["size"]))
    with open(os.path.join(out_dir, "pk_meta.inc"), "w") as f:
        for name in pk_gen.VARIANTS:
            c, g, _ = pk_gen.generate(name)
            f.write('{"%s", %d, %d, %d, %d, %d, %d, %d},\n' % (name, c.W, c.Cin, c.NCOLS, c.BN, c.stats, g.lds_bytes, pk_gen.Gen.KA["size"]))
    with open(os.path.join(out_dir, "po_meta.inc"), "w") as f:
        for name in po_gen.VARIANTS:
            c, g, _ = po_gen.generate(name)
            f.write('{"%s", %d, %d, %d, %d, %d, %d, %d, %d, %d},\n' % (name, c.K, c.BN, c.stats, c.add, c.TP, c.WM, c.bnin, g.lds_bytes, po_gen.Gen.KA["size"]))


if __name__ == "__main__":
    main()
