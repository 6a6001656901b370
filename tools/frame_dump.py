"""Print the block structure of a zstd frame (RFC 8878): one line per block with its literals section and the head of its
sequences section, then the skippable frames behind it.  A debugging aid for zstd_encode.hip's output.

    from frame_dump import dump; dump(frame_bytes)"""
import sys


def dump(buf, out=sys.stdout, limit=40):
    b = bytes(buf)
    p = 0
    magic = int.from_bytes(b[0:4], "little")
    if magic != 0xFD2FB528:
        print("not a zstd frame: magic %08x" % magic, file=out)
        return
    fhd = b[4]
    p = 5
    single = (fhd >> 5) & 1
    if not single:
        p += 1
    p += (0, 1, 2, 4)[fhd & 3]
    fl = fhd >> 6
    fsz = (1 if single else 0) if fl == 0 else (2, 4, 8)[fl - 1]
    fcs = int.from_bytes(b[p : p + fsz], "little") + (256 if fsz == 2 else 0)
    p += fsz
    print("frame: content %d, header %d bytes, checksum %d" % (fcs, p, (fhd >> 2) & 1), file=out)
    k = 0
    while True:
        bh = int.from_bytes(b[p : p + 3], "little")
        last, bt, bs = bh & 1, (bh >> 1) & 3, bh >> 3
        line = "block %3d @%7d type %d size %6d%s" % (k, p, bt, bs, " last" if last else "")
        q = p + 3
        if bt == 2:
            h0 = b[q]
            lt, fmt = h0 & 3, (h0 >> 2) & 3
            if lt < 2:
                if fmt in (0, 2):
                    lh, regen = 1, h0 >> 3
                elif fmt == 1:
                    lh, regen = 2, int.from_bytes(b[q : q + 2], "little") >> 4
                else:
                    lh, regen = 3, int.from_bytes(b[q : q + 3], "little") >> 4
                cs = regen if lt == 0 else 1
                streams = 0
            else:
                v = int.from_bytes(b[q : q + 5], "little")
                if fmt < 2:
                    lh, regen, cs, streams = 3, (v >> 4) & 0x3FF, (v >> 14) & 0x3FF, 1 if fmt == 0 else 4
                elif fmt == 2:
                    lh, regen, cs, streams = 4, (v >> 4) & 0x3FFF, (v >> 18) & 0x3FFF, 4
                else:
                    lh, regen, cs, streams = 5, (v >> 4) & 0x3FFFF, (v >> 22) & 0x3FFFF, 4
            sq = q + lh + cs
            n0 = b[sq]
            if n0 == 0:
                seq = "no sequences"
            else:
                if n0 < 128:
                    ns, u = n0, 1
                elif n0 < 255:
                    ns, u = ((n0 - 128) << 8) + b[sq + 1], 2
                else:
                    ns, u = b[sq + 1] + (b[sq + 2] << 8) + 0x7F00, 3
                modes = b[sq + u]
                seq = "%d sequences, modes %02x (LL %d OF %d ML %d), section %d bytes" % (ns, modes, modes >> 6, (modes >> 4) & 3, (modes >> 2) & 3, p + 3 + bs - sq)
                if (modes >> 4) & 3 == 1:
                    seq += ", OF code %d" % b[sq + u + 1 + (1 if modes >> 6 == 1 else 0)]
            line += "  literals type %d regen %6d comp %6d streams %d | %s" % (lt, regen, cs, streams, seq)
        if k < limit or last:
            print(line, file=out)
        p += 3 + (1 if bt == 1 else bs)
        k += 1
        if last:
            break
    if (fhd >> 2) & 1:
        p += 4
    while p + 8 <= len(b):
        m, sz = int.from_bytes(b[p : p + 4], "little"), int.from_bytes(b[p + 4 : p + 8], "little")
        print("skippable frame @%d magic %08x payload %d" % (p, m, sz), file=out)
        p += 8 + sz
    print("end @%d of %d" % (p, len(b)), file=out)
